"""Mode T - the Cartesian kinematic rollout with nearest-waypoint projection, BASELINE.json north_star's literal shape -
through the drop-in entry point: `rollout_mode: "T"` in the control config -> `ControlSolver` creates a mode T handle,
the device prologue leaves the pose and the [n][8] waypoint rows, `acmpc_control_tick` runs the same fused rounds."""
import copy

import numpy as np
import pytest

import acmpc_oracle as orc
from test_support import RACING, PlaceholderVehicle

pytestmark = pytest.mark.gpu


def _engine(n, n_candidates, nn_window, track="monza", v_max=28.0, v_min=8.0):
    from acmpc_amd import MODE_TEMPORAL, Engine
    cfg = RACING[track]
    lim = orc.vehicle_limits(2.65, 1.94, 0.30, v_min, v_max)
    lo, hi = orc.input_box(lim)
    return Engine(mode=MODE_TEMPORAL, max_problems=1, max_candidates=n_candidates, max_steps=n,
                  step_cost=cfg["step_cost"], r_term=cfg["r_term"], final_cost=cfg["final_cost"], u_min=lo, u_max=hi,
                  margin=lim.margin, wheelbase=lim.length, dt=0.05, nn_window=nn_window)


def _tick(H, cons, n_candidates, rounds, offset, seed):
    from acmpc_amd import _capi
    t = _capi.Tick()
    t.struct_size = _capi.C.sizeof(_capi.Tick)
    t.horizon, t.localised, t.has_end_velocity = H, 0, 1
    t.n_candidates, t.rounds, t.centre_is_reference = n_candidates, rounds, 1
    t.qp_max_iter, t.qp_check_every = 4000, 10
    t.offset = offset
    t.v_min, t.v_max, t.a_min, t.a_max = cons["v_min"], cons["v_max"], cons["a_min"], cons["a_max"]
    t.ay_max, t.ki_min, t.end_velocity = cons["ay_max"], cons["ki_min"], cons["end_velocity"]
    t.sigma[0], t.sigma[1], t.shrink = 0.5, 1e-3, 0.5
    t.qp_eps_abs = t.qp_eps_rel = 1e-3
    t.seed = seed
    return t


@pytest.mark.parametrize("nn_window", [(2, 5), (1, 2), None])
@pytest.mark.parametrize("H,N,rounds", [(50, 4096, 3), (20, 1000, 2), (81, 2048, 2),
                                        (101, 1024, 2),    # the mapping controller's horizon: the frames still fit the LDS
                                        (105, 1024, 2)])   # ... and no longer do: three waves without them
def test_tick_equals_set_paths_plus_optimize_in_mode_t(nn_window, H, N, rounds):
    """tick(T) == set_paths + optimize(T), bit for bit: the prologue's pose and waypoint rows handed to the two-call
    path (same seed, same spread) give the same winner's record; and the record is what the oracle rolls."""
    n = H - 1
    y = np.linspace(0, 2.45 * H, H)
    coords = np.stack([0.004 * y ** 2, y, np.linspace(10, 6, H)], axis=1)
    cons = dict(RACING["monza"]["speed_profile_constraints"], v_max=28.0)
    eng = _engine(n, N, nn_window)
    out = eng.control_tick(_tick(H, cons, N, rounds, 0.25, 77), coords, None)
    assert out["info"][4] == 0 and out["info"][7] == 0
    x0, u_ref, coef = eng.tick_device_tables(n)
    assert coef.shape == (n, 8)
    np.testing.assert_array_equal(x0, np.array([0.25, 0.0, np.pi / 2], dtype=np.float32))
    eng.set_paths(out["table"])
    np.testing.assert_allclose(coef, eng.coefficients(0), rtol=2e-7, atol=1e-30)   # (device cos / sin: the last bit)
    rec = out["record"]
    eng.set_coefficients(coef)   # the device's own rows, so that the comparison below is unconditional
    best = eng.optimize(x0[None], u_ref[None], u_ref[None], N, rounds, (0.5, 1e-3), shrink=0.5, seed=77)
    np.testing.assert_array_equal(best["records"][0], rec)
    # the winner re-rolled by the oracle on the table the device used: cost, violation and every pose
    cfg = RACING["monza"]
    lim = orc.vehicle_limits(2.65, 1.94, 0.30, 8.0, 28.0)
    lo, hi = orc.input_box(lim)
    U = rec[4:4 + 2 * n].reshape(1, n, 2)
    cost, viol, X = orc.rollout_temporal(x0, coef, U, cfg["step_cost"], cfg["r_term"], cfg["final_cost"], lo, hi, 1.0e6,
                                         0.05, dtype=np.float32, return_states=True, nn_window=nn_window)[:3]
    assert rec[0] == cost[0] and rec[1] == viol[0]
    np.testing.assert_array_equal(rec[4 + 2 * n:].reshape(n + 1, 3), X[0])
    # the unpacked plan: poses as the prediction, time = i dt, derivatives of the controls
    np.testing.assert_array_equal(out["decision"][:3 * (n + 1)], rec[4 + 2 * n:].astype(np.float64))
    np.testing.assert_array_equal(out["prediction"], rec[4 + 2 * n:].reshape(n + 1, 3)[:n, :2].astype(np.float64))
    np.testing.assert_allclose(out["cum_time"], 0.05 * np.arange(n), rtol=0, atol=1e-15)
    np.testing.assert_array_equal(out["projected_control"][0], rec[4:4 + 2 * n:2].astype(np.float64))
    np.testing.assert_allclose(out["accelerations"], np.diff(out["projected_control"][0]) / 0.05, rtol=1e-12, atol=1e-12)
    eng.close()


@pytest.mark.parametrize("track", ["monza", "silverstone"])
def test_get_control_in_mode_t_device_prologue_matches_host_prologue(track):
    from acmpc_amd import workloads
    from acmpc_amd.mpc import build_mpc
    cfgs = []
    for device in (True, False):
        cfg = copy.deepcopy(workloads.RACING_CONTROL[track])
        cfg["speed_profile_constraints"]["v_max"] = float(cfg["unlocalised_max_speed"])
        cfg.update(device_prologue=device, n_candidates=4096, rollout_mode="T", lq_candidate=False)   # (the tick plans for the
        # previous tick's problem, the host-prologue path for the current one: without the LQ candidate the two paths solve alike)
        cfgs.append(cfg)
    a, b = (build_mpc(c, PlaceholderVehicle()) for c in cfgs)
    circuit = workloads.synthetic_track(track)
    for i in range(30):
        centre = workloads.local_centreline(circuit, (i * 4) % len(circuit["centre"]), lateral_offset=0.2)
        path = workloads.reference_path_from_centreline(centre, cfgs[0]["horizon"])
        a.get_control(path, offset=0.2)
        b.get_control(path, offset=0.2)
        assert a.infeasibility_counter == 0 and b.infeasibility_counter == 0
        np.testing.assert_allclose(a.reference_path.table, b.reference_path.table, rtol=0, atol=1e-9)
        np.testing.assert_allclose(a.projected_control, b.projected_control, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(a.current_prediction, b.current_prediction, rtol=1e-5, atol=1e-5)
        np.testing.assert_array_equal(a.cum_time, b.cum_time)
        assert a.cum_time[1] == 0.05 and a.current_prediction.shape == (49, 2)


def test_mode_t_controller_keeps_the_car_on_the_track():
    """The Monza closed loop of test_gpu_closed_loop.py with `rollout_mode: "T"`."""
    from test_gpu_closed_loop import _local_centreline
    from acmpc_amd import workloads
    from acmpc_amd.command_selection import TemporalCommandSelector
    from acmpc_amd.mpc import build_mpc, published_plan
    cfg = copy.deepcopy(workloads.RACING_CONTROL["monza"])
    cfg["speed_profile_constraints"]["v_max"] = float(cfg["unlocalised_max_speed"])
    cfg["rollout_mode"] = "T"
    mpc = build_mpc(cfg, workloads.PlaceholderVehicle())
    centre = workloads.synthetic_track("monza")["centre"]
    tangent = centre[1] - centre[0]
    pose = np.array([centre[0, 0], centre[0, 1], np.arctan2(tangent[1], tangent[0])])
    pose[:2] += 1.2 * np.array([-np.sin(pose[2]), np.cos(pose[2])])
    speed, dt, wheelbase = 15.0, 0.05, workloads.VEHICLE.wheelbase
    lateral, speeds, progress = [], [], []
    for tick in range(400):
        local, start = _local_centreline(centre, pose)
        mpc.get_control(workloads.reference_path_from_centreline(local, 50), elapsed=dt if tick > 0 else None)
        assert mpc.infeasibility_counter == 0, "infeasible solve at tick %d" % tick
        plan = published_plan(mpc)
        v_cmd, delta = TemporalCommandSelector(plan)(float(mpc.cum_time[1]))
        speed += np.clip(v_cmd - speed, -6.0 * dt, 4.0 * dt)
        pose = pose + np.array([speed * np.cos(pose[2]), speed * np.sin(pose[2]), speed * np.tan(delta) / wheelbase]) * dt
        nearest = centre[int(np.argmin(((centre - pose[:2]) ** 2).sum(axis=1)))]
        lateral.append(np.linalg.norm(pose[:2] - nearest))
        speeds.append(speed)
        progress.append(start)
    lateral = np.array(lateral)
    assert lateral.max() < 2.5, "left the 9.5 m road: max lateral error %.2f m" % lateral.max()
    assert lateral[100:].mean() < 0.8, "does not converge to the centreline: %.2f m" % lateral[100:].mean()
    assert 8.0 <= min(speeds[50:]) and max(speeds) <= 30.1
    assert (progress[-1] - progress[0]) % len(centre) > 400


def test_mode_t_tick_forms_agree(monkeypatch):
    """However the mode T rounds are run - three waves per workgroup (poses / search / costs, the default), one wave,
    chained or each finalized, the winner copied from its trace or re-drawn and rolled again - a sequence of warm-started
    ticks returns the same numbers, bit for bit."""
    cons = dict(RACING["monza"]["speed_profile_constraints"], v_max=28.0)
    n = 49
    paths = [np.stack([0.004 * (1 + 0.1 * j) * np.linspace(0, 120, 50) ** 2, np.linspace(0, 120, 50),
                       np.linspace(10, 6, 50)], axis=1) for j in range(4)]

    def run(nn_window):
        eng = _engine(n, 16384, nn_window)
        outs, centre = [], None
        for j, coords in enumerate(paths):
            t = _tick(50, cons, 16384, 2, 0.1 * j, 5 + j)
            t.centre_is_reference = 1 if centre is None else 0
            out = eng.control_tick(t, coords, centre)
            centre = out["decision"][3 * (n + 1):].reshape(n, 2).astype(np.float32)
            outs.append(out)
        eng.close()
        return outs

    for nn_window in ((2, 5), None, (3, 6)):      # the unrolled 8-waypoint window, every waypoint, a generic width
        reference = run(nn_window)
        for switches in (("ACMPC_NO_TRIO_ROUNDS",), ("ACMPC_NO_TRIO_ROUNDS", "ACMPC_NO_CHAINED_ROUNDS"),
                         ("ACMPC_NO_CHAINED_ROUNDS",), ("ACMPC_NO_TRACED_FINALIZE",),
                         ("ACMPC_NO_VERIFIED_SEARCH",)):   # (every waypoint: the scan instead of the verified window)
            for name in switches:
                monkeypatch.setenv(name, "1")
            for want, got in zip(reference, run(nn_window)):
                for key in ("record", "table", "decision", "projected_control", "prediction", "cum_time"):
                    np.testing.assert_array_equal(want[key], got[key], err_msg="%s with %s, window %s" % (key, switches, nn_window))
            for name in switches:
                monkeypatch.delenv(name)


@pytest.mark.parametrize("H,metres_per_point", [(9, 2.45), (20, 2.45), (50, 2.45), (81, 2.45), (101, 2.45), (107, 2.45),
                                                (50, 3.0), (100, 1.5), (100, 0.7), (107, 1.2)])   # (the first choice
def test_the_prologue_tabulates_the_frames_the_host_would(H, metres_per_point):                  # of `near`: 16 / 32 / 64)
    """Mode T with the exhaustive search through the tick: the prologue's lanes leave the frames of the verified window
    search beside the waypoint rows - the arithmetic of acmpc_set_paths (csrc/acmpc_frames.h), so the same bits as
    `acmpc_search_frames` gives for the table the device wrote."""
    from acmpc_amd import _capi
    n = H - 1
    y = np.linspace(0, metres_per_point * H, H)
    coords = np.stack([0.004 * y ** 2 + 3.0 * np.sin(y / 17.0), y, np.linspace(10, 6, H)], axis=1)
    cons = dict(RACING["monza"]["speed_profile_constraints"], v_max=28.0)
    eng = _engine(n, 1024, None)
    out = eng.control_tick(_tick(H, cons, 1024, 2, -0.3, 3), coords, None)
    assert out["info"][4] == 0
    _, _, coef = eng.tick_device_tables(n)
    frames = eng.tick_device_frames(n)
    want = _capi.search_frames(coef[None])[0]
    np.testing.assert_array_equal(frames, want)
    assert np.isfinite(frames[4::8]).all() and (frames[4::8] > 0).all()   # every window of this path is usable
    eng.close()
    windowed = _engine(n, 1024, (2, 5))
    windowed.control_tick(_tick(H, cons, 1024, 2, -0.3, 3), coords, None)
    with pytest.raises(_capi.EngineError):
        windowed.tick_device_frames(n)
    windowed.close()


def test_frames_from_the_bound_map():
    """The path cut out of the bound map (get_control_at): the second workgroup of the prologue computes the window's
    rows itself, as the first does - its frames are those of the table the first one wrote, at a map index and at a pose."""
    from acmpc_amd import _capi, workloads
    from acmpc_amd.mpc import build_mpc
    cfg = copy.deepcopy(workloads.RACING_CONTROL["silverstone"])
    cfg["speed_profile_constraints"]["v_max"] = float(cfg["unlocalised_max_speed"])
    cfg.update(rollout_mode="T", nn_window=None, n_candidates=2048)
    mpc = build_mpc(cfg, PlaceholderVehicle())
    track = workloads.synthetic_track("silverstone")
    mpc.bind_map(track)
    n = cfg["horizon"] - 1
    for step in range(6):
        if step % 2 == 0:
            mpc.get_control_at(map_index=1234 + 700 * step, lateral_offset=0.3, offset=0.3)
        else:
            centre = track["centre"]
            at = 800 * step
            tangent = centre[at + 1] - centre[at]
            normal = np.array([-tangent[1], tangent[0]]) / np.linalg.norm(tangent)
            mpc.get_control_at(pose=tuple(centre[at] + 0.4 * normal), lateral_offset=0.4, offset=0.4)
        assert mpc.infeasibility_counter == 0
        engine = mpc._control_solver._engine
        _, _, coef = engine.tick_device_tables(n)
        np.testing.assert_array_equal(engine.tick_device_frames(n), _capi.search_frames(coef[None])[0])
