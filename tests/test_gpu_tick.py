"""The per-tick prologue on the device and the one-round-trip solve `acmpc_control_tick`:
waypoints -> speed profile (ADMM) -> t2s -> linearise, as the first node of the solve's hipGraph.

Oracles: the reference's own vectors (tests/golden/reference_ingredients.npz: G1 waypoint tables, G3 linearisation,
G6 speed-profile QP inputs) through the oracle's restatements, and the library's host path (`acmpc_waypoint_table`,
`acmpc_speed_profile_qp`, `acmpc_set_paths`), which the CPU tests pin to the same vectors.
"""
import copy

import numpy as np
import pytest

import acmpc_oracle as orc
from test_support import RACING, PlaceholderVehicle

pytestmark = pytest.mark.gpu

CASES = ["monza_H20_hairpin_10", "monza_H50_hairpin_25", "monza_H50_chicane_70", "monza_H50_curve_0.02",
         "monza_H50_straight_146.667", "nordschleife_H80_hairpin_10"]


def _engine(n, n_candidates=1024, v_max=28.0, v_min=8.0, track="monza"):
    from acmpc_amd import MODE_SPATIAL, Engine
    cfg = RACING[track]
    lim = orc.vehicle_limits(2.65, 1.94, 0.30, v_min, v_max)
    lo, hi = orc.input_box(lim)
    return Engine(mode=MODE_SPATIAL, max_problems=1, max_candidates=n_candidates, max_steps=n,
                  step_cost=cfg["step_cost"], r_term=cfg["r_term"], final_cost=cfg["final_cost"], u_min=lo, u_max=hi,
                  margin=lim.margin, wheelbase=lim.length), lim, (lo, hi)


def _tick(H, cons, n_candidates=1024, rounds=2, offset=0.0, localised=False, seed=5, qp_max_iter=4000, check_every=10,
          qp_method=0):
    from acmpc_amd import _capi
    t = _capi.Tick()
    t.struct_size = _capi.C.sizeof(_capi.Tick)
    t.horizon, t.localised, t.has_end_velocity = H, int(localised), 1
    t.n_candidates, t.rounds, t.centre_is_reference = n_candidates, rounds, 1
    t.qp_max_iter, t.qp_check_every = qp_max_iter, check_every
    t.qp_method = qp_method     # 0 = the QP's exact optimum in two sweeps (the default), 1 = always the OSQP-style splitting
    t.offset = offset
    t.v_min, t.v_max, t.a_min, t.a_max = cons["v_min"], cons["v_max"], cons["a_min"], cons["a_max"]
    t.ay_max, t.ki_min, t.end_velocity = cons["ay_max"], cons["ki_min"], cons["end_velocity"]
    t.sigma[0], t.sigma[1], t.shrink = 0.5, 1e-3, 0.5
    t.qp_eps_abs = t.qp_eps_rel = 1e-3
    t.seed = seed
    return t


@pytest.mark.parametrize("n", [2, 19, 49, 99, 128])
def test_device_admm_is_bit_identical_to_the_host_solver(n):
    """Same statement of the algorithm (csrc/acmpc_admm.h) on one wavefront and on the host: every float64 bit of the
    primal and dual iterate, the status and the iteration count agree - cold, warm-started, and through a step-size
    update (the long cold solve)."""
    from acmpc_amd import _capi
    eng, _, _ = _engine(max(n, 8))
    rng = np.random.default_rng(n)
    v_hi = np.clip(20 + 8 * np.sin(np.arange(n) / 7.0) + rng.normal(0, 1.5, n), 8.0, 30.0)
    v_hi[-1] = 14.0
    ds = rng.uniform(2.0, 3.5, n)
    args = (v_hi, ds, -1.3, 1.0, 8.0)
    host = _capi.speed_profile_qp(*args, max_iter=4000)
    dev = eng.speed_profile_qp_device(*args, max_iter=4000)
    assert host[2:] == dev[2:]
    np.testing.assert_array_equal(dev[0], host[0])
    np.testing.assert_array_equal(dev[1], host[1])
    # warm start from that iterate on a perturbed ceiling; the stopping test every 5 iterations
    v_hi2 = np.clip(v_hi + rng.normal(0, 0.3, n), 8.0, 30.0)
    host2 = _capi.speed_profile_qp(v_hi2, ds, -1.3, 1.0, 8.0, warm=host[:2], check_every=5)
    dev2 = eng.speed_profile_qp_device(v_hi2, ds, -1.3, 1.0, 8.0, warm=host[:2], check_every=5)
    assert host2[2:] == dev2[2:]
    np.testing.assert_array_equal(dev2[0], host2[0])
    np.testing.assert_array_equal(dev2[1], host2[1])
    # an equality row (v_min == v_hi somewhere) and an iteration cap that stops the solve early
    v_hi3 = v_hi.copy()
    v_hi3[n // 2] = 8.0
    host3 = _capi.speed_profile_qp(v_hi3, ds, -1.3, 1.0, 8.0, max_iter=37)
    dev3 = eng.speed_profile_qp_device(v_hi3, ds, -1.3, 1.0, 8.0, max_iter=37)
    assert host3[2:] == dev3[2:]
    np.testing.assert_array_equal(dev3[0], host3[0])
    eng.close()


@pytest.mark.parametrize("qp_method", [0, 1])
@pytest.mark.parametrize("case", CASES)
def test_prologue_against_the_reference_vectors(golden, case, qp_method):
    """What the prologue kernel leaves for the rollout, against the reference: the waypoint table (G1), the Frenet
    start state (G2), the linearisation + corridor rows (G3, through the oracle's packing) and the QP's ceiling (G6,
    through the profile the QP returns) - the QP solved exactly by the two sweeps (qp_method 0, the default: the host's
    sweeps on the reference's own ceiling, no iteration) and by the splitting that restates OSQP (1: the host solver's
    profile and iteration count)."""
    from acmpc_amd import _capi
    coords = golden[case + "/coords"]
    H = coords.shape[0]
    n = H - 1
    track = case.split("_")[0]
    cons = dict(RACING[track]["speed_profile_constraints"])     # as the vectors were made: v_max = 84
    offset = float(golden[case + "/offset"])
    eng, lim, (lo, hi) = _engine(n, v_max=cons["v_max"], v_min=cons["v_min"], track=track)
    tick = _tick(H, cons, offset=offset)
    tick.qp_method = qp_method
    out = eng.control_tick(tick, np.ascontiguousarray(coords, dtype=np.float64), None)
    table = out["table"]
    # G1: construct_waypoints (the device's atan2 may differ from libm's in the last float64 bit)
    np.testing.assert_allclose(table[:6], golden[case + "/table_no_v"][:6], rtol=0, atol=1e-12)
    # the speed profile: the host solver on the reference's own ceiling (G6), same warm state (none), same settings
    v_hi = golden[case + "/sp0_v_hi"]
    swept = _capi.speed_profile_exact(v_hi, table[4], cons["a_min"], cons["a_max"], cons["v_min"]) if qp_method == 0 else None
    if swept is not None:
        host_v, iters = swept[0], 0
        assert out["info"][4] == 0.0 and int(out["info"][5]) == 0
    else:
        host_v, _, status, iters = _capi.speed_profile_qp(v_hi, table[4], cons["a_min"], cons["a_max"], cons["v_min"])
        assert status == "solved" and out["info"][4] == 0.0 and int(out["info"][5]) == iters
    np.testing.assert_allclose(table[6], host_v, rtol=0, atol=1e-9)
    # G2: t2s of the pose (offset, 0, pi/2); G3: linearise + corridor rows, as float32 (1 ulp: the tolerance stated
    # for the device's float64 transcendentals)
    x0, u_ref, coef = eng.tick_device_tables(n)
    want_x0 = golden[case + "/spatial_state"].astype(np.float32)
    np.testing.assert_allclose(x0, want_x0, rtol=0, atol=float(np.spacing(np.float32(max(1.0, abs(want_x0[0]))))))
    full = table.copy()
    full[6] = host_v
    want = orc.coefficients_spatial(full, lim.margin)
    ulp = np.spacing(np.abs(want).astype(np.float32)).astype(np.float64)
    assert np.all(np.abs(coef.astype(np.float64) - want.astype(np.float64)) <= ulp + 1e-30), "packed table beyond 1 ulp"
    f, A, B = orc.linearise(full)
    np.testing.assert_allclose(coef[:, 4], f[:, 2], rtol=2e-7)
    np.testing.assert_allclose(coef[:, 1], A[:, 1, 0], rtol=2e-7, atol=1e-30)
    want_u = np.stack([np.clip(host_v, lo[0], hi[0]), np.clip(full[3], lo[1], hi[1])], axis=1).astype(np.float32)
    np.testing.assert_allclose(u_ref, want_u, rtol=2e-7, atol=1e-12)
    # and the library's own host path agrees with the device's tables the same way
    eng.set_paths(full)
    np.testing.assert_allclose(coef, eng.coefficients(0), rtol=2e-7, atol=1e-30)
    eng.close()


def test_tick_equals_set_paths_plus_optimize():
    """The rounds behind the prologue are acmpc_optimize's: with the tables the prologue made handed to the
    two-call path (set_paths + optimize, same seed, same spread) the winner's record is the same, bit for bit."""
    coords = np.stack([0.004 * np.linspace(0, 120, 50) ** 2, np.linspace(0, 120, 50), np.linspace(10, 6, 50)], axis=1)
    cons = dict(RACING["monza"]["speed_profile_constraints"], v_max=28.0)
    n = 49
    eng, _, _ = _engine(n, n_candidates=4096)
    t = _tick(50, cons, n_candidates=4096, rounds=3, offset=0.25, seed=77)
    out = eng.control_tick(t, coords, None)
    x0, u_ref, coef = eng.tick_device_tables(n)
    eng.set_paths(out["table"])
    eng.set_coefficients(coef)   # the device's own packed table (a host cos / sin may differ from it in a last float32 bit)
    best = eng.optimize(x0[None], u_ref[None], u_ref[None], 4096, 3, (0.5, 1e-3), shrink=0.5, seed=77)
    np.testing.assert_array_equal(best["records"][0], out["record"])
    # dec.x layout and the unpacked plan
    rec = out["record"]
    np.testing.assert_array_equal(out["decision"][:3 * (n + 1)], rec[4 + 2 * n:].astype(np.float64))
    np.testing.assert_array_equal(out["decision"][3 * (n + 1):], rec[4:4 + 2 * n].astype(np.float64))
    np.testing.assert_allclose(out["projected_control"][0], rec[4:4 + 2 * n:2], rtol=0, atol=0)
    np.testing.assert_allclose(out["cum_time"], rec[4 + 2 * n + 2::3][:n], rtol=0, atol=0)
    eng.close()


@pytest.mark.parametrize("H,N,rounds", [(3, 100, 2), (4, 64, 1), (65, 8192, 3), (129, 4096, 2)])
def test_tick_at_the_horizon_limits(H, N, rounds):
    """The shortest horizon the controller can pose (two control steps), one step more, the first horizon whose
    operands need a second register set, and the longest the device prologue takes (128 steps: the record is re-rolled
    there, a trace would not fit the LDS): each tick equals the two-call path on the tables the prologue made."""
    n = H - 1
    y = np.linspace(0, 2.5 * H, H)
    coords = np.stack([0.002 * y ** 2, y, np.linspace(10, 6, H)], axis=1)
    cons = dict(RACING["monza"]["speed_profile_constraints"], v_max=28.0)
    eng, _, _ = _engine(n, n_candidates=N)
    t = _tick(H, cons, n_candidates=N, rounds=rounds, offset=0.1, seed=11)
    out = eng.control_tick(t, coords, None)
    assert out["info"][4] == 0 and out["info"][7] == 0 and np.isfinite(out["record"]).all()
    x0, u_ref, coef = eng.tick_device_tables(n)
    eng.set_paths(out["table"])
    eng.set_coefficients(coef)   # (as above: the comparison is unconditional)
    best = eng.optimize(x0[None], u_ref[None], u_ref[None], N, rounds, (0.5, 1e-3), shrink=0.5, seed=11)
    np.testing.assert_array_equal(best["records"][0], out["record"])
    # a second, warm-started tick on the same handle
    centre = out["decision"][3 * (n + 1):].reshape(n, 2).astype(np.float32)
    t2 = _tick(H, cons, n_candidates=N, rounds=rounds, offset=0.12, seed=12)
    t2.centre_is_reference = 0
    again = eng.control_tick(t2, coords, centre)
    assert again["info"][7] == 0 and again["record"][0] <= out["record"][0] * 1.5 + 1.0
    eng.close()


def test_tick_forms_agree(monkeypatch):
    """However the tick is run - completion flag polled in pinned memory (default) or the stream synchronised, rounds on
    four waves per workgroup (default), two or one, chained or each finalized, the winner copied from its trace or rolled
    again - a sequence of warm-started ticks returns the same numbers."""
    cons = dict(RACING["monza"]["speed_profile_constraints"], v_max=28.0)
    n = 49
    paths = [np.stack([0.004 * (1 + 0.1 * j) * np.linspace(0, 120, 50) ** 2, np.linspace(0, 120, 50),
                       np.linspace(10, 6, 50)], axis=1) for j in range(4)]

    def run():
        eng, _, _ = _engine(n, n_candidates=16384)
        outs, centre = [], None
        for j, coords in enumerate(paths):
            t = _tick(50, cons, n_candidates=16384, rounds=2, offset=0.1 * j, seed=5 + j)
            t.centre_is_reference = 1 if centre is None else 0
            out = eng.control_tick(t, coords, centre)
            centre = out["decision"][3 * (n + 1):].reshape(n, 2).astype(np.float32)
            outs.append(out)
        eng.close()
        return outs

    reference = run()
    for switches in (("ACMPC_TICK_NO_FLAG",), ("ACMPC_NO_QUAD_ROUNDS",), ("ACMPC_NO_QUAD_ROUNDS", "ACMPC_NO_CHAINED_ROUNDS"),
                     ("ACMPC_NO_PAIR_ROUNDS",), ("ACMPC_NO_PAIR_ROUNDS", "ACMPC_NO_CHAINED_ROUNDS"),
                     ("ACMPC_NO_CHAINED_ROUNDS",), ("ACMPC_NO_TRACED_FINALIZE",)):
        for name in switches:
            monkeypatch.setenv(name, "1")
        for want, got in zip(reference, run()):
            for key in ("record", "table", "decision", "projected_control", "prediction", "cum_time", "coords"):
                np.testing.assert_array_equal(want[key], got[key], err_msg="%s with %s" % (key, switches))
            np.testing.assert_array_equal(want["info"][:6], got["info"][:6])
        for name in switches:
            monkeypatch.delenv(name)


def test_warm_state_lives_on_the_device_per_solver():
    """The splitting (qp_method 1): the second tick of the same solver warm-starts from the first (10 iterations instead
    of ~100); the localised solver has its own state; an iteration cap the solve cannot meet leaves the velocities at zero,
    as the reference keeps the path's (spatial_mpc.py:119-122)."""
    coords = np.stack([np.zeros(50), np.linspace(0, 120, 50), np.linspace(10, 6, 50)], axis=1)
    coords[:, 0] = 5.0 / (1 + np.exp(-0.1 * (coords[:, 1] - 60)))
    cons = dict(RACING["monza"]["speed_profile_constraints"], v_max=28.0)
    eng, _, _ = _engine(49)
    first = eng.control_tick(_tick(50, cons, qp_method=1), coords, None)["info"]
    second = eng.control_tick(_tick(50, cons, qp_method=1), coords, None)["info"]
    assert first[4] == 0 and second[4] == 0 and first[5] > 10 and second[5] == 10
    loc = eng.control_tick(_tick(50, cons, localised=True, qp_method=1), coords, None)
    assert loc["info"][4] == 0 and loc["info"][5] >= 10
    np.testing.assert_allclose(loc["table"][6], 28.0, atol=0.2)      # ceiling = v_max everywhere, no end velocity
    capped = eng.control_tick(_tick(50, cons, qp_max_iter=7, qp_method=1), coords, None)
    assert capped["info"][4] == 1 and capped["info"][5] == 7
    assert np.all(capped["table"][6] == 0.0)
    again = eng.control_tick(_tick(50, cons, qp_method=1), coords, None)["info"]   # the failed solve did not disturb the state
    assert again[4] == 0 and again[5] == 10
    eng.close()


def test_the_exact_profile_is_the_default_and_hands_infeasible_problems_to_the_splitting():
    """qp_method 0: every tick's profile is the QP's optimum from the two sweeps - no iterations, whatever the tick before
    was, equal to the host's sweeps bit for bit and within the splitting's tolerance of what that returns; the localised
    solver's is its ceiling cut down from the end; a problem without a feasible profile (the end velocity below v_min)
    goes to the splitting, whose verdict - and zero velocities - the tick reports as before."""
    from acmpc_amd import _capi
    coords = np.stack([np.zeros(50), np.linspace(0, 120, 50), np.linspace(10, 6, 50)], axis=1)
    coords[:, 0] = 5.0 / (1 + np.exp(-0.1 * (coords[:, 1] - 60)))
    cons = dict(RACING["monza"]["speed_profile_constraints"], v_max=28.0)
    eng, _, _ = _engine(49)
    split = eng.control_tick(_tick(50, cons, qp_method=1), coords, None)
    for _ in range(2):
        out = eng.control_tick(_tick(50, cons), coords, None)
        assert out["info"][4] == 0 and out["info"][5] == 0
        table = out["table"]
        ceiling = _capi.velocity_ceiling(table[3], cons["ay_max"], cons["ki_min"], cons["v_min"], cons["v_max"], False,
                                         cons["end_velocity"])
        swept = _capi.speed_profile_exact(ceiling, table[4], cons["a_min"], cons["a_max"], cons["v_min"])
        np.testing.assert_array_equal(table[6], swept[0])
        assert np.abs(table[6] - split["table"][6]).max() < 1.0
    loc = eng.control_tick(_tick(50, cons, localised=True), coords, None)
    assert loc["info"][4] == 0 and loc["info"][5] == 0 and np.all(loc["table"][6] == 28.0)
    hopeless = dict(cons, v_min=20.0)                                   # end velocity 14 < v_min: no feasible profile
    out = eng.control_tick(_tick(50, hopeless, qp_max_iter=60), coords, None)
    assert out["info"][4] == 1 and out["info"][5] == 60 and np.all(out["table"][6] == 0.0)
    out = eng.control_tick(_tick(50, cons), coords, None)               # and the next feasible tick is exact again
    assert out["info"][4] == 0 and out["info"][5] == 0
    eng.close()


@pytest.mark.parametrize("H", [3, 9, 33, 64, 65, 66, 100, 129])
def test_the_exact_profile_on_the_device_at_other_horizons(H):
    """The scan of the exact speed profile on the device - an element per lane up to 64 waypoints (H <= 65), through the
    workspace beyond - against the host's, bit for bit, on wiggly paths with random constraints: braking zones, acceleration
    limits and the end velocity all active somewhere."""
    from acmpc_amd import _capi
    rng = np.random.default_rng(H)
    n = H - 1
    eng, _, _ = _engine(n)
    held = 0
    for trial in range(6):
        y = np.linspace(0.0, 3.0 * n, H)
        x = rng.uniform(2.0, 12.0) * np.sin(y / rng.uniform(8.0, 40.0)) + 0.002 * rng.uniform(-1, 1) * y ** 2
        coords = np.stack([x - x[0], y, np.linspace(10, 6, H)], axis=1)
        cons = dict(RACING["monza"]["speed_profile_constraints"], v_max=float(rng.uniform(20.0, 86.0)),
                    a_min=-float(rng.uniform(0.3, 4.0)), a_max=float(rng.uniform(0.2, 3.0)), ay_max=float(rng.uniform(3.0, 9.0)),
                    end_velocity=float(rng.uniform(9.0, 20.0)))
        out = eng.control_tick(_tick(H, cons), coords, None)
        table = out["table"]
        ceiling = _capi.velocity_ceiling(table[3], cons["ay_max"], cons["ki_min"], cons["v_min"], cons["v_max"], False,
                                         cons["end_velocity"])
        swept = _capi.speed_profile_exact(ceiling, table[4], cons["a_min"], cons["a_max"], cons["v_min"])
        assert swept is not None and out["info"][4] == 0 and out["info"][5] == 0
        np.testing.assert_array_equal(table[6], swept[0])
        held += int((table[6] < ceiling - 1e-9).any())
    assert held > 0 or H < 9                                # rate rows are active in these problems
    eng.close()


def test_the_controller_takes_the_speed_profile_method_from_its_config():
    """`speed_profile_method` of the control config: "exact" (default) and "admm" drive the same laps - no solve rejected,
    reference speeds within the splitting's own tolerance band of each other - and are not the same numbers; the host
    statements of the prologue (`device_prologue: false`) follow the key the same way."""
    from acmpc_amd import workloads
    from acmpc_amd.mpc import build_mpc
    circuit = workloads.synthetic_track("monza")
    profiles = {}
    for method, device in (("exact", True), ("exact", False), ("admm", True)):
        cfg = copy.deepcopy(workloads.RACING_CONTROL["monza"])
        cfg["speed_profile_constraints"]["v_max"] = float(cfg["unlocalised_max_speed"])
        cfg.update(speed_profile_method=method, device_prologue=device, n_candidates=4096)
        mpc = build_mpc(cfg, PlaceholderVehicle())
        seen = []
        for i in range(25):
            centre = workloads.local_centreline(circuit, (i * 40) % len(circuit["centre"]), lateral_offset=0.2)
            mpc.get_control(workloads.reference_path_from_centreline(centre, cfg["horizon"]), offset=0.2)
            assert mpc.infeasibility_counter == 0
            seen.append(mpc.speed_profile.copy())
        profiles[(method, device)] = np.array(seen)
    # (the device's atan2 may differ from libm's in the last bit of a heading: 1e-12 on the curvature behind a ceiling)
    np.testing.assert_allclose(profiles[("exact", True)], profiles[("exact", False)], rtol=0, atol=1e-8)
    gap = np.abs(profiles[("exact", True)] - profiles[("admm", True)])
    assert 1e-6 < gap.max() < 3.0, gap.max()


@pytest.mark.parametrize("track", ["monza", "silverstone"])
def test_get_control_with_the_device_prologue_matches_the_host_prologue(track):
    """The drop-in controller with the prologue on the device against the same controller with the host statements
    of the same steps: same seeds, same schedule -> the same plans (to the float32 table's last bit)."""
    from acmpc_amd import workloads
    from acmpc_amd.mpc import build_mpc
    cfgs = []
    for device in (True, False):
        cfg = copy.deepcopy(workloads.RACING_CONTROL[track])
        cfg["speed_profile_constraints"]["v_max"] = float(cfg["unlocalised_max_speed"])
        cfg.update(device_prologue=device, n_candidates=4096, lq_candidate=False)   # (the tick's LQ plan is the previous
        # tick's, the host-prologue path's the current one's: this test compares the prologues)
        cfgs.append(cfg)
    a, b = (build_mpc(c, PlaceholderVehicle()) for c in cfgs)
    circuit = workloads.synthetic_track(track)
    for i in range(40):
        centre = workloads.local_centreline(circuit, (i * 4) % len(circuit["centre"]), lateral_offset=0.2)
        path = workloads.reference_path_from_centreline(centre, cfgs[0]["horizon"])
        a.get_control(path, offset=0.2)
        b.get_control(path, offset=0.2)
        assert a.infeasibility_counter == 0 and b.infeasibility_counter == 0
        np.testing.assert_allclose(a.reference_path.table, b.reference_path.table, rtol=0, atol=1e-9)
        np.testing.assert_allclose(a.projected_control, b.projected_control, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(a.cum_time, b.cum_time, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(a.current_prediction, b.current_prediction, rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(a.speed_profile, b.speed_profile, rtol=0, atol=1e-9)
