"""Host-side mirror of the reference's controller interface, against golden vectors from the reference (no GPU)."""
from types import SimpleNamespace

import numpy as np
import pytest

import acmpc_oracle as orc
from test_support import RACING, PlaceholderVehicle


def _model(track="monza"):
    from acmpc_amd.bicycle_model import SpatialBicycleModel
    c = RACING[track]["speed_profile_constraints"]
    return SpatialBicycleModel(PlaceholderVehicle(), {"min": c["v_min"], "max": c["v_max"]})


def test_waypoint_table(golden, golden_cases):
    from acmpc_amd.mpc import waypoint_table
    for key in golden_cases:
        np.testing.assert_allclose(waypoint_table(golden[key + "/coords"]), golden[key + "/table_no_v"], rtol=0,
                                   atol=1e-12, err_msg=key)


def test_reference_path_views():
    from acmpc_amd.reference_path import ReferencePath
    path = ReferencePath(5)
    path.xs, path.kappas, path.velocities = np.arange(5), 0.1, [1, 2, 3, 4, 5]
    assert len(path) == 5 and path.table.shape == (7, 5)
    np.testing.assert_array_equal(path.table[0], np.arange(5))
    np.testing.assert_array_equal(path.table[3], np.full(5, 0.1))
    np.testing.assert_array_equal(path.table[6], [1, 2, 3, 4, 5])
    path.psis[2] = 0.5  # views are writable
    np.testing.assert_array_equal(path.get_state(2), [2.0, 0.0, 0.5])


def test_bicycle_model_against_reference(golden, golden_cases):
    from acmpc_amd.reference_path import ReferencePath
    model = _model()
    lim = golden[golden_cases[0] + "/limits"]
    assert (model.length, model.width, model.delta_max, model.margin) == tuple(lim[:4])
    np.testing.assert_array_equal(model.min_u, lim[4:6])
    np.testing.assert_array_equal(model.max_u, lim[6:8])
    got = np.array([model.t2s(w, s) for w, s in zip(golden["t2s/waypoints"], golden["t2s/states"])])
    np.testing.assert_allclose(got, golden["t2s/out"], rtol=0, atol=1e-12)
    for key in golden_cases[::3]:
        path = ReferencePath.from_table(golden[key + "/table"])
        f, A, B = model.linearise(path)
        np.testing.assert_allclose(f, golden[key + "/lin_f"], rtol=1e-14)
        np.testing.assert_allclose(A, golden[key + "/lin_A"], rtol=1e-14)
        np.testing.assert_allclose(B, golden[key + "/lin_B"], rtol=1e-14)
        np.testing.assert_allclose(model.s2t(path, golden[key + "/s2t_states"]), golden[key + "/s2t_out"], rtol=0,
                                   atol=1e-12)


def test_speed_profile_problem_matches_reference_inputs(golden, golden_cases):
    from acmpc_amd.reference_path import ReferencePath
    from acmpc_amd.speed_profile import LocalisedSpeedProfileSolver, SpeedProfileSolver
    for key in golden_cases[::2]:
        cons = RACING[key.split("_")[0]]["speed_profile_constraints"]
        path = ReferencePath.from_table(golden[key + "/table_no_v"])
        for cls, tag in ((SpeedProfileSolver, "sp0_"), (LocalisedSpeedProfileSolver, "sp1_")):
            solver = cls({"control_horizon": len(path), "max_iterations": 4000, "constraints": cons})
            prob = solver.problem(path, cons["end_velocity"])   # dense statement of what the native solver solves
            for name in ("q", "A", "l", "u"):
                np.testing.assert_allclose(prob[name], golden[key + "/" + tag + name], rtol=1e-14, atol=0)


@pytest.mark.parametrize("key", ["monza_H50_chicane_90", "monza_H50_hairpin_10", "spa_H50_chicane_70",
                                 "nordschleife_H80_chicane_70"])
def test_speed_profile_solution_satisfies_kkt(golden, key):
    """QP solutions are parity-unpinned (osqp absent), so optimality of the native solver is certified
    solver-independently: primal feasibility, dual signs and stationarity at a tight tolerance - on the QP data
    the REFERENCE assembled (golden A, l, u, q)."""
    from acmpc_amd import _capi
    from acmpc_amd.reference_path import ReferencePath
    from acmpc_amd.speed_profile import SpeedProfileSolver
    cons = RACING[key.split("_")[0]]["speed_profile_constraints"]
    path = ReferencePath.from_table(golden[key + "/table_no_v"])
    solver = SpeedProfileSolver({"control_horizon": len(path), "max_iterations": 4000, "constraints": cons})
    A, l, u, q = (golden[key + "/sp0_" + k] for k in ("A", "l", "u", "q"))
    ceiling = solver.velocity_ceiling(path, cons["end_velocity"])
    np.testing.assert_allclose(ceiling, golden[key + "/sp0_v_hi"], rtol=1e-14)
    x, y, status, iters = _capi.speed_profile_qp(ceiling, path.distances, cons["a_min"], cons["a_max"], cons["v_min"],
                                                 max_iter=200000, eps_abs=1e-9, eps_rel=1e-9)
    assert status == "solved"
    Ax = A @ x
    assert (Ax >= l - 1e-6).all() and (Ax <= u + 1e-6).all()
    np.testing.assert_allclose(x + q + A.T @ y, 0, atol=1e-6)       # P = I
    assert (y[Ax < u - 1e-5] <= 1e-6).all()   # y > 0 only on active upper bounds
    assert (y[Ax > l + 1e-5] >= -1e-6).all()  # y < 0 only on active lower bounds
    # the default-tolerance solve the controller uses reports "solved" within OSQP's own criterion, warm-starts
    first = solver.solve(path, cons["end_velocity"])
    second = solver.solve(path, cons["end_velocity"])
    assert first.info.status == "solved" and second.info.status == "solved" and second.info.iter <= first.info.iter
    assert np.abs(first.x - x).max() < 3.0    # 1e-3 relative tolerance on ~86 m/s
    # the oracle's independent dense restatement of the same published algorithm agrees
    ref = orc.osqp_restated(np.ones(len(path)), q, A, l, u, max_iter=200000, eps_abs=1e-9, eps_rel=1e-9, check_every=50)
    if ref.info.status == "solved":
        np.testing.assert_allclose(x, ref.x, atol=1e-4)


def test_whole_lap_speed_profile_is_linear_time():
    """compute_map_speed_profile (spatial_mpc.py:60-87) on a 10^4-waypoint lap: the tridiagonal solver handles it
    (a dense solver would need an 800 MB matrix) and the result is feasible."""
    import time
    from acmpc_amd import workloads
    from acmpc_amd.mpc import build_mpc
    mpc = build_mpc(RACING["monza"], PlaceholderVehicle())
    track = workloads.synthetic_track("monza")
    centre = track["centre"]
    coords = np.concatenate([centre, np.full((len(centre), 1), 9.5)], axis=1)
    path = mpc.construct_waypoints(coords)
    assert len(path) > 11000
    t0 = time.time()
    out = mpc.compute_map_speed_profile(path, ay_max=7.0, a_min=-0.15)     # configs/monza.yaml:83-84
    elapsed = time.time() - t0
    v = out.velocities
    acc = np.diff(v) / (2 * path.distances[:-1])
    tol = 1e-3 + 1e-3 * v.max()   # "solved" means within OSQP's eps_abs + eps_rel * |Ax| of feasibility
    assert v.min() >= 8.0 - tol and v.max() <= 84.0 + 2.0 + tol
    assert acc.min() >= -0.15 - tol and acc.max() <= 1.0 + tol
    assert elapsed < 60.0


def test_command_selection_against_reference(golden):
    from acmpc_amd.command_selection import TemporalCommandInterpolator, TemporalCommandSelector
    holder = SimpleNamespace(control_cumtime=golden["cmd/cum_time"], control_inputs=golden["cmd/commands"])
    selector = TemporalCommandSelector(holder)
    for t, want in zip(golden["cmd/elapsed"], golden["cmd/selected"]):
        np.testing.assert_array_equal(selector(t), want)
    interp = TemporalCommandInterpolator(SimpleNamespace(control_cumtime=golden["cmd/interp_b_cum_time"],
                                                         control_inputs=golden["cmd/interp_b_commands"]))
    expected = np.array([[17, -0.03], [4.2, 0.12], [-2.0, 0.02], [-0.5, 0.21], [6.9, 0.01], [-2.0, 0.02]])
    for t, want, ref in zip(golden["cmd/interp_b_elapsed"], expected, golden["cmd/interp_b_out"]):
        np.testing.assert_allclose(interp(t), want, atol=1e-7)
        np.testing.assert_allclose(interp(t), ref, atol=1e-14)
    interp = TemporalCommandInterpolator(SimpleNamespace(control_cumtime=golden["cmd/interp_a_cum_time"],
                                                         control_inputs=np.zeros((2, 10))))
    for t, i_ref in zip(golden["cmd/interp_a_elapsed"], golden["cmd/interp_a_index"]):
        assert interp._get_closet_command_index(t)[0] == i_ref


def test_workload_downsample_matches_reference(golden):
    from acmpc_amd import workloads
    for H in (20, 50, 100):
        got = workloads.reference_path_from_centreline(golden["downsample/centreline"], H)
        np.testing.assert_array_equal(got, golden["downsample/H%d" % H])
    with pytest.raises(ValueError):
        workloads.reference_path_from_centreline(golden["downsample/centreline"], 80)


def test_build_mpc_constructs_without_touching_the_gpu():
    """The reference builds its MPCs in the parent and forks (controller.py:293-297): construction must not
    create a HIP context - checked by building one here, where there may be no GPU at all."""
    from acmpc_amd.mpc import SpatialMPC, build_mpc
    mpc = build_mpc(RACING["monza"], PlaceholderVehicle())
    assert isinstance(mpc, SpatialMPC)
    assert mpc.MPC_horizon == 50 and mpc.delta_max == 0.30 and mpc.projected_control.shape == (2, 50)
    assert mpc.speed_profile_constraints is RACING["monza"]["speed_profile_constraints"]  # live dict, not a copy
    path = mpc.construct_waypoints(np.stack([np.zeros(50), np.linspace(0, 150, 50), np.full(50, 8.0)], axis=1))
    assert len(path) == 49 and np.allclose(path.psis, np.pi / 2)
    out = mpc.compute_speed_profile(path, False, end_vel=14.0)
    assert out is path and out.velocities.min() > 7.9 and hasattr(mpc, "speed_profile")


def test_particle_scorer_host_side(golden):
    """ParticleScorer construction does no device work; its host-side downsampling equals the reference's."""
    from acmpc_amd.particle_filter import ParticleScorer
    cfg = dict(n_particles=500, score_distribution=dict(mean=0, sigma=10),
               thresholds=dict(offset=10, rotation=90, minimum_particles=20, track_limit=20.0))
    scorer = ParticleScorer(cfg, dict(centre=golden["pf/centre"], left=golden["pf/left"], right=golden["pf/right"]))
    assert abs(scorer.scale - float(golden["pf/scale"])) < 1e-17
    assert abs(scorer._average_distance_between_map_points - float(golden["pf/average_map_spacing"])) < 1e-15
    obs = scorer.downsample_observations({"left": golden["pf/obs_left_raw"], "right": golden["pf/obs_right_raw"]})
    np.testing.assert_array_equal(obs[0], golden["pf/obs_left_downsampled"])
    np.testing.assert_array_equal(obs[1], golden["pf/obs_right_downsampled"])


def test_track_map_ingestion(golden, tmp_path):
    """utils/load.py:9-35 - file format, renaming and de-duplication; agent.py:137-143 - the speed window."""
    from acmpc_amd import track_map
    np.testing.assert_array_equal(track_map.remove_near_duplicate_points(golden["map/points"]), golden["map/deduplicated"])
    pts = golden["map/points"]
    raw = {"outside_track": pts + 5.0, "inside_track": pts - 5.0, "centre_track": pts}
    path = tmp_path / "synthetic_map.npy"
    np.save(path, raw, allow_pickle=True)                     # the reference's map format: a pickled dict
    loaded = track_map.load_track_map(str(path))
    assert set(loaded) == {"left", "right", "centre"}
    np.testing.assert_array_equal(loaded["centre"], golden["map/deduplicated"])
    np.testing.assert_array_equal(loaded["left"], golden["map/deduplicated"] + 5.0)
    lap = track_map.lap_reference_path(loaded["centre"])
    assert lap.shape == (len(loaded["centre"]), 3) and (lap[:, 2] == 9.5).all()
    speeds = np.arange(1000, dtype=np.float64)
    assert track_map.reference_speed_window(speeds, 500) == np.mean(np.arange(475, 575))
    assert track_map.reference_speed_window(speeds, 10) == np.mean(np.r_[985:1000, 0:85])   # wraps round the lap
    with pytest.raises(ValueError):
        track_map.load_track_map("map.csv")


def test_warm_start_shift_follows_the_elapsed_time():
    """the previous plan advanced by the elapsed time - one whole step shifts by one entry, half a
    step blends neighbours, the tail holds the last command, and a plan of another horizon is left alone."""
    from acmpc_amd.sampling_solver import ControlSolver
    cfg = dict(RACING["monza"], horizon=6)
    solver = ControlSolver(cfg, SimpleNamespace(min_u=[0, -0.1], max_u=[50, 0.1], margin=1.0, length=2.65))
    plan = np.array([[10.0, 0.00], [12.0, 0.01], [14.0, 0.02], [16.0, 0.03], [18.0, 0.04]])
    cum_time = np.array([0.0, 0.1, 0.2, 0.3, 0.4])
    solver._incumbent = plan.copy()
    solver.shift_warm_start(0.1, cum_time)
    np.testing.assert_allclose(solver._incumbent, np.vstack([plan[1:], plan[-1:]]), atol=1e-12)
    solver._incumbent = plan.copy()
    solver.shift_warm_start(0.05, cum_time)
    np.testing.assert_allclose(solver._incumbent[:4], 0.5 * (plan[:4] + plan[1:]), atol=1e-12)
    np.testing.assert_allclose(solver._incumbent[4], plan[4], atol=1e-12)
    solver._incumbent = plan.copy()
    solver.shift_warm_start(0.1, cum_time[:3])   # stale horizon: untouched
    np.testing.assert_array_equal(solver._incumbent, plan)
    solver.shift_warm_start(-1.0, cum_time)
    np.testing.assert_array_equal(solver._incumbent, plan)


PF_CONFIG = dict(n_particles=300, n_converged_particles=300,                   # configs/monza.yaml:43-66, 300 particles
                 sampling_noise=dict(x=1.1, y=1.1, yaw=3.0), control_noise=dict(velocity=0.25, yaw=2.0),
                 thresholds=dict(offset=10, rotation=90, minimum_particles=20, track_limit=20.0),
                 score_distribution=dict(mean=0, sigma=10),
                 convergence_criteria=dict(maximum_distance=50, maximum_angle=90))


def test_particle_filter_resampling_and_reset_match_the_reference(golden):
    """`ParticleFilter.resample` / `.reset` (host side, no device work) against the reference's
    `_resample_particles` / `_reset_filter` run under the same NumPy seed (localiser.py:420-545)."""
    import copy
    from acmpc_amd.particle_filter import ParticleFilter
    g = golden
    pf = ParticleFilter(PF_CONFIG, dict(centre=g["pf/centre"], left=g["pf/left"], right=g["pf/right"]))
    np.testing.assert_array_equal(pf.states, g["pf/reset_states"])          # constructed = reset
    np.testing.assert_array_equal(pf.scores, g["pf/reset_scores"])
    pf.states, pf.scores = g["pf/states"].copy(), g["pf/published_scores"].copy()
    np.random.seed(int(g["pf/resample_seed"]))
    pf.resample({"score": g["pf/score"], "valid_mask": g["pf/valid_mask"]})
    assert not pf.was_reset
    np.testing.assert_array_equal(pf.states, g["pf/resampled_states"])
    np.testing.assert_array_equal(pf.scores, g["pf/resampled_scores"])
    # own generator instead of the global stream: same law, different numbers, still the valid ones first
    cfg = copy.deepcopy(PF_CONFIG)
    cfg["thresholds"]["minimum_particles"] = 301
    pf2 = ParticleFilter(cfg, dict(centre=g["pf/centre"], left=g["pf/left"], right=g["pf/right"]),
                         rng=np.random.default_rng(0))
    pf2.states, pf2.scores = g["pf/states"].copy(), g["pf/published_scores"].copy()
    pf2.resample({"score": g["pf/score"], "valid_mask": g["pf/valid_mask"]})
    assert pf2.was_reset and not pf2.is_converged
    np.testing.assert_array_equal(pf2.states, g["pf/reset_states"])


def test_path_families_equal_the_reference_generators(golden, golden_cases):
    """workloads.family_path == control/utils.py:11-32 with the parameters of tests/test_spatial_mpc.py:45-75
    (the golden `coords` were produced by the reference's generators)."""
    from acmpc_amd import workloads
    exact = {"hairpin": np.linspace(10, 100, 7), "chicane": np.linspace(40, 100, 7),
             "curve": np.linspace(-0.02, 0.02, 7), "straight": np.linspace(40, 200, 7)}
    checked = 0
    for case in golden_cases:
        track, horizon, kind, printed = case.split("_")          # the name carries the parameter to 6 digits
        parameter = exact[kind][np.argmin(np.abs(exact[kind] - float(printed)))]
        H = int(horizon[1:])
        angle = -np.pi / 6 if kind == "hairpin" else 0.1
        got = workloads.family_path(kind, parameter, H, angle=angle)
        np.testing.assert_allclose(got[:, :2], golden[case + "/coords"][:, :2], rtol=0, atol=1e-12)
        checked += 1
    assert checked >= 28
    with pytest.raises(ValueError):
        workloads.family_path("oval", 1.0, 50)


def test_native_host_helpers_equal_their_numpy_statements(golden, golden_cases):
    """acmpc_waypoint_table / acmpc_velocity_ceiling / acmpc_unpack_decision (csrc/acmpc_host_path.cpp) against the
    NumPy statements they replaced, on the reference's path families; the NumPy forms are themselves pinned to the
    reference's golden vectors by the tests above."""
    from acmpc_amd import _capi
    from acmpc_amd.bicycle_model import SpatialBicycleModel
    from acmpc_amd.mpc import waypoint_table, waypoint_table_numpy
    from acmpc_amd.reference_path import ReferencePath
    from acmpc_amd.speed_profile import LocalisedSpeedProfileSolver, SpeedProfileSolver
    rng = np.random.default_rng(3)
    model = SpatialBicycleModel(PlaceholderVehicle(), {"min": 8.0, "max": 84.0})
    for case in golden_cases:
        coords = golden[case + "/coords"]
        table = waypoint_table(coords)
        np.testing.assert_allclose(table, waypoint_table_numpy(coords), rtol=0, atol=1e-12)
        np.testing.assert_allclose(table[:6], golden[case + "/table_no_v"][:6], rtol=0, atol=1e-12)
        n = table.shape[1]
        path = ReferencePath.from_table(golden[case + "/table"].copy())
        cons = dict(RACING["monza"]["speed_profile_constraints"])
        for cls in (SpeedProfileSolver, LocalisedSpeedProfileSolver):
            solver = cls({"control_horizon": n, "max_iterations": 4000, "constraints": cons})
            for end_velocity in (None, 14.0):
                np.testing.assert_array_equal(solver.velocity_ceiling(path, end_velocity),
                                              solver.velocity_ceiling_numpy(path, end_velocity))
        # the tail of get_control, on a random decision vector
        z = rng.normal(0, 1.0, 5 * n + 3)
        z[2:3 * (n + 1):3] = np.cumsum(rng.uniform(0.02, 0.08, n + 1))      # times increase
        pc, pred, cum, times, acc, rates = _capi.unpack_decision(z, n, path.table, model.length)
        controls = z[-2 * n:].reshape(n, 2)
        states = z[:3 * n].reshape(n, 3)
        np.testing.assert_allclose(pc, np.array([controls[:, 0], np.arctan(controls[:, 1] * model.length)]), atol=1e-15)
        np.testing.assert_allclose(pred, model.s2t(path, states)[:-1].T, rtol=0, atol=1e-12)
        np.testing.assert_array_equal(cum, states[:, 2])
        np.testing.assert_allclose(times, np.diff(states[:, 2]), rtol=0, atol=1e-15)
        np.testing.assert_allclose(acc, np.diff(states[:, 0]) / np.diff(states[:, 2]), rtol=1e-13)
        np.testing.assert_allclose(rates, np.diff(states[:, 1]) / np.diff(states[:, 2]), rtol=1e-13)
    with pytest.raises(Exception):
        _capi.waypoint_table(np.zeros((2, 3)))
    with pytest.raises(ValueError):
        _capi.unpack_decision(np.zeros(10), 49, np.zeros((7, 49)), 2.65)



def test_native_waypoint_table_equals_numpy_on_arbitrary_paths():
    """Property (hypothesis): for arbitrary finite coordinates - repeated points, reversals, huge and tiny steps -
    the library's waypoint table equals the NumPy statement: positions, spacings and widths exactly, headings to the
    last bit or two (libm's atan2 vs NumPy's), curvature accordingly - where the 1e-12-regularised division by a
    vanishing spacing amplifies that bit, only relative to its own huge magnitude."""
    from hypothesis import example, given, settings, strategies as st
    from hypothesis.extra import numpy as hnp
    from acmpc_amd.mpc import waypoint_table, waypoint_table_numpy

    finite = st.floats(min_value=-1e4, max_value=1e4, allow_nan=False, allow_infinity=False, width=64)

    @settings(max_examples=200, deadline=None, derandomize=True)   # (the same 200 paths in every run: no flaky suite)
    @given(hnp.arrays(np.float64, st.tuples(st.integers(3, 40), st.just(3)), elements=finite))
    @example(np.array([[-8145.796875, -8145.796875, -8145.796875], [-1.5, 4149.13455, -8145.796875],
                       [-8145.796875, -8145.796875, -8145.796875], [0.0, -8145.796875, -8145.796875]]))
    def check(coords):
        got, want = waypoint_table(coords), waypoint_table_numpy(coords)
        assert got.shape == want.shape == (7, coords.shape[0] - 1)
        for row in (0, 1, 4, 5, 6):
            np.testing.assert_array_equal(got[row], want[row])
        # headings: libm's atan2 here, NumPy's own (on some CPUs vectorised, 1 ulp apart) there
        np.testing.assert_allclose(got[2], want[2], rtol=1e-14, atol=1e-15)
        # curvature = wrapped heading change / spacing.  A path that doubles back EXACTLY turns by +pi or by -pi - which of the
        # two the wrap gives hangs on that last bit of atan2 - so such steps are compared by magnitude (found by this very
        # test in round 5: [-8145.8, -8145.8] -> [-1.5, 4149.1] -> back)
        turn = np.abs(want[3] - 1e-12) * (want[4] + 1e-12)
        reversal = np.abs(turn - np.pi) < 1e-9
        reversal[0] = reversal[0] or reversal[1]          # (kappa[0] is a copy of kappa[1])
        atol = 1e-3 * np.abs(want[3]).max() if np.abs(want[3]).max() > 1e9 else 1e-9
        np.testing.assert_allclose(got[3][~reversal], want[3][~reversal], rtol=1e-9, atol=atol)
        np.testing.assert_allclose(np.abs(got[3][reversal]), np.abs(want[3][reversal]), rtol=1e-9, atol=atol)
        assert got[3, 0] == got[3, 1]

    check()


def test_steer_target_against_the_reference_vector():
    """the steering target of ElTuarMPC._process_yaw (agent.py:106-115), recorded from the reference's
    own method by tests/golden/gen_agent_golden.py."""
    import os
    from acmpc_amd.command_selection import steer_target
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "agent_steer_target.npz"))
    delta_max = float(g["delta_max"])
    for yaw, current, want in zip(g["yaw"], g["current"], g["steering"]):
        # the reference returns current + PID(current, target); the recorded stand-in PID returns target - current
        assert current + (steer_target(float(yaw), delta_max) - current) == want
        assert -1.0 <= steer_target(float(yaw), delta_max) <= 1.0
    assert steer_target(delta_max, delta_max) == -1.0 and steer_target(-2 * delta_max, delta_max) == 1.0
    assert steer_target(0.0, delta_max) == 0.0


def test_mode_t_search_window_default_by_horizon():
    """`rollout_mode: "T"` without an `nn_window` key: the nearest of all waypoints (None) up to 106 steps (what the three-wave
    round's LDS holds of the verified search's frames), the (2,5) window beyond - said in the log; an explicit key - null
    included - is taken as given."""
    from acmpc_amd.sampling_solver import ControlSolver
    from test_support import RACING, PlaceholderVehicle
    from acmpc_amd.bicycle_model import SpatialBicycleModel

    def window(**extra):
        cfg = dict(RACING["monza"], rollout_mode="T", **extra)
        cons = cfg["speed_profile_constraints"]
        model = SpatialBicycleModel(PlaceholderVehicle(), {"min": cons["v_min"], "max": cons["v_max"]})
        return ControlSolver(cfg, model)._nn_window

    assert window(horizon=50) is None and window(horizon=107) is None
    assert window(horizon=108) == (2, 5) and window(horizon=129) == (2, 5)
    assert window(horizon=129, nn_window=None) is None and window(horizon=50, nn_window=[1, 2]) == (1, 2)
