"""HIP path vs the oracle through the C ABI (needs an MI355X: `pytest -m gpu`).

Bar (BASELINE.json north_star): per-candidate costs within 1e-4, selected control within 1e-5, argmin index
bit-identical.  The implementation is specified to be bit-identical to the oracle's float32 path, so the checks
below assert exact equality, which is stronger.
"""
import numpy as np
import pytest

import acmpc_oracle as orc
from test_support import engine_kwargs, full_size_controls, make_problem

pytestmark = pytest.mark.gpu

LAYOUT_CM, LAYOUT_SM = 0, 1


def _oracle(problem, coef, mode, states=False):
    cfg = problem["cfg"]
    args = (coef, problem["U"], cfg["step_cost"], cfg["r_term"], cfg["final_cost"], problem["u_lo"], problem["u_hi"],
            1.0e6)
    if mode == 0:
        return orc.rollout_spatial(problem["x0"], *args, dtype=np.float32, return_states=states)
    return orc.rollout_temporal(problem["pose0"], *args, 0.05, dtype=np.float32, return_states=states)


def _engine(problems, mode, N, n):
    from acmpc_amd import Engine
    eng = Engine(**engine_kwargs(problems[0], mode, len(problems), N, n))
    eng.set_paths(np.stack([p["table"] for p in problems]))
    return eng


def _as_layout(U, layout):
    # U [P,N,n,2] -> requested layout
    return U if layout == LAYOUT_CM else np.ascontiguousarray(U.transpose(0, 2, 3, 1))


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("layout", [LAYOUT_CM, LAYOUT_SM])
@pytest.mark.parametrize("track,H,N,P", [
    ("monza", 20, 128, 1),      # BASELINE config 1 shape
    ("monza", 50, 4096, 1),     # config 2
    ("spa", 50, 1000, 3),       # ragged N (not a multiple of 64), several problems
    ("nordschleife", 80, 520, 2),
    ("silverstone", 50, 1, 1),  # single candidate
    ("monza", 50, 67, 5),
])
def test_costs_argmin_and_winner_record(mode, layout, track, H, N, P):
    n = H - 1
    if mode == 1 and N > 1024:
        N = 1024  # the NumPy mode-T oracle is O(N n^2)
    problems = [make_problem(orc, track, H, N, seed=10 * P + p) for p in range(P)]
    eng = _engine(problems, mode, N, n)
    x0 = np.stack([p["x0"] if mode == 0 else p["pose0"] for p in problems])
    U = np.stack([p["U"] for p in problems])
    out = eng.solve(x0, _as_layout(U, layout), layout=layout)
    for p, prob in enumerate(problems):
        coef = eng.coefficients(p)
        cost, viol, X = _oracle(prob, coef, mode, states=True)[:3]
        np.testing.assert_array_equal(out["costs"][p], cost, err_msg="per-candidate costs")
        best, best_cost = orc.pick_best(cost)
        assert out["best_idx"][p] == best
        assert out["cost"][p] == np.float32(best_cost)
        assert out["violation"][p] == viol[best]
        assert out["n_feasible"][p] == np.count_nonzero(viol == 0)
        assert out["owner"][p] == 1.0
        np.testing.assert_array_equal(out["u"][p], prob["U"][best])
        np.testing.assert_array_equal(out["x"][p], X[best])


@pytest.mark.parametrize("mode", [0, 1])
def test_large_step_major_batch_uses_vector_loads(mode):
    """P*N large enough for the 4-candidates-per-lane kernel; properties that need no O(N) oracle run:
    the winner's cost equals min(costs), the record re-rolls to the same cost, and a sampled subset matches."""
    H, N, P = 50, 8192, 40
    n = H - 1
    problems = [make_problem(orc, "monza", H, 8, seed=77 + p) for p in range(P)]
    rng = np.random.default_rng(5)
    u_ref = np.stack([np.stack([p["table"][orc.ROW_V], p["table"][orc.ROW_KAPPA]], axis=1) for p in problems])
    U = (u_ref[:, None] + rng.standard_normal((P, N, n, 2)) * np.array([2.0, 0.01])).astype(np.float32)
    eng = _engine(problems, mode, N, n)
    x0 = np.stack([p["x0"] if mode == 0 else p["pose0"] for p in problems])
    out = eng.solve(x0, _as_layout(U, LAYOUT_SM), layout=LAYOUT_SM)
    out_cm = eng.solve(x0, U, layout=LAYOUT_CM)
    np.testing.assert_array_equal(out["costs"], out_cm["costs"])  # layout must not change a single bit
    np.testing.assert_array_equal(out["best_idx"], out_cm["best_idx"])
    for p in range(P):
        best, best_cost = orc.pick_best(out["costs"][p])
        assert out["best_idx"][p] == best and out["cost"][p] == np.float32(best_cost)
        np.testing.assert_array_equal(out["u"][p], U[p, best])
    sub = rng.choice(N, 256 if mode == 0 else 64, replace=False)
    for p in (0, P - 1):
        prob = dict(problems[p], U=U[p, sub])
        cost = _oracle(prob, eng.coefficients(p), mode)[0]
        np.testing.assert_array_equal(out["costs"][p, sub], cost)


@pytest.mark.parametrize("table", ["default", "lds", "scalar"])
@pytest.mark.parametrize("H", [20, 34, 50, 52, 66, 81, 82])
def test_candidate_major_rows_kernel_at_every_register_size(H, table, monkeypatch):
    """Candidate-major launches of 2 048 tiles and more take the kernel that walks its rows out of registers, in one of
    four register sizes (32 / 50 / 64 / 80 steps; 81 steps fall back to the LDS-resident tile).  Ragged N (a partial
    last tile in every problem): costs, winner and record equal the step-major layout's bit for bit, and a subset
    equals the oracle."""
    if table != "default":   # both ways the kernel can take its table rows (launch-size rule overridden)
        monkeypatch.setenv("ACMPC_TILE_TABLE", table)
    N, P = 6700, 20
    n = H - 1
    problems = [make_problem(orc, "monza", H, 8, seed=300 + p) for p in range(P)]
    rng = np.random.default_rng(H)
    u_ref = np.stack([np.stack([p["table"][orc.ROW_V], p["table"][orc.ROW_KAPPA]], axis=1) for p in problems])
    U = (u_ref[:, None] + rng.standard_normal((P, N, n, 2)) * np.array([2.0, 0.01])).astype(np.float32)
    eng = _engine(problems, 0, N, n)
    x0 = np.stack([p["x0"] for p in problems])
    out_sm = eng.solve(x0, _as_layout(U, LAYOUT_SM), layout=LAYOUT_SM)
    out = eng.solve(x0, U, layout=LAYOUT_CM)
    np.testing.assert_array_equal(out["costs"], out_sm["costs"])
    np.testing.assert_array_equal(out["best_idx"], out_sm["best_idx"])
    np.testing.assert_array_equal(out["records"], out_sm["records"])
    sub = np.concatenate([rng.choice(N, 96, replace=False), np.arange(N - 48, N)])   # (the partial tile included)
    for p in (0, P - 1):
        cost = _oracle(dict(problems[p], U=U[p, sub]), eng.coefficients(p), 0)[0]
        np.testing.assert_array_equal(out["costs"][p, sub], cost)
        best, best_cost = orc.pick_best(out["costs"][p])
        assert out["best_idx"][p] == best and out["cost"][p] == np.float32(best_cost)
    eng.close()


@pytest.mark.parametrize("H", [50, 34])
def test_candidate_major_rows_kernel_single_problem_odd_last_tile(H):
    """One problem of 131 109 candidates: 2 049 tiles, the last of 37 rows - with an odd step count its span ends on a
    lone (v, kappa) pair past the last 16-byte piece (a path only a single-problem launch can reach: with several
    problems an odd N puts every second span off the 16-byte boundary and the LDS-resident kernel runs instead)."""
    N, n = 131072 + 37, H - 1
    prob = make_problem(orc, "spa", H, 8, seed=77)
    rng = np.random.default_rng(3)
    u_ref = np.stack([prob["table"][orc.ROW_V], prob["table"][orc.ROW_KAPPA]], axis=1)
    U = (u_ref[None, None] + rng.standard_normal((1, N, n, 2)) * np.array([2.0, 0.01])).astype(np.float32)
    eng = _engine([prob], 0, N, n)
    x0 = prob["x0"][None]
    out_sm = eng.solve(x0, _as_layout(U, LAYOUT_SM), layout=LAYOUT_SM)
    out = eng.solve(x0, U, layout=LAYOUT_CM)
    np.testing.assert_array_equal(out["costs"], out_sm["costs"])
    np.testing.assert_array_equal(out["records"], out_sm["records"])
    sub = np.concatenate([rng.choice(N, 64, replace=False), np.arange(N - 40, N)])
    cost = _oracle(dict(prob, U=U[0, sub]), eng.coefficients(0), 0)[0]
    np.testing.assert_array_equal(out["costs"][0, sub], cost)
    eng.close()


def test_candidate_major_matrix_off_the_16_byte_boundary():
    """A device control matrix that is a view 8 bytes into a caller's buffer (candidate-major, a launch large enough
    for the rows kernel, which moves 16-byte pieces): the same costs and records as the aligned copy."""
    import torch
    from acmpc_amd import _capi
    H, N, P = 50, 6700, 20
    n = H - 1
    problems = [make_problem(orc, "monza", H, 8, seed=500 + p) for p in range(P)]
    rng = np.random.default_rng(21)
    u_ref = np.stack([np.stack([p["table"][orc.ROW_V], p["table"][orc.ROW_KAPPA]], axis=1) for p in problems])
    U_h = (u_ref[:, None] + rng.standard_normal((P, N, n, 2)) * np.array([2.0, 0.01])).astype(np.float32)
    eng = _engine(problems, 0, N, n)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    x0 = torch.tensor(np.stack([p["x0"] for p in problems]), device=dev)
    flat = torch.zeros(U_h.size + 6, dtype=torch.float32, device=dev)
    results = []
    for shift in (0, 2, 4):   # floats: 0 and 16 bytes keep the boundary, 8 bytes does not
        view = flat[shift:shift + U_h.size]
        view.copy_(torch.tensor(U_h.reshape(-1)))
        assert (view.data_ptr() % 16 == 0) == (shift != 2)
        costs = torch.empty(P, N, device=dev)
        keys = torch.empty(P, dtype=torch.int64, device=dev)
        rec = torch.empty(P, _capi.record_floats(n), device=dev)
        eng.solve_device(x0.data_ptr(), view.data_ptr(), P, N, n, LAYOUT_CM, costs.data_ptr(), keys.data_ptr(),
                         rec.data_ptr(), stream)
        torch.cuda.synchronize()
        results.append((costs.cpu().numpy(), keys.cpu().numpy(), rec.cpu().numpy()))
    for got in results[1:]:
        for a, b in zip(results[0], got):
            np.testing.assert_array_equal(a, b)
    cost = _oracle(dict(problems[3], U=U_h[3, :64]), eng.coefficients(3), 0)[0]
    np.testing.assert_array_equal(results[1][0][3, :64], cost)
    eng.close()


def test_nonfinite_costs_rank_last():
    prob = make_problem(orc, "monza", 20, 64, seed=3)
    prob["U"][5, 3, 0] = np.nan
    prob["U"][9, 0, 1] = np.inf
    eng = _engine([prob], 0, 64, 19)
    out = eng.solve(prob["x0"][None], prob["U"][None])
    cost = _oracle(prob, eng.coefficients(0), 0)[0]
    assert not np.isfinite(out["costs"][0, 5]) and not np.isfinite(out["costs"][0, 9])
    assert out["best_idx"][0] == orc.pick_best(cost)[0]
    assert out["best_idx"][0] not in (5, 9)


def test_capacity_and_state_errors():
    from acmpc_amd import Engine, EngineError
    prob = make_problem(orc, "monza", 20, 16, seed=1)
    eng = Engine(**engine_kwargs(prob, 0, 1, 16, 19))
    with pytest.raises(EngineError) as e:
        eng.solve(prob["x0"][None], prob["U"][None])
    assert e.value.code == -5  # ACMPC_ESTATE: no tables yet
    eng.set_paths(prob["table"])
    with pytest.raises(EngineError) as e:
        eng.solve(prob["x0"][None], np.zeros((1, 32, 19, 2), np.float32))
    assert e.value.code == -4  # ACMPC_ECAPACITY


@pytest.mark.parametrize("layout", [LAYOUT_CM, LAYOUT_SM])
@pytest.mark.parametrize("window", [(2, 5), (1, 2), (0, 2)])
def test_mode_t_windowed_nearest_search(layout, window):
    """Mode T with nn_window: bit-identical to the oracle's windowed spec (also where the window is too short to
    find the global nearest waypoint - the spec, not the outcome, is what must match).  Windows of 8 and 4 waypoints
    take the unrolled packed search, 3 the generic loop; candidates whose state overflows or turns NaN must still
    pick the same waypoints as the `d < best` scan of the oracle."""
    from acmpc_amd import Engine
    H, N, P = 50, 777, 2
    n = H - 1
    problems = [make_problem(orc, "monza", H, N, seed=300 + p) for p in range(P)]
    problems[0]["U"][5, 3, 0] = np.nan
    problems[0]["U"][9, 0, 1] = np.inf
    problems[0]["U"][11, 10, 0] = 3.0e38     # position overflows to inf two steps later
    problems[1]["U"][7, 20, 0] = -1.0e30
    eng = Engine(**engine_kwargs(problems[0], 1, P, N, n, nn_window=window))
    eng.set_paths(np.stack([p["table"] for p in problems]))
    x0 = np.stack([p["pose0"] for p in problems])
    U = np.stack([p["U"] for p in problems])
    out = eng.solve(x0, _as_layout(U, layout), layout=layout)
    for p, prob in enumerate(problems):
        cfg = prob["cfg"]
        cost, viol, S, _ = orc.rollout_temporal(prob["pose0"], eng.coefficients(p), prob["U"], cfg["step_cost"],
                                                cfg["r_term"], cfg["final_cost"], prob["u_lo"], prob["u_hi"], 1.0e6,
                                                0.05, dtype=np.float32, return_states=True, nn_window=window)
        np.testing.assert_array_equal(out["costs"][p], cost)
        best = orc.pick_best(cost)[0]
        assert out["best_idx"][p] == best and out["violation"][p] == viol[best]
        np.testing.assert_array_equal(out["x"][p], S[best])


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("H,N,P", [(100, 300, 2), (3, 130, 1), (66, 129, 3)])
def test_other_horizons(mode, H, N, P):
    """The reference's mapping horizon (100, configs/monza.yaml:31), the shortest legal one (n = 2) and one with
    more than 64 steps and an odd count (n = 65: the finalize kernel's second 64-step chunk holds one step)."""
    n = H - 1
    problems = [make_problem(orc, "monza", H, N, seed=900 + 7 * H + p) for p in range(P)]
    eng = _engine(problems, mode, N, n)
    x0 = np.stack([p["x0"] if mode == 0 else p["pose0"] for p in problems])
    U = np.stack([p["U"] for p in problems])
    for layout in (LAYOUT_CM, LAYOUT_SM):
        out = eng.solve(x0, _as_layout(U, layout), layout=layout)
        for p, prob in enumerate(problems):
            cost, viol, X = _oracle(prob, eng.coefficients(p), mode, states=True)[:3]
            np.testing.assert_array_equal(out["costs"][p], cost)
            best = orc.pick_best(cost)[0]
            assert out["best_idx"][p] == best and out["violation"][p] == viol[best]
            np.testing.assert_array_equal(out["x"][p], X[best])
            np.testing.assert_array_equal(out["u"][p], prob["U"][best])


@pytest.mark.parametrize("mode,window", [(0, None), (1, None), (1, (2, 5))])
def test_maximum_horizon(mode, window):
    """n = 1024 control steps, the largest the handle accepts (the finalize's record image and, in mode T, the
    waypoint table still fit the 64 KB of LDS a workgroup may have): costs, argmin and the winner's 1025 states."""
    from acmpc_amd import Engine, EngineError
    H, N = 1025, 130
    n = H - 1
    prob = make_problem(orc, "monza", H, N, seed=77)
    eng = Engine(**engine_kwargs(prob, mode, 1, N, n, nn_window=window))
    eng.set_paths(prob["table"])
    x0 = prob["x0"] if mode == 0 else prob["pose0"]
    cfg = prob["cfg"]
    args = (prob["U"], cfg["step_cost"], cfg["r_term"], cfg["final_cost"], prob["u_lo"], prob["u_hi"], 1.0e6)
    if mode == 0:
        cost, viol, X = orc.rollout_spatial(x0, eng.coefficients(0), *args, dtype=np.float32, return_states=True)
    else:
        cost, viol, X, _ = orc.rollout_temporal(x0, eng.coefficients(0), *args, 0.05, dtype=np.float32,
                                                return_states=True, nn_window=window)
    for layout in (LAYOUT_CM, LAYOUT_SM):
        out = eng.solve(x0[None], _as_layout(prob["U"][None], layout), layout=layout)
        np.testing.assert_array_equal(out["costs"][0], cost)
        best = orc.pick_best(cost)[0]
        assert out["best_idx"][0] == best and out["violation"][0] == viol[best]
        np.testing.assert_array_equal(out["x"][0], X[best])
    with pytest.raises(EngineError):
        Engine(**engine_kwargs(prob, mode, 1, N, n + 1))


@pytest.mark.parametrize("layout", [LAYOUT_CM, LAYOUT_SM])
def test_verified_window_search_is_the_exhaustive_search(layout, monkeypatch):
    """Mode T with the exhaustive specification runs as a verified window search (acmpc_device.h: nearest_verified).
    On a path that folds back on itself - waypoints far apart in index are close in space, so the certification
    thresholds are small and many candidates need the wave-wide fallback - with candidates that wander off their
    stretch, costs must still equal the oracle's scan over ALL waypoints bit for bit, and the plain-scan
    kernel's (ACMPC_NO_VERIFIED_SEARCH)."""
    from acmpc_amd import Engine
    H, N, P = 50, 1111, 2
    n = H - 1
    problems = []
    for p, gap in enumerate((6.0, 9.0)):
        prob = make_problem(orc, "monza", H, N, seed=640 + p, sigma=(6.0, 0.05))
        # out along +y for 69 m, a tight turn, and back `gap` metres to the right: a candidate that drifts right is
        # nearer to the return leg - forty waypoints further on - than to the stretch it is driving along
        up = [(0.0, 3.0 * i) for i in range(24)]
        turn = [(gap / 2 + gap / 2 * np.cos(a), 69.0 + gap / 2 * np.sin(a)) for a in np.radians([135.0, 90.0, 45.0])]
        down = [(gap, 69.0 - 3.0 * i) for i in range(23)]
        xy = np.array(up + turn + down)
        table = orc.construct_waypoints(np.column_stack([xy, np.full(H, 9.5)]))
        table[orc.ROW_V] = prob["table"][orc.ROW_V]
        prob["table"] = table
        problems.append(prob)
    results = []
    for plain in (False, True):
        if plain:
            monkeypatch.setenv("ACMPC_NO_VERIFIED_SEARCH", "1")
        eng = Engine(**engine_kwargs(problems[0], 1, P, N, n))
        eng.set_paths(np.stack([p["table"] for p in problems]))
        x0 = np.stack([p["pose0"] for p in problems])
        U = np.stack([p["U"] for p in problems])
        results.append((eng.solve(x0, _as_layout(U, layout), layout=layout), [eng.coefficients(p) for p in range(P)]))
    (out, coefs), (out_plain, _) = results
    np.testing.assert_array_equal(out["costs"], out_plain["costs"])
    np.testing.assert_array_equal(out["records"], out_plain["records"])
    for p, prob in enumerate(problems):
        cfg = prob["cfg"]
        cost, viol, S, J = orc.rollout_temporal(prob["pose0"], coefs[p], prob["U"], cfg["step_cost"], cfg["r_term"],
                                                cfg["final_cost"], prob["u_lo"], prob["u_hi"], 1.0e6, 0.05,
                                                dtype=np.float32, return_states=True)
        np.testing.assert_array_equal(out["costs"][p], cost)
        assert out["best_idx"][p] == orc.pick_best(cost)[0]
        # the input really is the hard case: some rollouts jump between far-apart stretches of the path
        assert np.abs(np.diff(J, axis=1)).max() > 8


@pytest.mark.parametrize("name", ["loop", "spiral", "figure_eight", "random_walk", "stall", "far_from_origin", "fold_2"])
def test_verified_search_on_paths_that_defeat_its_certificate(name, monkeypatch):
    """The geometries tests/test_search_frames.py checks the certificate on - paths that close on themselves, cross
    themselves, stall on one spot, lie kilometres from the origin - rolled on the GPU with candidates that wander:
    exhaustive semantics through the verified window search = the plain scan of all waypoints = the oracle."""
    from acmpc_amd import Engine
    from test_search_frames import _paths
    H, N = 50, 640
    n = H - 1
    prob = make_problem(orc, "monza", H, N, seed=77, sigma=(6.0, 0.05))
    xy = _paths()[name]
    assert xy.shape[0] == H
    table = orc.construct_waypoints(np.column_stack([xy, np.full(H, 9.5)]))
    table[orc.ROW_V] = prob["table"][orc.ROW_V]
    heading = np.arctan2(xy[1, 1] - xy[0, 1], xy[1, 0] - xy[0, 0])
    pose0 = np.array([[xy[0, 0], xy[0, 1], heading]], dtype=np.float32)   # on the path's first waypoint, along it
    outs = []
    for plain in (False, True):
        if plain:
            monkeypatch.setenv("ACMPC_NO_VERIFIED_SEARCH", "1")
        eng = Engine(**engine_kwargs(prob, 1, 1, N, n))
        eng.set_paths(table[None])
        outs.append((eng.solve(pose0, _as_layout(prob["U"][None], LAYOUT_SM), layout=LAYOUT_SM), eng.coefficients(0)))
        eng.close()
    (out, coef), (out_plain, _) = outs
    np.testing.assert_array_equal(out["costs"], out_plain["costs"])
    np.testing.assert_array_equal(out["records"], out_plain["records"])
    cfg = prob["cfg"]
    cost = orc.rollout_temporal(pose0[0], coef, prob["U"], cfg["step_cost"], cfg["r_term"], cfg["final_cost"],
                                prob["u_lo"], prob["u_hi"], 1.0e6, 0.05, dtype=np.float32)[0]
    np.testing.assert_array_equal(out["costs"][0], cost)


@pytest.mark.parametrize("H", [9, 10, 13, 65, 66, 101, 140])
def test_verified_search_at_other_horizons(H, monkeypatch):
    """Horizons round the verified search's limits: n = 8 (the window is the whole path), just above it, n = 64 / 65
    (the wave-wide fallback holds one waypoint per lane up to 64, several beyond), the mapping controller's n = 100 and
    a longer one - on a path that turns back on itself, so that the fallback is taken."""
    from acmpc_amd import Engine
    N = 448
    n = H - 1
    prob = make_problem(orc, "monza", H, N, seed=300 + H, sigma=(6.0, 0.05))
    half = H // 2
    gap = 5.0
    up = [(0.0, 2.0 * i) for i in range(half)]
    down = [(gap, 2.0 * (half - 1) - 2.0 * i) for i in range(H - half)]
    xy = np.array(up + down)
    table = orc.construct_waypoints(np.column_stack([xy, np.full(H, 9.5)]))
    table[orc.ROW_V] = prob["table"][orc.ROW_V]
    pose0 = np.array([[0.3, 0.0, np.pi / 2]], dtype=np.float32)
    outs = []
    for plain in (False, True):
        if plain:
            monkeypatch.setenv("ACMPC_NO_VERIFIED_SEARCH", "1")
        eng = Engine(**engine_kwargs(prob, 1, 1, N, n))
        eng.set_paths(table[None])
        outs.append((eng.solve(pose0, _as_layout(prob["U"][None], LAYOUT_SM), layout=LAYOUT_SM), eng.coefficients(0)))
        eng.close()
    (out, coef), (out_plain, _) = outs
    np.testing.assert_array_equal(out["costs"], out_plain["costs"])
    np.testing.assert_array_equal(out["records"], out_plain["records"])
    cfg = prob["cfg"]
    cost, _, _, J = orc.rollout_temporal(pose0[0], coef, prob["U"], cfg["step_cost"], cfg["r_term"], cfg["final_cost"],
                                         prob["u_lo"], prob["u_hi"], 1.0e6, 0.05, dtype=np.float32, return_states=True)
    np.testing.assert_array_equal(out["costs"][0], cost)
    if H >= 65:
        assert np.abs(np.diff(J, axis=1)).max() > 4   # rollouts do jump to the return leg


def test_every_launch_shape_gives_the_same_bits(monkeypatch):
    """ACMPC_SHAPE forces the workgroup size / candidates per lane; results must not depend on it."""
    H, N, P = 50, 2048, 2
    n = H - 1
    problems = [make_problem(orc, "spa", H, N, seed=1234 + p) for p in range(P)]
    x0 = np.stack([p["x0"] for p in problems])
    U = _as_layout(np.stack([p["U"] for p in problems]), LAYOUT_SM)
    results = []
    for shape in ("64,1", "256,1", "256,2", "256,4"):
        monkeypatch.setenv("ACMPC_SHAPE", shape)
        eng = _engine(problems, 0, N, n)
        results.append(eng.solve(x0, U, layout=LAYOUT_SM))
    for other in results[1:]:
        np.testing.assert_array_equal(other["costs"], results[0]["costs"])
        np.testing.assert_array_equal(other["records"], results[0]["records"])


@pytest.mark.parametrize("track,H,N,mode,window", [
    ("spa", 50, 65536, 0, None),            # BASELINE config 3 at full size
    ("nordschleife", 80, 262144, 0, None),  # BASELINE config 4 at full size (all candidates on one GPU)
    ("spa", 50, 65536, 1, (2, 5)),          # config 3, Cartesian mode with the windowed search
    ("spa", 50, 65536, 1, None),            # config 3, mode T's DEFAULT search: exhaustive semantics (verified window
    ("nordschleife", 80, 262144, 1, None),  #   + wave-cooperative fallback) against the C oracle's scan of all n
    ("nordschleife", 80, 262144, 1, (2, 5)),
])
def test_baseline_configs_at_full_size_against_the_c_oracle(track, H, N, mode, window):
    """Every one of the N costs bit-identical to the oracle's C restatement (the NumPy oracle would take minutes at
    these sizes; C == NumPy is established in tests/test_oracle_c_vs_numpy.py), plus argmin and winner record."""
    import c_oracle
    from acmpc_amd import Engine
    n = H - 1
    prob = make_problem(orc, track, H, 16, seed=4242)
    U = full_size_controls(orc, prob, N, n)
    eng = Engine(**engine_kwargs(prob, mode, 1, N, n, nn_window=window))
    eng.set_paths(prob["table"])
    cfg = prob["cfg"]
    w = c_oracle.make_weights(cfg["step_cost"], cfg["r_term"], cfg["final_cost"], prob["u_lo"], prob["u_hi"], 1.0e6,
                              nn_window=window)
    x0 = prob["x0"] if mode == 0 else prob["pose0"]
    want_cost, want_viol, want_states = c_oracle.rollout(mode, x0, eng.coefficients(0), U, 0, w, return_states=True)
    best = c_oracle.argmin(want_cost)
    U_sm = np.ascontiguousarray(U.transpose(1, 2, 0))
    for layout, data in ((LAYOUT_SM, U_sm), (LAYOUT_CM, U)):
        out = eng.solve(x0[None], data[None], layout=layout)
        np.testing.assert_array_equal(out["costs"][0], want_cost)
        assert out["best_idx"][0] == best and out["cost"][0] == want_cost[best]
        assert out["n_feasible"][0] == np.count_nonzero(want_viol == 0)
        np.testing.assert_array_equal(out["x"][0], want_states[best])
        np.testing.assert_array_equal(out["u"][0], U[best])


def test_empty_and_oversized_inputs_are_rejected():
    from acmpc_amd import Engine, EngineError
    prob = make_problem(orc, "monza", 20, 8, seed=2)
    eng = Engine(**engine_kwargs(prob, 0, 1, 8, 19))
    eng.set_paths(prob["table"])
    with pytest.raises(EngineError) as e:
        eng.solve(prob["x0"][None], np.zeros((1, 0, 19, 2), np.float32))   # no candidates
    assert e.value.code == -1
    with pytest.raises(EngineError) as e:
        eng.solve(prob["x0"][None], np.zeros((1, 8, 18, 2), np.float32))   # horizon differs from the tables
    assert e.value.code == -1
    with pytest.raises(EngineError):
        eng.set_paths(np.zeros((1, 7, 1)))                                  # a one-point path has no dynamics


def test_device_allocation_failure_is_an_error_not_a_crash():
    """A capacity no GPU can hold: the first device call reports the failed hipMalloc (ACMPC_EHIP, with the HIP
    message) and the process carries on - a handle of sensible size works afterwards."""
    import torch
    from acmpc_amd import Engine, EngineError
    prob = make_problem(orc, "monza", 20, 64, seed=5)
    huge = Engine(**engine_kwargs(prob, 0, 65535, 2**31 - 1, 19))      # petabytes of partial keys
    huge.set_paths(np.stack([prob["table"]] * 2))
    x0 = torch.zeros(2, 3, device="cuda")
    U = torch.zeros(2, 19, 2, 64, device="cuda")
    with pytest.raises(EngineError) as e:
        huge.rollout_device(x0.data_ptr(), U.data_ptr(), 2, 64, 19, LAYOUT_SM, 0, 0, 0, 0)
    assert e.value.code == -2 and "memory" in str(e.value).lower()
    # a retry on the same handle fails the same way without allocating again what the first attempt already obtained,
    # and closing the handle gives everything back (the failed bring-up used to leak)
    free_after_first = torch.cuda.mem_get_info()[0]
    for _ in range(3):
        with pytest.raises(EngineError):
            huge.rollout_device(x0.data_ptr(), U.data_ptr(), 2, 64, 19, LAYOUT_SM, 0, 0, 0, 0)
    assert abs(torch.cuda.mem_get_info()[0] - free_after_first) < (8 << 20)
    huge.close()
    assert torch.cuda.mem_get_info()[0] >= free_after_first
    eng = _engine([prob], 0, 64, 19)
    out = eng.solve(prob["x0"][None], prob["U"][None])
    cost = _oracle(prob, eng.coefficients(0), 0)[0]
    np.testing.assert_array_equal(out["costs"][0], cost)


def test_bad_device_ordinal_and_closed_handle_are_errors():
    from acmpc_amd import Engine, EngineError
    prob = make_problem(orc, "monza", 20, 16, seed=6)
    eng = Engine(**engine_kwargs(prob, 0, 1, 16, 19, device=99))      # construction does no device work
    eng.set_paths(prob["table"])
    with pytest.raises(EngineError) as e:
        eng.solve(prob["x0"][None], prob["U"][None])
    assert e.value.code in (-2, -3)
    eng.close()
    eng.close()                                                        # idempotent
    with pytest.raises(EngineError):
        eng.solve(prob["x0"][None], prob["U"][None])                   # a closed handle is refused, not dereferenced
    good = _engine([prob], 0, 16, 19)
    assert good.solve(prob["x0"][None], prob["U"][None])["owner"][0] == 1.0



@pytest.mark.parametrize("window", [None, (2, 5)])
def test_a_one_generation_mode_t_launch_starts_its_workgroups_together(window):
    """1 M candidates = 2 048 workgroups = eight waves on every SIMD: all of them fit from the first microsecond.  Round 4's
    launch of the 8-waypoint window (an instantiation of its own) had 12 to 60 of them wait 65 us - half the launch - for a
    compute unit while others ran seven (tools/modeT_stamps.py, profiles/r05_mode_T_timeline.json); the launcher now takes ONE
    instantiation for every search, which the dispatcher deals evenly.  The library's own diagnostic
    (ACMPC_START_CLOCKS / acmpc_rollout_start_clocks) holds it there."""
    import torch
    from acmpc_amd import Engine, workloads
    P, N, H = 256, 4096, 50
    n = H - 1
    batch = workloads.problem_batch("monza", P, H, seed=0)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    eng = Engine(**workloads.engine_kwargs(batch, 1, N, device=0, nn_window=window))
    eng.set_paths(batch.tables)
    eng.sync_tables(stream)
    u_ref = torch.tensor(np.stack([batch.tables[:, 6, :], batch.tables[:, 3, :]], axis=2), dtype=torch.float32, device=dev).contiguous()
    x0 = torch.tensor(batch.pose0, device=dev)
    U = torch.empty((P, n, 2, N), device=dev)
    costs = torch.empty(P, N, device=dev)
    eng.sample_device(u_ref.data_ptr(), 2 * n, u_ref.data_ptr(), P, N, n, 1, 0, (2.0, 0.01), 77, 0, U.data_ptr(), stream)
    assert eng.rollout_start_clocks().size == 0          # nothing stamped without the option
    eng.set_option("ACMPC_START_CLOCKS", "1")
    late = []
    for _ in range(6):
        eng.rollout_device(x0.data_ptr(), U.data_ptr(), P, N, n, 1, 0, costs.data_ptr(), 0, stream)
        starts = eng.rollout_start_clocks()
        assert starts.shape == (2048,) and starts.min() == 0.0
        late.append(int((starts > 10.0).sum()))
    eng.close()
    assert max(late[1:]) <= 4, late        # (the first launch of a handle also loads the code object)
