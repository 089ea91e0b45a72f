"""Randomised shapes: mode, layout, horizon, candidate count, problem count, search window and path family drawn
from a seeded generator; every per-candidate cost, the argmin and the winner record must equal the oracle's C
restatement bit for bit (C == NumPy is held by tests/test_oracle_c_vs_numpy.py).  ACMPC_FUZZ_CASES raises the count
for a longer soak (default 40; 400 cases were run when this test was written)."""
import os

import numpy as np
import pytest

import acmpc_oracle as orc
import c_oracle
from test_support import engine_kwargs, make_problem

pytestmark = pytest.mark.gpu


def _case(rng):
    mode = int(rng.integers(0, 2))
    layout = int(rng.integers(0, 2))
    H = int(rng.choice([3, 4, 9, 17, 20, 33, 50, 65, 66, 80, 100, 130]))
    N = int(rng.choice([1, 2, 63, 64, 65, 127, 256, 257, 1000, 1024, 2049, 4096]))
    P = int(rng.integers(1, 5))
    window = None
    if mode == 1:
        window = [None, (2, 5), (1, 2), (0, 2), (3, 12)][int(rng.integers(0, 5))]
    track = ["monza", "spa", "nordschleife", "silverstone"][int(rng.integers(0, 4))]
    sigma = [(2.0, 0.01), (6.0, 0.05), (0.5, 0.002)][int(rng.integers(0, 3))]
    return mode, layout, H, N, P, window, track, sigma


def test_random_shapes_against_the_c_oracle():
    from acmpc_amd import Engine
    cases = int(os.environ.get("ACMPC_FUZZ_CASES", "40"))
    rng = np.random.default_rng(20260917)
    for index in range(cases):
        mode, layout, H, N, P, window, track, sigma = _case(rng)
        n = H - 1
        label = "case %d: mode %d layout %d H %d N %d P %d window %s %s sigma %s" % (index, mode, layout, H, N, P, window,
                                                                                   track, sigma)
        problems = [make_problem(orc, track, H, N, seed=9000 + 10 * index + p, sigma=sigma) for p in range(P)]
        if index % 7 == 3 and N > 8:                      # some non-finite candidates
            problems[0]["U"][5, n // 2, 0] = np.nan
            problems[0]["U"][7, 0, 1] = np.inf
        eng = Engine(**engine_kwargs(problems[0], mode, P, N, n, nn_window=window))
        eng.set_paths(np.stack([p["table"] for p in problems]))
        x0 = np.stack([p["x0"] if mode == 0 else p["pose0"] for p in problems])
        U = np.stack([p["U"] for p in problems])
        data = U if layout == 0 else np.ascontiguousarray(U.transpose(0, 2, 3, 1))
        out = eng.solve(x0, data, layout=layout)
        for p, prob in enumerate(problems):
            cfg = prob["cfg"]
            w = c_oracle.make_weights(cfg["step_cost"], cfg["r_term"], cfg["final_cost"], prob["u_lo"], prob["u_hi"],
                                      1.0e6, nn_window=window)
            cost, viol, states = c_oracle.rollout(mode, x0[p], eng.coefficients(p), prob["U"], 0, w, return_states=True)
            np.testing.assert_array_equal(out["costs"][p], cost, err_msg=label)
            best = c_oracle.argmin(cost)
            assert out["best_idx"][p] == best, label
            assert out["n_feasible"][p] == np.count_nonzero(viol == 0), label
            if np.isfinite(cost[best]):
                np.testing.assert_array_equal(out["x"][p], states[best], err_msg=label)
                np.testing.assert_array_equal(out["u"][p], prob["U"][best], err_msg=label)
        eng.close()


def test_random_optimise_shapes_against_the_manual_round_loop():
    """acmpc_optimize (fused sample + rollout + finalize launches, hipGraph) == the loop of sample_device and
    solve_device calls, for random modes, windows, horizons, candidate counts, problem counts and round counts."""
    import torch
    from acmpc_amd import Engine, _capi
    cases = max(20, int(os.environ.get("ACMPC_FUZZ_CASES", "40")))
    rng = np.random.default_rng(777)
    dev = torch.device("cuda", 0)
    s = torch.cuda.current_stream().cuda_stream
    for index in range(cases):
        mode = int(rng.integers(0, 2))
        window = [None, (2, 5), (1, 2), (0, 2)][int(rng.integers(0, 4))] if mode == 1 else None
        H = int(rng.choice([3, 9, 20, 50, 66, 100]))
        N = int(rng.choice([1, 63, 64, 65, 500, 2048, 16384]))
        P = int(rng.integers(1, 4))
        rounds = int(rng.integers(1, 5))
        n = H - 1
        label = "case %d: mode %d window %s H %d N %d P %d rounds %d" % (index, mode, window, H, N, P, rounds)
        problems = [make_problem(orc, "silverstone", H, 4, seed=12000 + 10 * index + p) for p in range(P)]
        eng = Engine(**engine_kwargs(problems[0], mode, P, N, n, nn_window=window))
        eng.set_paths(np.stack([p["table"] for p in problems]))
        u_ref = np.stack([np.stack([p["table"][orc.ROW_V], p["table"][orc.ROW_KAPPA]], axis=1)
                          for p in problems]).astype(np.float32)
        x0 = np.stack([p["x0"] if mode == 0 else p["pose0"] for p in problems])
        sigma, seed = (3.0, 0.01), 100 + index
        out = eng.optimize(x0, u_ref, u_ref, N, rounds, sigma, shrink=0.5, seed=seed)
        R = _capi.record_floats(n)
        d_x0, d_ref = torch.tensor(x0, device=dev), torch.tensor(u_ref, device=dev)
        U = torch.empty(P, n, 2, N, device=dev)
        rec = torch.empty(P, R, device=dev)
        keys = torch.empty(P, dtype=torch.int64, device=dev)
        for r in range(rounds):
            centre_ptr, stride = (d_ref.data_ptr(), 2 * n) if r == 0 else (rec.data_ptr() + 4 * _capi.REC_HEADER, R)
            eng.sample_device(centre_ptr, stride, d_ref.data_ptr(), P, N, n, 1, 0, (sigma[0] * 0.5**r, sigma[1] * 0.5**r),
                              seed, r, U.data_ptr(), s)
            eng.solve_device(d_x0.data_ptr(), U.data_ptr(), P, N, n, 1, 0, keys.data_ptr(), rec.data_ptr(), s)
            torch.cuda.synchronize()
        np.testing.assert_array_equal(out["records"], rec.cpu().numpy(), err_msg=label)
        eng.close()

