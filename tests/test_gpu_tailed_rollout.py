"""The batched solve's finalize in its three forms (round 4): a wavefront per problem (finalize_kernel), a LANE per problem
from 256 problems up (finalize_lanes_kernel, the default there), and inside the rollout's launch (rollout_tailed_kernel: the
last workgroup of every problem takes the argmin and re-rolls the winner while other problems are still streaming).  The
records must be
the two-launch form's (rollout_kernel + finalize_kernel, the default: it is the faster of the two on the headline's batch,
csrc/acmpc_capi.hip solve_batched; ACMPC_TAILED_ROLLOUT=1 selects the one launch) bit for bit, and the oracle's."""
import numpy as np
import pytest

import acmpc_oracle as orc
from test_support import engine_kwargs, make_problem

pytestmark = pytest.mark.gpu


def _solve(monkeypatch, switch, problems, P, N, n, sampled):
    import torch
    from acmpc_amd import Engine, _capi
    for name in ("ACMPC_TAILED_ROLLOUT", "ACMPC_NO_LANE_FINALIZE"):
        monkeypatch.delenv(name, raising=False)
    if switch:
        monkeypatch.setenv(switch, "1")
    dev = torch.device("cuda", 0)
    eng = Engine(**engine_kwargs(problems[0], 0, P, N, n))      # (the switch is read when the handle is created)
    eng.set_paths(np.stack([p["table"] for p in problems]))
    s = torch.cuda.current_stream().cuda_stream
    x0 = torch.tensor(np.stack([p["x0"] for p in problems]), device=dev)
    u_ref = torch.tensor(np.stack([np.stack([p["table"][orc.ROW_V], p["table"][orc.ROW_KAPPA]], axis=1) for p in problems]),
                         dtype=torch.float32, device=dev).contiguous()
    U = torch.empty(P, n, 2, N, device=dev)
    sigma, seed, rnd = (2.0, 0.01), 4321, 1
    eng.sample_device(u_ref.data_ptr(), 2 * n, u_ref.data_ptr(), P, N, n, 1, 0, sigma, seed, rnd, U.data_ptr(), s)
    costs = torch.empty(P, N, device=dev)
    keys = torch.empty(P, dtype=torch.int64, device=dev)
    rec = torch.empty(P, _capi.record_floats(n), device=dev)
    outs = []
    for _ in range(3):   # repeated launches: the tickets are left as they were found
        rec.zero_()
        if sampled:
            eng.solve_sampled_device(x0.data_ptr(), U.data_ptr(), u_ref.data_ptr(), 2 * n, u_ref.data_ptr(), P, N, n, 1,
                                     sigma, seed, rnd, costs.data_ptr(), keys.data_ptr(), rec.data_ptr(), s)
        else:
            eng.solve_device(x0.data_ptr(), U.data_ptr(), P, N, n, 1, costs.data_ptr(), keys.data_ptr(), rec.data_ptr(), s)
        torch.cuda.synchronize()
        outs.append((rec.cpu().numpy().copy(), keys.cpu().numpy().copy(), costs.cpu().numpy().copy()))
    eng.close()
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            np.testing.assert_array_equal(a, b)
    return outs[0] + (U.cpu().numpy().transpose(0, 3, 1, 2).copy(),)


@pytest.mark.parametrize("P,N,H", [(64, 4096, 50),      # 4 workgroups of 1 024 candidates per problem (the headline's shape)
                                   (300, 2048, 20),     # two candidates per lane: 4 workgroups of 512
                                   (3, 65536 + 1024, 50),   # beyond the one-launch solve's 1 024 workgroups of 64
                                   (520, 1000, 33)])    # a ragged last workgroup, one candidate per lane
@pytest.mark.parametrize("sampled", [True, False])
def test_tailed_rollout_equals_two_launches_and_the_oracle(monkeypatch, P, N, H, sampled):
    from acmpc_amd import _capi
    n = H - 1
    problems = [make_problem(orc, "monza", H, 4, seed=900 + p % 7) for p in range(P)]
    one = _solve(monkeypatch, "ACMPC_TAILED_ROLLOUT", problems, P, N, n, sampled)
    two = _solve(monkeypatch, None, problems, P, N, n, sampled)                       # (the default: two launches; from
    waves = _solve(monkeypatch, "ACMPC_NO_LANE_FINALIZE", problems, P, N, n, sampled)  # 256 problems a lane per problem)
    for other in (two, waves):
        for a, b, what in zip(one, other, ("records", "keys", "costs", "controls")):
            np.testing.assert_array_equal(a, b, err_msg=what)
    rec, keys, costs, U = one
    for p in (0, P // 2, P - 1):
        prob, cfg = problems[p], problems[p]["cfg"]
        coef = orc.coefficients_spatial(prob["table"], prob["limits"].margin)
        cost, viol, X = orc.rollout_spatial(prob["x0"], coef, U[p], cfg["step_cost"], cfg["r_term"], cfg["final_cost"],
                                            prob["u_lo"], prob["u_hi"], 1.0e6, dtype=np.float32, return_states=True)
        np.testing.assert_array_equal(costs[p], cost)
        best, _ = orc.pick_best(cost)
        assert _capi.key_index(int(keys[p])) == best
        out = _capi.split_record(rec[p], n)
        assert out["cost"] == cost[best] and out["violation"] == viol[best]
        assert int(out["n_feasible"]) == int(np.count_nonzero(viol == 0))
        np.testing.assert_array_equal(out["u"], U[p, best])
        np.testing.assert_array_equal(out["x"], X[best])
