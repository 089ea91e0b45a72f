"""The host-pointer solve (acmpc_solve, `Engine.solve`) in its transfer forms (round 4): start states, keys and records read
and written IN PLACE in the handle's page-locked block by the one-launch solve, the control matrix read in place when
the caller built it in page-locked memory (acmpc_host_alloc / `pinned_empty`), every transfer a copy with
ACMPC_NO_ZERO_COPY=1 - the same bits, and the oracle's."""
import numpy as np
import pytest

import acmpc_oracle as orc
from test_support import engine_kwargs, make_problem

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("layout", [0, 1])
@pytest.mark.parametrize("P,N", [(1, 4096),      # the one-launch solve (mode S): everything in place
                                 (3, 1536),
                                 (40, 8192)])    # beyond it: rollout + finalize, the small transfers as copies
def test_pinned_pageable_and_copied_forms_agree(monkeypatch, mode, layout, P, N):
    from acmpc_amd import Engine, _capi
    H = 50
    n = H - 1
    problems = [make_problem(orc, "silverstone", H, N, seed=400 + p) for p in range(min(P, 4))]
    problems = [problems[p % len(problems)] for p in range(P)]
    x0 = np.stack([p["x0"] if mode == 0 else p["pose0"] for p in problems]).astype(np.float32)
    U = np.stack([np.roll(p["U"], 31 * i, axis=0) for i, p in enumerate(problems)])          # [P,N,n,2]
    U = U if layout == 0 else np.ascontiguousarray(U.transpose(0, 2, 3, 1))
    results = {}
    for form in ("pageable", "pinned", "copies"):
        monkeypatch.delenv("ACMPC_NO_ZERO_COPY", raising=False)
        if form == "copies":
            monkeypatch.setenv("ACMPC_NO_ZERO_COPY", "1")
        eng = Engine(**engine_kwargs(problems[0], mode, P, N, n))
        eng.set_paths(np.stack([p["table"] for p in problems]))
        matrix = U
        if form == "pinned":
            matrix = _capi.pinned_empty(U.shape, np.float32)
            matrix[...] = U
        for _ in range(2):   # (the handle's block is reused: the second call must not see the first one's leftovers)
            out = eng.solve(x0, matrix, layout=layout)
        results[form] = out
        eng.close()
    for form in ("pinned", "copies"):
        for key in ("records", "costs", "best_idx", "cost"):
            np.testing.assert_array_equal(results[form][key], results["pageable"][key], err_msg="%s %s" % (form, key))
    prob, cfg = problems[0], problems[0]["cfg"]
    U0 = U[0] if layout == 0 else U[0].transpose(2, 0, 1)
    args = (U0, cfg["step_cost"], cfg["r_term"], cfg["final_cost"], prob["u_lo"], prob["u_hi"], 1.0e6)
    if mode == 0:
        cost, _ = orc.rollout_spatial(prob["x0"], orc.coefficients_spatial(prob["table"], prob["limits"].margin), *args,
                                      dtype=np.float32)[:2]
    else:
        cost = orc.rollout_temporal(prob["pose0"], orc.coefficients_temporal(prob["table"], prob["limits"].margin), *args, 0.05,
                                    dtype=np.float32)[0]
    np.testing.assert_array_equal(results["pageable"]["costs"][0], cost)
