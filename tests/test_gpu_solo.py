"""acmpc_solve_device / acmpc_solve in ONE launch (rollout_solo_kernel, mode S): rollout, argmin and the winner's record
assembled from the winning workgroup's state trace - no second roll of the winner.  Three forms of the same solve must
agree bit for bit with each other and with the oracle: two waves per workgroup (cost wave + bounds wave), one wave per
workgroup - each with the walk's states in registers (horizon 50) or in LDS - and the two launches (rollout_kernel +
finalize_kernel) that ACMPC_NO_SOLO keeps."""
import numpy as np
import pytest

import acmpc_oracle as orc
from test_support import engine_kwargs, make_problem

pytestmark = pytest.mark.gpu

FORMS = {"split": {"ACMPC_SOLO_SPLIT": "1"}, "one_wave": {"ACMPC_SOLO_SPLIT": "0"}, "two_launches": {"ACMPC_NO_SOLO": "1"},
         # horizon 50 only - where the states of a walk are kept: registers or LDS (otherwise chosen by launch size)
         "registers": {"ACMPC_SOLO_REGISTERS": "1"}, "lds_trace": {"ACMPC_SOLO_REGISTERS": "0"},
         "one_wave_registers": {"ACMPC_SOLO_SPLIT": "0", "ACMPC_SOLO_REGISTERS": "1"},
         "one_wave_lds_trace": {"ACMPC_SOLO_SPLIT": "0", "ACMPC_SOLO_REGISTERS": "0"}}
KEYS = ("ACMPC_SOLO_SPLIT", "ACMPC_NO_SOLO", "ACMPC_SOLO_REGISTERS")


def _solve(monkeypatch, form, eng, x0, U, layout):
    for key in KEYS:
        monkeypatch.delenv(key, raising=False)
    for key, value in FORMS[form].items():
        monkeypatch.setenv(key, value)
    return eng.solve(x0, U, layout=layout)


def _as_layout(U, layout):
    return U if layout == 0 else np.ascontiguousarray(U.transpose(0, 2, 3, 1))


@pytest.mark.parametrize("layout", [0, 1])
@pytest.mark.parametrize("track,H,N,P", [
    ("monza", 50, 4096, 1),        # BASELINE config 2 as one problem
    ("spa", 50, 1000, 3),          # ragged N, several problems
    ("nordschleife", 80, 2085, 2), # config 4's horizon, a partial last workgroup
    ("monza", 20, 128, 1),
    ("silverstone", 50, 1, 1),     # one candidate
    ("monza", 50, 67, 5),
    ("monza", 101, 640, 1),        # the mapping controller's horizon (H = 100 + 1)
])
def test_three_forms_and_the_oracle_agree(monkeypatch, layout, track, H, N, P):
    from acmpc_amd import Engine
    n = H - 1
    problems = [make_problem(orc, track, H, N, seed=500 + 7 * P + p) for p in range(P)]
    if N > 8:
        problems[0]["U"][5, 3, 0] = np.nan      # a non-finite candidate
        problems[-1]["U"][2, :, 1] += 0.5       # far outside the input box: infeasible, huge cost
    eng = Engine(**engine_kwargs(problems[0], 0, P, N, n))
    eng.set_paths(np.stack([p["table"] for p in problems]))
    x0 = np.stack([p["x0"] for p in problems])
    U = _as_layout(np.stack([p["U"] for p in problems]), layout)
    outs = {form: _solve(monkeypatch, form, eng, x0, U, layout) for form in FORMS}
    for form in FORMS:
        for key in ("costs", "best_idx", "records"):
            np.testing.assert_array_equal(outs[form][key], outs["two_launches"][key], err_msg="%s: %s" % (form, key))
    out = outs["split"]
    cfg = problems[0]["cfg"]
    for p, prob in enumerate(problems):
        cost, viol, X = orc.rollout_spatial(prob["x0"], eng.coefficients(p), prob["U"], cfg["step_cost"], cfg["r_term"],
                                            cfg["final_cost"], prob["u_lo"], prob["u_hi"], 1.0e6, dtype=np.float32,
                                            return_states=True)[:3]
        np.testing.assert_array_equal(out["costs"][p], cost)
        best, best_cost = orc.pick_best(cost)
        assert out["best_idx"][p] == best and out["cost"][p] == np.float32(best_cost)
        assert out["violation"][p] == viol[best] and out["n_feasible"][p] == np.count_nonzero(viol == 0)
        assert out["owner"][p] == 1.0
        np.testing.assert_array_equal(out["u"][p], prob["U"][best])
        np.testing.assert_array_equal(out["x"][p], X[best])
    eng.close()


@pytest.mark.parametrize("layout", [0, 1])
def test_config_3_as_one_launch(monkeypatch, layout):
    """65 536 candidates x horizon 50 = 1 024 workgroups: the largest launch the one-launch form takes (32 ticket
    groups, sixteen partial keys per lane of the last workgroup); one more workgroup goes back to two launches."""
    import torch
    from acmpc_amd import Engine, _capi
    H, n = 50, 49
    prob = make_problem(orc, "spa", H, 8, seed=3)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    cfg = prob["cfg"]
    for N in (65536, 65536 + 64):
        eng = Engine(**engine_kwargs(prob, 0, 1, N, n))
        eng.set_paths(prob["table"][None])
        eng.sync_tables(stream)
        u_ref = torch.tensor(np.stack([prob["table"][orc.ROW_V], prob["table"][orc.ROW_KAPPA]], axis=1)[None],
                             dtype=torch.float32, device=dev).contiguous()
        U = torch.empty((1, n, 2, N) if layout == 1 else (1, N, n, 2), device=dev)
        eng.sample_device(u_ref.data_ptr(), 2 * n, u_ref.data_ptr(), 1, N, n, layout, 0, (2.0, 0.01), 11, 0, U.data_ptr(),
                          stream)
        x0 = torch.tensor(prob["x0"][None], device=dev)
        results = {}
        for form in FORMS:
            for key in KEYS:
                monkeypatch.delenv(key, raising=False)
            for key, value in FORMS[form].items():
                monkeypatch.setenv(key, value)
            costs = torch.full((1, N), -1.0, device=dev)
            keys = torch.zeros(1, dtype=torch.int64, device=dev)
            rec = torch.zeros(1, _capi.record_floats(n), device=dev)
            for _ in range(3):   # repeated launches: the tickets are left as they were found
                eng.solve_device(x0.data_ptr(), U.data_ptr(), 1, N, n, layout, costs.data_ptr(), keys.data_ptr(),
                                 rec.data_ptr(), stream)
            torch.cuda.synchronize()
            results[form] = (costs.cpu().numpy(), keys.cpu().numpy(), rec.cpu().numpy())
        for form in FORMS:
            for got, want in zip(results[form], results["two_launches"]):
                np.testing.assert_array_equal(got, want, err_msg=form)
        costs, keys, rec = results["split"]
        best, best_cost = orc.pick_best(costs[0])
        assert _capi.key_index(int(keys[0])) == best and rec[0, _capi.REC_COST] == np.float32(best_cost)
        Uh = U.cpu().numpy()[0]
        Uh = Uh if layout == 0 else Uh.transpose(2, 0, 1)
        sub = np.concatenate([[best], np.random.default_rng(N).choice(N, 255, replace=False)])
        cost, viol, X = orc.rollout_spatial(prob["x0"], eng.coefficients(0), np.ascontiguousarray(Uh[sub]), cfg["step_cost"],
                                            cfg["r_term"], cfg["final_cost"], prob["u_lo"], prob["u_hi"], 1.0e6,
                                            dtype=np.float32, return_states=True)[:3]
        np.testing.assert_array_equal(costs[0, sub], cost)
        split = _capi.split_record(rec, n)
        np.testing.assert_array_equal(split["u"][0], Uh[best])
        np.testing.assert_array_equal(split["x"][0], X[0])
        assert split["violation"][0] == viol[0]
        eng.close()


def test_keys_only_and_records_only(monkeypatch):
    """d_records or d_keys may be null: the one-launch form writes what was asked for."""
    import torch
    from acmpc_amd import Engine, _capi
    H, N, n = 50, 2048, 49
    prob = make_problem(orc, "monza", H, N, seed=9)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    eng = Engine(**engine_kwargs(prob, 0, 1, N, n))
    eng.set_paths(prob["table"][None])
    want = eng.solve(prob["x0"][None], prob["U"][None], layout=0)
    x0 = torch.tensor(prob["x0"][None], device=dev)
    U = torch.tensor(prob["U"][None], device=dev)
    keys = torch.zeros(1, dtype=torch.int64, device=dev)
    rec = torch.zeros(1, _capi.record_floats(n), device=dev)
    eng.solve_device(x0.data_ptr(), U.data_ptr(), 1, N, n, 0, 0, keys.data_ptr(), 0, stream)
    eng.solve_device(x0.data_ptr(), U.data_ptr(), 1, N, n, 0, 0, 0, rec.data_ptr(), stream)
    torch.cuda.synchronize()
    assert _capi.key_index(int(keys[0])) == want["best_idx"][0]
    np.testing.assert_array_equal(rec.cpu().numpy(), want["records"])
    eng.close()
