"""The frozen outputs of the build-defined composition (tests/golden/gen_composition.py -> composition.npz):
the NumPy oracle and its C restatement must still produce them bit for bit (CPU), and so must the HIP kernels
(`-m gpu`, through the C ABI), at the stored sizes and - by SHA-256 of the cost vector - at BASELINE's full sizes."""
import hashlib
import os

import numpy as np
import pytest

import acmpc_oracle as orc
import c_oracle
from test_support import engine_kwargs, full_size_controls, make_problem

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "composition.npz")


@pytest.fixture(scope="module")
def frozen():
    return np.load(GOLDEN)


def _names(key):
    return [str(c) for c in np.load(GOLDEN)[key]]


def _case(frozen, name):
    g = {k.split("/", 1)[1]: frozen[k] for k in frozen.files if k.startswith(name + "/")}
    window = tuple(int(v) for v in g["window"])
    g["window"] = None if window[0] < 0 else window
    g["mode"] = int(g["mode"])
    return g


@pytest.mark.parametrize("name", _names("cases"))
def test_oracles_reproduce_the_frozen_composition(frozen, name):
    g = _case(frozen, name)
    Q, R, QN = g["weights"][:3], g["weights"][3:5], g["weights"][5:]
    lo, hi = g["box"][:2], g["box"][2:]
    if g["mode"] == 0:
        coef = orc.coefficients_spatial(g["table"], float(g["margin"]))
        cost, viol, X = orc.rollout_spatial(g["start"], coef, g["U"], Q, R, QN, lo, hi, 1.0e6, dtype=np.float32,
                                            return_states=True)
    else:
        coef = orc.coefficients_temporal(g["table"], float(g["margin"]))
        cost, viol, X, _ = orc.rollout_temporal(g["start"], coef, g["U"], Q, R, QN, lo, hi, 1.0e6, 0.05,
                                                dtype=np.float32, return_states=True, nn_window=g["window"])
    np.testing.assert_array_equal(coef, g["coef"])
    np.testing.assert_array_equal(cost, g["cost"])
    np.testing.assert_array_equal(viol, g["violation"])
    best, best_cost = orc.pick_best(cost)
    assert best == int(g["best"]) and np.float32(best_cost) == g["best_cost"]
    np.testing.assert_array_equal(X[best], g["best_x"])
    w = c_oracle.make_weights(Q, R, QN, lo, hi, 1.0e6, nn_window=g["window"])
    c_cost, c_viol, c_states = c_oracle.rollout(g["mode"], g["start"], coef, g["U"], 0, w, return_states=True)
    np.testing.assert_array_equal(c_cost, g["cost"])
    np.testing.assert_array_equal(c_viol, g["violation"])
    np.testing.assert_array_equal(c_states[best], g["best_x"])


def _full_case(frozen, name):
    H, N, mode, seed, back, ahead = (int(v) for v in frozen[name + "/spec"])
    track = str(frozen[name + "/track"])
    prob = make_problem(orc, track, H, 16, seed=seed)
    U = full_size_controls(orc, prob, N, H - 1)
    assert hashlib.sha256(U.tobytes()).hexdigest() == str(frozen[name + "/controls_sha256"]), "inputs drifted"
    return prob, U, H, N, mode, (None if back < 0 else (back, ahead))


@pytest.mark.parametrize("name", _names("full_cases"))
def test_c_oracle_reproduces_the_full_size_digests(frozen, name):
    prob, U, H, N, mode, window = _full_case(frozen, name)
    cfg = prob["cfg"]
    w = c_oracle.make_weights(cfg["step_cost"], cfg["r_term"], cfg["final_cost"], prob["u_lo"], prob["u_hi"], 1.0e6,
                              nn_window=window)
    if mode == 0:
        coef, start = orc.coefficients_spatial(prob["table"], prob["limits"].margin), prob["x0"]
    else:
        coef, start = orc.coefficients_temporal(prob["table"], prob["limits"].margin), prob["pose0"]
    cost, viol = c_oracle.rollout(mode, start, coef, U, 0, w)
    assert hashlib.sha256(cost.tobytes()).hexdigest() == str(frozen[name + "/cost_sha256"])
    assert c_oracle.argmin(cost) == int(frozen[name + "/best"])
    assert np.count_nonzero(viol == 0) == int(frozen[name + "/n_feasible"])


@pytest.mark.gpu
@pytest.mark.parametrize("layout", [0, 1])
@pytest.mark.parametrize("name", _names("cases"))
def test_kernels_reproduce_the_frozen_composition(frozen, name, layout):
    from acmpc_amd import Engine
    g = _case(frozen, name)
    N, n = g["U"].shape[0], g["U"].shape[1]
    eng = Engine(mode=g["mode"], max_problems=1, max_candidates=N, max_steps=n, step_cost=g["weights"][:3],
                 r_term=g["weights"][3:5], final_cost=g["weights"][5:], u_min=g["box"][:2], u_max=g["box"][2:],
                 margin=float(g["margin"]), wheelbase=2.65, t_min=0.01, dt=0.05, w_bound=1.0e6, nn_window=g["window"])
    eng.set_paths(g["table"])
    np.testing.assert_array_equal(eng.coefficients(0), g["coef"])
    U = g["U"] if layout == 0 else np.ascontiguousarray(g["U"].transpose(1, 2, 0))
    out = eng.solve(g["start"][None], U[None], layout=layout)
    np.testing.assert_array_equal(out["costs"][0], g["cost"])
    assert out["best_idx"][0] == int(g["best"]) and out["cost"][0] == g["best_cost"]
    assert out["violation"][0] == g["violation"][int(g["best"])]
    np.testing.assert_array_equal(out["u"][0], g["best_u"])
    np.testing.assert_array_equal(out["x"][0], g["best_x"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", _names("full_cases"))
def test_kernels_reproduce_the_full_size_digests(frozen, name):
    from acmpc_amd import Engine
    prob, U, H, N, mode, window = _full_case(frozen, name)
    n = H - 1
    eng = Engine(**engine_kwargs(prob, mode, 1, N, n, nn_window=window))
    eng.set_paths(prob["table"])
    start = prob["x0"] if mode == 0 else prob["pose0"]
    out = eng.solve(start[None], np.ascontiguousarray(U.transpose(1, 2, 0))[None], layout=1)
    assert hashlib.sha256(out["costs"][0].tobytes()).hexdigest() == str(frozen[name + "/cost_sha256"])
    assert out["best_idx"][0] == int(frozen[name + "/best"])
    assert out["n_feasible"][0] == int(frozen[name + "/n_feasible"])


@pytest.mark.parametrize("name", _names("solves"))
def test_the_oracle_reproduces_the_frozen_solves(frozen, name):
    """One whole solve (sample -> roll -> argmin -> recentre, two rounds of 16 384 candidates, the LQ plan in the last):
    oracle.optimize_restated still returns the frozen record, and the library's host-side LQ plan (acmpc_lq_plan) is the
    frozen one bit for bit.  The GPU half is tests/test_gpu_restated_solve.py."""
    from acmpc_amd import _capi
    g = {k.split("/", 1)[1]: frozen[k] for k in frozen.files if k.startswith(name + "/")}
    mode = int(g["mode"])
    window = tuple(int(v) for v in g["window"])
    window = None if window[0] < 0 else window
    Q, R, QN = g["weights"][:3], g["weights"][3:5], g["weights"][5:]
    lo, hi = g["box"][:2], g["box"][2:]
    frenet = g["start"].astype(np.float64) if mode == 0 else orc.frenet_start(g["table"], g["start"].astype(np.float64))
    plan = orc.lq_plan(g["table"], frenet, Q, R, QN, lo, hi)
    np.testing.assert_array_equal(plan, g["plan"])
    np.testing.assert_array_equal(_capi.lq_plan(g["table"], frenet, Q, R, QN, lo, hi), g["plan"])
    won = orc.optimize_restated(mode, g["start"], g["coef"], g["centre"], g["u_ref"], int(g["n_candidates"]), int(g["rounds"]),
                                tuple(g["sigma"]), float(g["shrink"]), int(g["seed"]), Q, R, QN, lo, hi, 1.0e6, 0.05, window,
                                extra=plan)
    assert won["cost"] == g["cost"] and won["violation"] == g["violation"] and won["n_feasible"] == int(g["n_feasible"])
    assert won["winners"] == [int(v) for v in g["winners"]]
    np.testing.assert_array_equal(won["u"], g["u"])
    np.testing.assert_array_equal(won["x"], g["x"])
