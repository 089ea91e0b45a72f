"""Ties the build-defined sampling composition to the reference-assembled QP (not GPU).

For any control sequence U, the mode-S rollout X(U) must satisfy the reference's equality rows
(`A_eq z = l_eq`, control.py:35-45,67) and its cost must equal `1/2 z'Pz + q'z` (control.py:72-79,151-158) up to the
constant `1/2 sum u_ref' R u_ref` - with P, q, A, l taken from the golden file, i.e. built by the reference itself.
"""
from types import SimpleNamespace

import numpy as np

import acmpc_oracle as orc
from test_support import make_problem


def _limits(v):
    return SimpleNamespace(length=v[0], width=v[1], delta_max=v[2], margin=v[3],
                           min_u=np.array(v[4:6]), max_u=np.array(v[6:8]))


def test_rollout_satisfies_reference_equalities_and_cost(golden, golden_cases):
    rng = np.random.default_rng(0)
    for key in golden_cases[::3]:
        table = golden[key + "/table"]
        n = table.shape[1]
        w = golden[key + "/weights"]
        lim = _limits(golden[key + "/limits"])
        x0 = golden[key + "/spatial_state"]
        coef64 = _coef64(table, lim.margin)
        u_ref = np.stack([table[orc.ROW_V], table[orc.ROW_KAPPA]], axis=1)
        U = u_ref[None] + rng.standard_normal((6, n, 2)) * np.array([1.5, 0.005])
        u_lo, u_hi = orc.input_box(lim)
        cost, viol, X = orc.rollout_spatial(x0, coef64, U, w[0:3], w[3:5], w[5:8], u_lo, u_hi, 0.0,
                                            dtype=np.float64, return_states=True)
        A, l, Pd, q = golden[key + "/qp_A"], golden[key + "/qp_l"], golden[key + "/qp_Pdiag"], golden[key + "/qp_q"]
        n_eq = 3 * (n + 1)
        const = 0.5 * float(np.sum(w[3:5] * u_ref**2))
        for c in range(U.shape[0]):
            z = orc.pack_decision_vector(X[c], U[c])
            scale = 1.0 + np.abs(z).max()
            np.testing.assert_allclose(A[:n_eq] @ z, l[:n_eq], rtol=0, atol=1e-9 * scale, err_msg=key)
            assert abs(orc.qp_objective(Pd, q, z) - (cost[c] - const)) <= 1e-9 * (1.0 + abs(cost[c]))


def _coef64(table, margin):
    # the float64 table, not rounded to float32, for the 1e-9 identity check
    f, A, B = orc.linearise(table)
    n = table.shape[1]
    c = np.zeros((n, orc.COEF_STRIDE_S))
    c[:, orc.CS_DS], c[:, orc.CS_A21], c[:, orc.CS_A31] = A[:, 0, 1], A[:, 1, 0], A[:, 2, 0]
    c[:, orc.CS_B31], c[:, orc.CS_F3] = B[:, 2, 0], f[:, 2]
    c[:, orc.CS_VREF], c[:, orc.CS_KREF] = table[orc.ROW_V], table[orc.ROW_KAPPA]
    c[:, orc.CS_EYLO] = -table[orc.ROW_WIDTH] / 2 + margin
    c[:, orc.CS_EYHI] = table[orc.ROW_WIDTH] / 2 - margin
    return c


def test_feasibility_matches_reference_box_rows(golden, golden_cases):
    """viol == 0  <=>  every box row l <= z <= u of the reference QP holds (x_0's own rows excepted: the reference
    pins t_0 = 0 by equality while boxing t >= 0.01, control.py:134 vs :67 - SURVEY.md section 7)."""
    rng = np.random.default_rng(1)
    for key in golden_cases[::4]:
        table = golden[key + "/table"]
        n = table.shape[1]
        w = golden[key + "/weights"]
        lim = _limits(golden[key + "/limits"])
        x0 = golden[key + "/spatial_state"]
        u_ref = np.stack([table[orc.ROW_V], table[orc.ROW_KAPPA]], axis=1)
        U = u_ref[None] + rng.standard_normal((40, n, 2)) * np.array([4.0, 0.03])
        u_lo, u_hi = orc.input_box(lim)
        cost, viol, X = orc.rollout_spatial(x0, _coef64(table, lim.margin), U, w[0:3], w[3:5], w[5:8], u_lo, u_hi,
                                            1e6, dtype=np.float64, return_states=True)
        l, u = golden[key + "/qp_l"], golden[key + "/qp_u"]
        n_eq = 3 * (n + 1)
        lo, hi = l[n_eq + 3:], u[n_eq + 3:]  # skip the three box rows of x_0
        for c in range(U.shape[0]):
            z = orc.pack_decision_vector(X[c], U[c])[3:]
            inside = bool(np.all(z >= lo) and np.all(z <= hi))
            assert inside == (viol[c] == 0.0), key


def test_float32_spec_tracks_float64(golden):
    """fp32 drift of the spec order (reported, bounded): costs agree with float64 to 1e-4 relative."""
    for mode in (0, 1):
        prob = make_problem(orc, "monza", 50, 512, seed=4)
        cfg = prob["cfg"]
        if mode == 0:
            coef = orc.coefficients_spatial(prob["table"], prob["limits"].margin)
            a32 = orc.rollout_spatial(prob["x0"], coef, prob["U"], cfg["step_cost"], cfg["r_term"], cfg["final_cost"],
                                      prob["u_lo"], prob["u_hi"], 1e6, dtype=np.float32)
            a64 = orc.rollout_spatial(prob["x0"], coef, prob["U"], cfg["step_cost"], cfg["r_term"], cfg["final_cost"],
                                      prob["u_lo"], prob["u_hi"], 1e6, dtype=np.float64)
        else:
            coef = orc.coefficients_temporal(prob["table"], prob["limits"].margin)
            a32 = orc.rollout_temporal(prob["pose0"], coef, prob["U"], cfg["step_cost"], cfg["r_term"],
                                       cfg["final_cost"], prob["u_lo"], prob["u_hi"], 1e6, 0.05, dtype=np.float32)
            a64 = orc.rollout_temporal(prob["pose0"], coef, prob["U"], cfg["step_cost"], cfg["r_term"],
                                       cfg["final_cost"], prob["u_lo"], prob["u_hi"], 1e6, 0.05, dtype=np.float64,
                                       libm_trig=True)
        # (a candidate clipped onto the input box can be "outside" by one float32 rounding in float64: compare
        # by cost, where such a 1e-9 excursion weighs 1e6 * 1e-18)
        feasible = (a32[1] == 0) & (a64[0] < 100.0)
        assert feasible.sum() > 50
        rel = np.abs(a32[0][feasible] - a64[0][feasible]) / np.maximum(1.0, np.abs(a64[0][feasible]))
        assert rel.max() < 1e-4, (mode, rel.max())
        assert a32[0].dtype == np.float32


def test_spec_trig_accuracy():
    phi = np.linspace(-20, 20, 200001)
    s, c = orc.sincos_spec(phi.astype(np.float32))
    assert np.abs(s - np.sin(phi.astype(np.float32).astype(np.float64))).max() < 2.5e-7
    assert np.abs(c - np.cos(phi.astype(np.float32).astype(np.float64))).max() < 2.5e-7
    a = np.linspace(-12, 12, 100001).astype(np.float32)
    assert np.abs(orc.wrap_spec(a) - orc.wrap_to_pi(a.astype(np.float64))).max() < 2e-6 + 0  # off only at the seam
