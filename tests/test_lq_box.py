"""The box-constrained LQ plan (csrc/acmpc_lq_box.h, `lq_candidate = 2`) on the CPU: the library's host solver against its
line-by-line restatement in the oracle (bit for bit, cold and warm), against the optimum of the reference's QP WITH its box
rows (/root/reference/src/acmpc/control/solvers/control.py:26-79,130-144, assembled by `oracle.control_qp` - pinned to the
reference's own assembly by G4 / G5 - and solved by `oracle.osqp_restated`; QP solutions are parity-unpinned, this is a
quantified bound), and its invariants: never worse than the LQ plan under the cost the kernels charge, untouched where
the LQ plan already is the optimum.  No GPU work: `acmpc_lq_box_plan` needs no handle."""
import numpy as np
import pytest

import acmpc_oracle as orc
from test_support import DELTA_MAX, RACING, WHEELBASE, WIDTH

H = 50
W_BOUND = 1.0e4   # the controller's default (sampling_solver.py)


def _problem(track, coords, offset=0.0):
    """The racing configuration of `track` (configs/<track>.yaml:67-81; the control process overwrites v_max with the
    unlocalised reference speed, controller.py:241-243) on one H x 3 path: table with a solved speed profile, Frenet
    start state, weights, limits."""
    from acmpc_amd.reference_path import ReferencePath
    from acmpc_amd.speed_profile import SpeedProfileSolver
    cfg = RACING[track]
    cons = dict(cfg["speed_profile_constraints"])
    limits = orc.vehicle_limits(WHEELBASE, WIDTH, DELTA_MAX, cons["v_min"], cons["v_max"])
    live = dict(cons, v_max=float(cfg["unlocalised_max_speed"]))
    table = orc.construct_waypoints(coords)
    dec = SpeedProfileSolver({"control_horizon": H - 1, "max_iterations": 4000, "constraints": live}).solve(
        ReferencePath.from_table(table), live["end_velocity"])
    assert dec.info.status == "solved"
    table[orc.ROW_V] = dec.x
    x0 = orc.t2s(table[:3, 0], np.array([offset, 0.0, np.pi / 2]))
    lo, hi = orc.input_box(limits)
    return dict(cfg=cfg, limits=limits, table=table, x0=x0, lo=lo, hi=hi)


def _args(p, iterations=40):
    cfg = p["cfg"]
    return (p["table"], p["x0"], cfg["step_cost"], cfg["r_term"], cfg["final_cost"], p["lo"], p["hi"], p["limits"].margin,
            W_BOUND, iterations)


def _state_vector(n, state):
    return np.concatenate([[n]] + [state[k].ravel() for k in ("wx", "wu", "lx", "lu")])


def _same(lib_out, orc_out, n):
    assert lib_out["iterations"] == orc_out["iterations"] and lib_out["chosen"] == orc_out["chosen"]
    assert lib_out["triggered"] == orc_out["triggered"]
    np.testing.assert_array_equal(lib_out["plan"], orc_out["plan"])
    assert lib_out["J"] == orc_out["J"] and lib_out["V"] == orc_out["V"]
    assert (lib_out["state"] is None) == (orc_out["state"] is None)
    if lib_out["state"] is not None:
        np.testing.assert_array_equal(lib_out["state"], _state_vector(n, orc_out["state"]))


def _cases():
    from acmpc_amd import workloads as wl
    return {
        "corner entry r 8, monza": ("monza", wl.corner_entry_path(8.0, 30.0, H), 0.0),
        "corner entry r 7.5 after 60 m, nordschleife": ("nordschleife", wl.corner_entry_path(7.5, 60.0, H), 0.0),
        "s-bend r 8, spa": ("spa", wl.s_bend_path(8.0, 20.0, H), 0.0),
        "corner entry r 10 from 2 m off, silverstone": ("silverstone", wl.corner_entry_path(10.0, 30.0, H), 2.0),
        "hairpin r 9 (input box only), monza": ("monza", wl.racing_widths(wl.family_path("hairpin", 9.0, H)), 0.0),
        "hairpin r 10 entered at 30 degrees (no feasible plan), monza":
            ("monza", wl.racing_widths(wl.family_path("hairpin", 10.0, H, angle=-np.pi / 6)), 0.0),
        "gentle bend (nothing active), monza": ("monza", wl.racing_widths(wl.family_path("hairpin", 40.0, H)), 1.0),
    }


@pytest.mark.parametrize("name", list(_cases()))
def test_the_host_solver_equals_its_restatement(name):
    """acmpc_lq_box_plan == oracle.lq_box_plan bit for bit: cold, warm from its own iterate, and warm on the path as the
    next tick sees it (the car 0.3 m further along and 5 cm to the side) - plan, iterate, iteration count, choice."""
    from acmpc_amd import _capi
    track, coords, offset = _cases()[name]
    p = _problem(track, coords, offset)
    n = H - 1
    cold_lib, cold_orc = _capi.lq_box_plan(*_args(p)), orc.lq_box_plan(*_args(p))
    _same(cold_lib, cold_orc, n)
    warm_lib = _capi.lq_box_plan(*_args(p), state=cold_lib["state"])
    warm_orc = orc.lq_box_plan(*_args(p), state=cold_orc["state"])
    _same(warm_lib, warm_orc, n)
    moved = np.array(coords)
    moved[:, 1] -= 0.3
    q = _problem(track, moved, offset + 0.05)
    next_lib = _capi.lq_box_plan(*_args(q, 10), state=warm_lib["state"])
    next_orc = orc.lq_box_plan(*_args(q, 10), state=warm_orc["state"])
    _same(next_lib, next_orc, n)
    if cold_lib["triggered"] and cold_lib["iterations"] < 40:   # a converged iterate confirms itself at once
        assert warm_lib["iterations"] <= 2, warm_lib["iterations"]


def test_untouched_where_the_lq_plan_is_the_optimum():
    from acmpc_amd import _capi
    track, coords, offset = _cases()["gentle bend (nothing active), monza"]
    p = _problem(track, coords, offset)
    out = _capi.lq_box_plan(*_args(p))
    plain = _capi.lq_plan(*_args(p)[:7])
    assert not out["triggered"] and out["iterations"] == 0 and out["chosen"] == 0 and out["state"] is None
    np.testing.assert_array_equal(out["plan"], plain)


def _qp_tracking_optimum(p):
    """Tracking cost 1/2 (z - z_ref)'P(z - z_ref) of the reference QP's optimum (x_0's box rows dropped: the reference
    pins t_0 = 0 while boxing t >= 0.01, control.py:134 vs :67)."""
    n = H - 1
    qp = orc.control_qp(p["x0"], p["table"], p["cfg"], p["limits"])
    n_eq = 3 * (n + 1)
    keep = np.r_[0:n_eq, n_eq + 3:len(qp["l"])]
    ref = orc.osqp_restated(qp["P_diag"], qp["q"], qp["A"][keep], qp["l"][keep], qp["u"][keep], max_iter=20000, eps_abs=1e-5,
                            eps_rel=1e-5, adaptive_rho=True)
    assert ref.info.status == "solved"
    weighted = qp["P_diag"] > 0
    return orc.qp_objective(qp["P_diag"], qp["q"], ref.x) + 0.5 * float(np.sum(qp["q"][weighted] ** 2 / qp["P_diag"][weighted]))


@pytest.mark.parametrize("name", ["corner entry r 8, monza", "corner entry r 7.5 after 60 m, nordschleife", "s-bend r 8, spa",
                                  "hairpin r 9 (input box only), monza"])
def test_the_optimum_the_plan_is_measured_against_is_a_kkt_point(name):
    """`oracle.osqp_restated` restates a published algorithm and is pinned by nothing of the reference's (no OSQP here), so
    the optimum the bounds of this file and of tests/test_gpu_qp_gap.py are quoted against is certified solver-independently:
    at 1e-7 its primal-dual pair satisfies the KKT conditions of the reference's QP - rows met, stationary, multipliers of the
    right sign and only on rows that are active - which for a convex QP makes it THE optimum; and the looser solve the bounds
    use (1e-5) has the same objective to 1e-4."""
    track, coords, offset = _cases()[name]
    p = _problem(track, coords, offset)
    n = H - 1
    qp = orc.control_qp(p["x0"], p["table"], p["cfg"], p["limits"])
    n_eq = 3 * (n + 1)
    keep = np.r_[0:n_eq, n_eq + 3:len(qp["l"])]
    A, l, u = qp["A"][keep], qp["l"][keep], qp["u"][keep]
    tight = orc.osqp_restated(qp["P_diag"], qp["q"], A, l, u, max_iter=100000, eps_abs=1e-7, eps_rel=1e-7, adaptive_rho=True)
    assert tight.info.status == "solved"
    x, y = tight.x, tight.y
    Ax = A @ x
    assert np.max(l - Ax) <= 1e-5 and np.max(Ax - u) <= 1e-5
    np.testing.assert_allclose(qp["P_diag"] * x + qp["q"] + A.T @ y, 0.0, atol=1e-6)
    box = np.isfinite(u) & np.isfinite(l) & (u > l)
    assert np.max(np.maximum(y, 0.0)[np.isfinite(u)] * (u - Ax)[np.isfinite(u)], initial=0.0) <= 1e-5     # y > 0: at its upper bound
    assert np.max(np.maximum(-y, 0.0)[np.isfinite(l)] * (Ax - l)[np.isfinite(l)], initial=0.0) <= 1e-5    # y < 0: at its lower bound
    assert not (y[~np.isfinite(u)] > 1e-9).any() and not (y[~np.isfinite(l)] < -1e-9).any()
    assert box.any() and (np.abs(y[box]) > 1e-3).any()          # and box rows ARE active in these cases
    loose = orc.osqp_restated(qp["P_diag"], qp["q"], A, l, u, max_iter=20000, eps_abs=1e-5, eps_rel=1e-5, adaptive_rho=True)
    f_tight, f_loose = (orc.qp_objective(qp["P_diag"], qp["q"], r.x) for r in (tight, loose))
    assert abs(f_tight - f_loose) <= 1e-4 * abs(f_tight)


@pytest.mark.parametrize("name", ["corner entry r 8, monza", "corner entry r 7.5 after 60 m, nordschleife", "s-bend r 8, spa",
                                  "hairpin r 9 (input box only), monza"])
def test_a_third_party_solver_brackets_the_same_optimum(name):
    """Independent of the restated OSQP: SciPy's generic constrained minimiser (trust-constr: an interior-point method, so its
    answer meets the equality rows to 1e-15 and lies strictly inside the box) on the reference's QP.  A feasible point
    cannot cost less than the optimum and the restated OSQP's iterate, infeasible by its tolerance, does not cost more (to 1e-4):
    the two bracket it within 1.5 %; and the refined plan of `acmpc_lq_box_plan` - feasible as well - is within 2 % of SciPy's."""
    from scipy import sparse
    from scipy.optimize import Bounds, LinearConstraint, minimize
    from acmpc_amd import _capi
    track, coords, offset = _cases()[name]
    p = _problem(track, coords, offset)
    n = H - 1
    qp = orc.control_qp(p["x0"], p["table"], p["cfg"], p["limits"])
    n_eq = 3 * (n + 1)
    P, q = qp["P_diag"], qp["q"]
    rows, rhs = qp["A"][:n_eq], qp["l"][:n_eq]
    lo, hi = qp["l"][n_eq:].copy(), qp["u"][n_eq:].copy()
    lo[:3], hi[:3] = -np.inf, np.inf            # x_0's box rows dropped, as in _qp_tracking_optimum
    found = minimize(lambda z: 0.5 * z @ (P * z) + q @ z, np.zeros(len(q)), jac=lambda z: P * z + q, hess=lambda z: sparse.diags(P),
                     method="trust-constr", constraints=[LinearConstraint(sparse.csr_matrix(rows), rhs, rhs)], bounds=Bounds(lo, hi),
                     options=dict(gtol=1e-9, xtol=1e-12, maxiter=5000, sparse_jacobian=True))
    assert np.abs(rows @ found.x - rhs).max() < 1e-9 and (found.x >= lo - 1e-9).all() and (found.x <= hi + 1e-9).all()
    weighted = P > 0
    from_inside = found.fun + 0.5 * float(np.sum(q[weighted] ** 2 / P[weighted]))      # tracking cost, as J
    from_outside = _qp_tracking_optimum(p)
    assert from_outside <= from_inside * (1.0 + 1e-4) and (from_inside - from_outside) / from_inside <= 0.015, (from_inside, from_outside)
    out = _capi.lq_box_plan(*_args(p, 60))
    assert out["V"] <= 1e-8 and abs(out["J"] - from_inside) / from_inside <= 0.02, (out["J"], from_inside)


@pytest.mark.parametrize("name,lq_excess_at_least", [("corner entry r 8, monza", 1.0),
                                                     ("corner entry r 7.5 after 60 m, nordschleife", 5.0),
                                                     ("s-bend r 8, spa", 0.0),
                                                     ("hairpin r 9 (input box only), monza", 0.0)])
def test_against_the_qp_optimum_where_box_rows_are_active(name, lq_excess_at_least):
    """Where the QP's box rows bind steps ahead the clipped LQ plan is several times the optimum's tracking cost (or
    leaves the corridor); the refined plan is within 2 % of it - the restated OSQP's own slack at 1e-5 is ~1 % here, its
    equality rows are met to the tolerance, the plan's exactly - with every state row met (V = 0 to 1e-8)."""
    from acmpc_amd import _capi
    track, coords, offset = _cases()[name]
    p = _problem(track, coords, offset)
    best = _qp_tracking_optimum(p)
    cfg = p["cfg"]
    plain = _capi.lq_plan(*_args(p)[:7])
    lq = orc.lq_box_rollout_cost(p["table"], p["x0"], cfg["step_cost"], cfg["r_term"], cfg["final_cost"], p["lo"], p["hi"],
                                 p["limits"].margin, plain)
    out = _capi.lq_box_plan(*_args(p, 60))
    assert out["triggered"] and 0 < out["iterations"] <= 60
    assert (lq["J"] - best) / best >= lq_excess_at_least or lq["V"] > 1e-3, (lq, best)
    assert abs(out["J"] - best) / best <= 0.02, (out["J"], best)
    assert out["V"] <= 1e-8, out["V"]
    assert out["J"] + W_BOUND * out["V"] <= lq["J"] + W_BOUND * lq["V"]


def test_never_worse_than_the_lq_plan():
    """Forty seeded curvature profiles (corners of radius 6-13 m after 5-80 m, from up to 3 m off the centre line), four
    tracks' weights: the slot's plan never rolls out dearer than the LQ plan under J + w_bound V, whatever the iteration
    cap, and every plan is inside the input box."""
    from acmpc_amd import _capi
    from acmpc_amd import workloads as wl
    rng = np.random.default_rng(5)
    triggered = 0
    for k in range(40):
        track = ("monza", "spa", "nordschleife", "silverstone")[k % 4]
        coords = wl.corner_entry_path(float(rng.uniform(6.0, 13.0)), float(rng.uniform(5.0, 80.0)), H,
                                      arc_angle=float(rng.uniform(0.3, 1.3)) * np.pi, tail=float(rng.uniform(0.0, 40.0)))
        p = _problem(track, coords, float(rng.uniform(-3.0, 3.0)))
        cfg = p["cfg"]
        plain = _capi.lq_plan(*_args(p)[:7])
        lq = orc.lq_box_rollout_cost(p["table"], p["x0"], cfg["step_cost"], cfg["r_term"], cfg["final_cost"], p["lo"], p["hi"],
                                     p["limits"].margin, plain)
        for cap in (3, 40):
            out = _capi.lq_box_plan(*_args(p, cap))
            assert out["J"] + W_BOUND * out["V"] <= lq["J"] + W_BOUND * lq["V"], (k, cap)
            assert np.all(out["plan"] >= p["lo"].astype(np.float32)) and np.all(out["plan"] <= p["hi"].astype(np.float32))
        triggered += out["triggered"]
    assert triggered >= 10, triggered


def test_degenerate_paths_are_refused_or_survived():
    """Paths the solver cannot factor - a zero step, a zero reference speed, a non-finite waypoint, a horizon of one step - come
    back as "no plan" (the round then runs without the candidate) or as a finite plan inside the input box: no exception,
    no non-finite control, no endless loop."""
    from acmpc_amd import _capi
    track, coords, offset = _cases()["corner entry r 8, monza"]
    p = _problem(track, coords, offset)
    cfg = p["cfg"]
    base = p["table"]

    def run(table, x0=p["x0"], iterations=40):
        out = _capi.lq_box_plan(table, x0, cfg["step_cost"], cfg["r_term"], cfg["final_cost"], p["lo"], p["hi"], p["limits"].margin,
                                W_BOUND, iterations)
        if out["plan"] is not None:
            assert np.isfinite(out["plan"]).all()
            assert np.all(out["plan"] >= p["lo"].astype(np.float32)) and np.all(out["plan"] <= p["hi"].astype(np.float32))
            assert 0 <= abs(out["iterations"]) <= 40
        return out

    for row, value in ((orc.ROW_DS, 0.0), (orc.ROW_V, 0.0), (orc.ROW_KAPPA, np.nan), (orc.ROW_DS, np.inf), (orc.ROW_WIDTH, np.nan),
                       (orc.ROW_WIDTH, 0.5)):          # (a corridor narrower than the car: the e_y box is empty)
        table = np.array(base)
        table[row, 20] = value
        lib, restated = run(table), orc.lq_box_plan(table, p["x0"], cfg["step_cost"], cfg["r_term"], cfg["final_cost"], p["lo"], p["hi"],
                                                    p["limits"].margin, W_BOUND, 40)
        assert (lib["plan"] is None) == (restated["plan"] is None), (row, value)
        if lib["plan"] is not None:
            np.testing.assert_array_equal(lib["plan"], restated["plan"])
            assert lib["iterations"] == restated["iterations"] and lib["chosen"] == restated["chosen"]
    run(base, x0=np.array([np.nan, 0.0, 0.0]))
    run(base, x0=np.array([1e30, 0.0, 0.0]))
    run(np.ascontiguousarray(base[:, :1]))              # one step
    run(base, iterations=0)                             # the refinement switched off: the LQ plan
    assert run(base, iterations=0)["iterations"] == 0
