"""The speed-profile QP solved exactly, without iterating (`acmpc_speed_profile_exact`, csrc/acmpc_admm.h exact_profile): the
library's host code against its restatement in the oracle (bit for bit), against the optimum of the QP the REFERENCE
assembled (golden A, l, u, q of speed_profile.py:26-59; optimum = the native splitting and the oracle's restated OSQP at
1e-10), its optimality checked solver-independently (feasible, and no feasible point above it), and the cases it hands
back to the splitting.  No GPU work."""
import numpy as np
import pytest

import acmpc_oracle as orc
from test_support import RACING

KEYS = ["monza_H50_chicane_90", "monza_H50_hairpin_10", "spa_H50_chicane_70", "nordschleife_H80_chicane_70",
        "silverstone_H50_chicane_70", "monza_H50_straight_146.667", "monza_H20_hairpin_10"]


def _golden_problem(golden, key, tag):
    A, l, u, q = (golden[key + "/" + tag + k] for k in ("A", "l", "u", "q"))
    n = A.shape[1]
    ds = 1.0 / (2.0 * A[np.arange(n - 1), np.arange(n - 1) + 1])      # the rows are (v[i+1] - v[i]) / (2 ds[i])
    return A, l, u, q, n, ds


@pytest.mark.parametrize("tag", ["sp0_", "sp1_"])
@pytest.mark.parametrize("key", KEYS)
def test_the_sweeps_give_the_optimum_of_the_reference_assembled_qp(golden, golden_cases, key, tag):
    from acmpc_amd import _capi
    if key not in golden_cases:
        pytest.skip("no such golden case")
    A, l, u, q, n, ds = _golden_problem(golden, key, tag)
    v_hi = u[n - 1:]
    np.testing.assert_array_equal(q, -v_hi)                            # the objective pulls towards the upper bound itself
    a_min, a_max, v_min = float(l[0]), float(u[0]), float(l[n - 1])
    swept = _capi.speed_profile_exact(v_hi, np.append(ds, 1.0), a_min, a_max, v_min)
    want = orc.speed_profile_exact(v_hi, np.append(ds, 1.0), a_min, a_max, v_min)
    if want is None:                                                    # an infeasible golden problem: handed back
        assert swept is None
        return
    v, y = swept
    np.testing.assert_array_equal(v, want)                             # the library's sweeps == the restatement's
    assert not y.any()
    Av = A @ v
    assert (Av >= l - 1e-9).all() and (Av <= u + 1e-9).all()           # feasible for the reference's rows
    # the optimum: the splitting at a tight tolerance, native and restated
    x, _, status, _ = _capi.speed_profile_qp(v_hi, np.append(ds, 1.0), a_min, a_max, v_min, max_iter=400000, eps_abs=1e-10,
                                             eps_rel=1e-10)
    assert status == "solved"
    np.testing.assert_allclose(v, x, atol=2e-6)
    ref = orc.osqp_restated(np.ones(n), q, A, l, u, max_iter=400000, eps_abs=1e-10, eps_rel=1e-10, check_every=50)
    if ref.info.status == "solved":
        np.testing.assert_allclose(v, ref.x, atol=2e-6)
        # (the iterate is feasible to its tolerance only, which buys it a few 1e-5 of objective)
        f_swept, f_ref = orc.qp_objective(np.ones(n), q, v), orc.qp_objective(np.ones(n), q, ref.x)
        assert f_swept <= f_ref + 1e-8 * abs(f_ref)
    # and where the splitting stops at the controller's default tolerances (1e-3, OSQP's): up to 2.5 m/s from the optimum on
    # these problems (3 % of 84 m/s), on either side of it - rows are met to the tolerance only
    loose, _, status, iters = _capi.speed_profile_qp(v_hi, np.append(ds, 1.0), a_min, a_max, v_min, max_iter=4000)
    assert status == "solved" and iters > 0 and np.abs(loose - v).max() < 3.0


def test_every_golden_problem(golden, golden_cases):
    """All 52 x 2 problems the reference assembled (both solvers): the library's sweeps equal the restatement bit for bit,
    meet every row of the reference's A, l, u, and sit on the ceiling wherever no rate row holds them off it."""
    from acmpc_amd import _capi
    solved = 0
    for key in golden_cases:
        for tag in ("sp0_", "sp1_"):
            A, l, u, q, n, ds = _golden_problem(golden, key, tag)
            v_hi = u[n - 1:]
            a_min, a_max, v_min = float(l[0]), float(u[0]), float(l[n - 1])
            swept = _capi.speed_profile_exact(v_hi, np.append(ds, 1.0), a_min, a_max, v_min)
            want = orc.speed_profile_exact(v_hi, np.append(ds, 1.0), a_min, a_max, v_min)
            assert (swept is None) == (want is None), (key, tag)
            if swept is None:
                continue
            solved += 1
            v = swept[0]
            np.testing.assert_array_equal(v, want, err_msg=key + tag)
            Av = A @ v
            assert (Av >= l - 1e-9).all() and (Av <= u + 1e-9).all(), (key, tag)
            rate = Av[:n - 1]
            free = np.ones(n, dtype=bool)          # points no active rate row touches must be ON the ceiling
            free[1:] &= ~(rate >= a_max - 1e-9)
            free[:-1] &= ~(rate <= a_min + 1e-9)
            np.testing.assert_array_equal(v[free], v_hi[free])
    assert solved >= len(golden_cases)             # (most problems of the fixture are feasible)


def test_third_party_solvers_agree(golden, golden_cases):
    """Independent of everything written here: SciPy's HiGHS maximising the sum of v over the REFERENCE-assembled rows returns
    the same profile (the pointwise largest feasible one is the unique maximiser of any positive weighting) on every golden
    problem, to 1e-9; and SciPy's generic constrained minimiser (trust-constr, an interior-point method) run on the QP itself
    - 1/2 |v|^2 + q'v under the same rows - ends 1e-3 m/s from it with an objective that is not lower."""
    from scipy.optimize import Bounds, LinearConstraint, linprog, minimize
    from acmpc_amd import _capi
    checked = 0
    for key in golden_cases:
        for tag in ("sp0_", "sp1_"):
            A, l, u, q, n, ds = _golden_problem(golden, key, tag)
            swept = _capi.speed_profile_exact(u[n - 1:], np.append(ds, 1.0), float(l[0]), float(u[0]), float(l[n - 1]))
            if swept is None:
                continue
            D = A[:n - 1]
            lp = linprog(-np.ones(n), A_ub=np.vstack([D, -D]), b_ub=np.concatenate([u[:n - 1], -l[:n - 1]]),
                         bounds=list(zip(l[n - 1:], u[n - 1:])), method="highs")
            assert lp.status == 0
            np.testing.assert_allclose(lp.x, swept[0], rtol=0, atol=1e-9, err_msg=key + tag)
            checked += 1
    assert checked >= len(golden_cases)
    for key in ("monza_H50_chicane_90", "nordschleife_H80_chicane_70"):
        A, l, u, q, n, ds = _golden_problem(golden, key, "sp0_")
        v = _capi.speed_profile_exact(u[n - 1:], np.append(ds, 1.0), float(l[0]), float(u[0]), float(l[n - 1]))[0]
        objective = lambda x: 0.5 * x @ x + q @ x
        found = minimize(objective, np.clip(0.5 * u[n - 1:], l[n - 1:], u[n - 1:]), jac=lambda x: x + q, hess=lambda x: np.eye(n),
                         method="trust-constr", constraints=[LinearConstraint(A[:n - 1], l[:n - 1], u[:n - 1])],
                         bounds=Bounds(l[n - 1:], u[n - 1:]), options=dict(gtol=1e-10, xtol=1e-12, maxiter=3000))
        assert np.abs(found.x - v).max() < 1e-2 and objective(v) <= objective(found.x) + 1e-9


def test_no_feasible_profile_lies_above_the_swept_one():
    """Solver-independent: the swept profile is feasible, and every feasible profile is below it in every coordinate - so it
    minimises |v - v_hi|^2 term by term.  Random problems; feasible points from random starts pushed into the set by the same
    kind of sweeps from below."""
    from acmpc_amd import _capi
    rng = np.random.default_rng(3)
    for case in range(200):
        n = int(rng.integers(3, 120))
        ds = rng.uniform(0.2, 4.0, n)
        v_hi = rng.uniform(9.0, 90.0, n)
        a_min, a_max, v_min = -rng.uniform(0.1, 20.0), rng.uniform(0.0, 10.0), rng.uniform(0.0, 8.0)
        swept = _capi.speed_profile_exact(v_hi, ds, a_min, a_max, v_min)
        want = orc.speed_profile_exact(v_hi, ds, a_min, a_max, v_min)
        assert (swept is None) == (want is None)
        if swept is None:
            continue
        v = swept[0]
        np.testing.assert_array_equal(v, want)
        rate = np.diff(v) / (2.0 * ds[:-1])
        assert (v <= v_hi).all() and (v >= v_min).all() and (rate <= a_max + 1e-12).all() and (rate >= a_min - 1e-12).all()
        for _ in range(5):     # some feasible point: a random profile under v, made rate-feasible from below
            w = np.minimum(rng.uniform(v_min, v_hi), v_hi)
            for i in range(n - 1):
                w[i + 1] = min(w[i + 1], w[i] + 2.0 * ds[i] * a_max)
            for i in range(n - 2, -1, -1):
                w[i] = min(w[i], w[i + 1] - 2.0 * ds[i] * a_min)
            if (w >= v_min).all():
                assert (w <= v + 1e-12).all()


def test_what_the_sweeps_hand_back():
    from acmpc_amd import _capi
    ds = np.full(6, 2.0)
    v_hi = np.array([30.0, 30.0, 30.0, 12.0, 30.0, 30.0])
    assert _capi.speed_profile_exact(v_hi, ds, -1.0, 1.0, 8.0) is not None
    assert _capi.speed_profile_exact(v_hi, ds, -1.0, 1.0, 14.0) is None            # the slow point is below v_min: infeasible
    assert _capi.speed_profile_exact(v_hi, ds, 0.5, 1.0, 8.0) is None              # a_min > 0: not the sweeps' shape
    assert _capi.speed_profile_exact(v_hi, ds, -1.0, -0.5, 8.0) is None
    bad = v_hi.copy()
    bad[2] = np.nan
    assert _capi.speed_profile_exact(bad, ds, -1.0, 1.0, 8.0) is None and orc.speed_profile_exact(bad, ds, -1.0, 1.0, 8.0) is None
    gap = ds.copy()
    gap[1] = 0.0
    assert _capi.speed_profile_exact(v_hi, gap, -1.0, 1.0, 8.0) is None and orc.speed_profile_exact(v_hi, gap, -1.0, 1.0, 8.0) is None
    with pytest.raises(_capi.EngineError):
        _capi.speed_profile_exact(v_hi[:1], ds[:1], -1.0, 1.0, 8.0)


def test_the_host_solver_object_sweeps_by_default_and_iterates_on_request(golden):
    from acmpc_amd.reference_path import ReferencePath
    from acmpc_amd.speed_profile import SpeedProfileSolver
    key = "monza_H50_chicane_90"
    cons = RACING["monza"]["speed_profile_constraints"]
    path = ReferencePath.from_table(golden[key + "/table_no_v"])
    base = {"control_horizon": len(path), "max_iterations": 4000, "constraints": cons}
    exact = SpeedProfileSolver(base).solve(path, cons["end_velocity"])
    admm = SpeedProfileSolver(dict(base, method="admm")).solve(path, cons["end_velocity"])
    assert exact.info.status == "solved" and exact.info.iter == 0
    assert admm.info.status == "solved" and admm.info.iter > 0
    assert np.abs(exact.x - admm.x).max() < 3.0
    with pytest.raises(ValueError):
        SpeedProfileSolver(dict(base, method="osqp"))
