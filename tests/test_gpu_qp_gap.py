"""How good is the plan the SHIPPED sampling schedule finds?  On the 28 scenarios of the reference's own MPC exercise
(/root/reference/src/acmpc/tests/test_spatial_mpc.py:45-75: hairpin / chicane / curve / straight x 7 parameters, horizon
100, road width 100 m) the controller runs with its defaults - 16 384 candidates, a cold solve = 6 exploring rounds from
(3 m/s, 5e-2 1/m), a warm one = 2 refining rounds at (0.5 m/s, 1e-3 1/m) - and the plan's value of the reference's QP
objective 1/2 z'Pz + q'z (the QP assembled by the oracle's restatement of control.py, pinned to the reference's own
assembly by G4/G5) is compared with the optimum `oracle.osqp_restated` finds at 1e-5.

Not parity (QP solutions are unpinned: OSQP is absent) - a quantified sanity bound.  gap = (J_plan - J_qp) / (|J_qp| + 1),
the measure tests/test_gpu_controller.py uses; J carries the constant -1/2 z_ref'P z_ref, which dominates |J_qp|, so the
same excess is also held against the optimum's TRACKING cost 1/2 (z - z_ref)'P(z - z_ref) - the stricter reading.  The
distribution is printed (pytest -s) and quoted in DESIGN.md section 6."""
import copy
import json
import os

import numpy as np
import pytest

import acmpc_oracle as orc
from test_gpu_controller import REFERENCE_SCRIPT_CONFIG
from test_support import PlaceholderVehicle

pytestmark = pytest.mark.gpu

FAMILIES = [("hairpin", np.linspace(10, 100, 7), -np.pi / 6), ("chicane", np.linspace(40, 100, 7), 0.1),
            ("curve", np.linspace(-0.02, 0.02, 7), 0.1), ("straight", np.linspace(40, 200, 7), 0.1)]
# what the shipped schedule holds on these scenarios (measured round 3, before the LQ plan: cold max 0.0062, warm max 0.0056,
# medians 4e-4; excess over the optimum's tracking cost: median 0.25-0.3, worst 15x on hairpin(100), whose optimum tracks to 0.13)
COLD_MAX, WARM_MAX, WARM_MEDIAN_MAX = 0.02, 0.015, 0.002
# round 4: the last round of a warm solve holds the LQ plan (csrc/acmpc_lq.h) - the QP's optimum wherever no box row is
# active - so the excess over the optimum's TRACKING cost has a bound on every quantile, the worst case included
# (measured: median 0.000, p90 0.007, worst 0.011 for the plan alone)
TRACKING_MEDIAN_MAX, TRACKING_P90_MAX, TRACKING_WORST_MAX = 0.05, 0.3, 1.0


def _qp_optimum(mpc, cfg):
    """The reference's QP for the controller's current path, solved by the oracle's restated OSQP at the tightest of
    1e-5 / 1e-4 / 1e-3 that converges (one nearly straight scenario is feasible only to ~1e-4: the reference pins t_0 = 0
    while boxing t >= 0.01, control.py:134 vs :67, and the solver's tolerance has to absorb what is left of that)."""
    n = cfg["horizon"] - 1
    path = mpc.reference_path
    x0 = mpc.model.t2s(path.get_state(0), np.array([0.0, 0.0, np.pi / 2]))
    qp = orc.control_qp(x0, path.table, cfg, mpc.model)
    n_eq = 3 * (n + 1)   # drop x_0's box rows (see above)
    keep = np.r_[0:n_eq, n_eq + 3:len(qp["l"])]
    for eps in (1e-5, 1e-4, 1e-3):
        ref = orc.osqp_restated(qp["P_diag"], qp["q"], qp["A"][keep], qp["l"][keep], qp["u"][keep], max_iter=6000,
                                eps_abs=eps, eps_rel=eps, adaptive_rho=True)
        if ref.info.status == "solved":
            break
    assert ref.info.status == "solved"
    weighted = qp["P_diag"] > 0
    # J = 1/2 z'Pz + q'z is the tracking cost 1/2 (z - z_ref)'P(z - z_ref) less this constant (q = -P z_ref)
    constant = 0.5 * float(np.sum(qp["q"][weighted] ** 2 / qp["P_diag"][weighted]))
    return dict(qp=qp, x0=x0, j_qp=orc.qp_objective(qp["P_diag"], qp["q"], ref.x), constant=constant, eps=eps)


def _plan_objective(mpc, cfg, problem):
    path = mpc.reference_path
    u = np.stack([mpc.projected_control[0], np.tan(mpc.projected_control[1]) / mpc.model.length], axis=1)
    lo, hi = orc.input_box(mpc.model)
    coef = orc.coefficients_spatial(path.table, mpc.model.margin).astype(np.float64)
    _, viol, X = orc.rollout_spatial(problem["x0"], coef, u[None], cfg["step_cost"], cfg["r_term"], cfg["final_cost"], lo,
                                     hi, 0.0, dtype=np.float64, return_states=True)
    qp = problem["qp"]
    return orc.qp_objective(qp["P_diag"], qp["q"], orc.pack_decision_vector(X[0], u)), float(viol[0])


def test_shipped_schedule_against_the_qp_optimum():
    from acmpc_amd import workloads
    from acmpc_amd.mpc import build_mpc
    from acmpc_amd.sampling_solver import COLD_ROUNDS, DEFAULT_CANDIDATES, DEFAULT_ROUNDS
    rows = []
    for kind, parameters, angle in FAMILIES:
        for parameter in parameters:
            cfg = copy.deepcopy(REFERENCE_SCRIPT_CONFIG)
            mpc = build_mpc(cfg, PlaceholderVehicle())        # a fresh controller: the first solve is a cold one
            solver = mpc._control_solver
            assert (solver._n_candidates, solver._rounds, solver._cold_rounds) == (DEFAULT_CANDIDATES, DEFAULT_ROUNDS, COLD_ROUNDS)
            path_in = workloads.family_path(kind, float(parameter), cfg["horizon"], angle=angle, width=100.0)
            plans = []
            for solve in range(5):                            # cold, then four warm-started solves of the same pose
                mpc.get_control(path_in, offset=0.0)
                assert mpc.infeasibility_counter == 0, "%s(%g) infeasible at solve %d" % (kind, parameter, solve)
                if solve == 0:
                    problem = _qp_optimum(mpc, cfg)           # (the path, hence the QP, is the same in all five)
                if solve in (0, 1, 4):
                    plans.append(_plan_objective(mpc, cfg, problem))
            j_qp, scale = problem["j_qp"], abs(problem["j_qp"]) + 1.0
            rows.append({"scenario": "%s(%.4g)" % (kind, parameter), "cold": (plans[0][0] - j_qp) / scale,
                         "warm_1": (plans[1][0] - j_qp) / scale, "warm_4": (plans[2][0] - j_qp) / scale,
                         # the same excess as a fraction of the optimum's TRACKING cost 1/2 (z - z_ref)'P(z - z_ref)
                         "warm_4_of_tracking_cost": (plans[2][0] - j_qp) / (j_qp + problem["constant"] + 1e-12),
                         "J_qp": j_qp, "tracking_cost_qp": j_qp + problem["constant"], "qp_eps": problem["eps"],
                         "violation_warm_4": plans[2][1]})
    cold = np.array([r["cold"] for r in rows])
    warm1 = np.array([r["warm_1"] for r in rows])
    warm4 = np.array([r["warm_4"] for r in rows])
    print("\n%-18s %10s %10s %10s %12s %12s %8s" % ("scenario", "cold", "warm 1", "warm 4", "J_qp", "tracking", "w4/trk"))
    for r in rows:
        print("%-18s %10.5f %10.5f %10.5f %12.4f %12.5f %8.4f" % (r["scenario"], r["cold"], r["warm_1"], r["warm_4"], r["J_qp"],
                                                              r["tracking_cost_qp"], r["warm_4_of_tracking_cost"]))
    summary = {name: {"median": float(np.median(v)), "p90": float(np.percentile(v, 90)), "max": float(v.max()),
                      "min": float(v.min())} for name, v in (("cold", cold), ("warm_1", warm1), ("warm_4", warm4))}
    print(json.dumps(summary))
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "qp_gap.json"), "w") as handle:
            json.dump({"summary": summary, "rows": rows}, handle, indent=1)
    # a feasible rollout cannot beat the QP optimum by more than the solvers' tolerances
    assert min(cold.min(), warm1.min(), warm4.min()) > -1e-2
    assert cold.max() <= COLD_MAX, summary
    assert warm1.max() <= WARM_MAX and warm4.max() <= WARM_MAX, summary
    assert np.median(warm4) <= WARM_MEDIAN_MAX, summary
    of_tracking = np.array([r["warm_4_of_tracking_cost"] for r in rows])
    print("warm 4 excess over the optimum's tracking cost: median %.3f p90 %.3f max %.3f"
          % (np.median(of_tracking), np.percentile(of_tracking, 90), of_tracking.max()))
    assert np.median(of_tracking) <= TRACKING_MEDIAN_MAX, of_tracking
    assert np.percentile(of_tracking, 90) <= TRACKING_P90_MAX, of_tracking
    assert of_tracking.max() <= TRACKING_WORST_MAX, of_tracking
