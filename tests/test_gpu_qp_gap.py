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
# round 5: thresholds at <= 5x what was measured (they were ~90x: a 50-fold regression passed)
TRACKING_MEDIAN_MAX, TRACKING_P90_MAX, TRACKING_WORST_MAX = 0.001, 0.03, 0.05


def _qp_optimum(mpc, cfg, offset=0.0, max_iter=6000):
    """The reference's QP for the controller's current path, solved by the oracle's restated OSQP at the tightest of
    1e-5 / 1e-4 / 1e-3 that converges (one nearly straight scenario is feasible only to ~1e-4: the reference pins t_0 = 0
    while boxing t >= 0.01, control.py:134 vs :67, and the solver's tolerance has to absorb what is left of that)."""
    n = cfg["horizon"] - 1
    path = mpc.reference_path
    x0 = mpc.model.t2s(path.get_state(0), np.array([offset, 0.0, np.pi / 2]))
    qp = orc.control_qp(x0, path.table, cfg, mpc.model)
    n_eq = 3 * (n + 1)   # drop x_0's box rows (see above)
    keep = np.r_[0:n_eq, n_eq + 3:len(qp["l"])]
    for eps in (1e-5, 1e-4, 1e-3):
        ref = orc.osqp_restated(qp["P_diag"], qp["q"], qp["A"][keep], qp["l"][keep], qp["u"][keep], max_iter=max_iter,
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


# ---------------------------------------------------------------------------------------------------------------------
# Round 5: the racing configuration, where the QP's box rows are ACTIVE.  The 28 scenarios above drive a 100 m wide road:
# no corridor row binds and the LQ plan is the optimum.  The control process hands over widths linspace(10, 6, H)
# (/root/reference/src/acmpc/control/controller.py:256-267) at the per-track weights of configs/<track>.yaml:67-81, horizon
# 50; then the corridor e_y in +-(w/2 - margin) (control/solvers/control.py:57-60) and the input box (control.py:130-139)
# are what make the problem a QP.
# ---------------------------------------------------------------------------------------------------------------------
def _racing_scenarios():
    """(name, track, H x 3 path, offset, extra) - extra: dict(build_v_max=..) builds the model with that speed limit."""
    from acmpc_amd import workloads as wl
    H = 50
    out = []
    # (a) input saturation: the reference script's hairpins under the racing corridor ...
    for r in (9.0, 10.0, 12.0, 15.0):
        out.append(("hairpin(%g)" % r, "monza", wl.racing_widths(wl.family_path("hairpin", r, H)), 0.0, {}))
    out.append(("hairpin(15) entered at 30 deg", "monza", wl.racing_widths(wl.family_path("hairpin", 15.0, H, angle=-np.pi / 6)), 0.0, {}))
    # ... and corners at or inside the steering limit (kappa_max = tan(0.30) / 2.65 = 0.1167: r_min 8.57 m) entered from a
    # straight - the optimum prepares steps ahead; the clipped LQ plan's excess here is 2x - 15x, or it leaves the corridor
    for track, r, lead in (("monza", 8.0, 30.0), ("monza", 7.5, 60.0), ("monza", 9.0, 15.0), ("spa", 8.0, 30.0), ("spa", 7.5, 15.0),
                           ("nordschleife", 8.0, 30.0), ("nordschleife", 7.5, 60.0), ("silverstone", 8.0, 15.0),
                           ("silverstone", 7.5, 30.0)):
        out.append(("straight %g m + corner r %g" % (lead, r), track, wl.corner_entry_path(r, lead, H), 0.0, {}))
    for track, r in (("monza", 8.0), ("spa", 10.0), ("nordschleife", 8.0)):
        out.append(("s-bend r %g" % r, track, wl.s_bend_path(r, 20.0, H), 0.0, {}))
    # (b) corridor: from 2 / 3 / 3.8 m off the centre line (the corridor is +-4.03 m at the car, +-2.03 m at the horizon)
    for off in (-3.8, -3.0, -2.0, 2.0, 3.0, 3.8):
        out.append(("straight 150 m from %+g m" % off, "silverstone" if off < 0 else "nordschleife",
                    wl.racing_widths(wl.family_path("straight", 150.0, H)), off, {}))
    for track, r, off in (("monza", 8.0, -2.0), ("monza", 8.0, 2.0), ("spa", 10.0, 2.0), ("nordschleife", 8.0, -2.0),
                          ("silverstone", 10.0, -2.0)):
        out.append(("straight 30 m + corner r %g from %+g m" % (r, off), track, wl.corner_entry_path(r, 30.0, H), off, {}))
    # (c) speed box: the model built with v_max = 20 m/s (input box 20.1) under a live reference speed of 28 - the speed
    # profile, hence u_ref, lies above the box along the straights
    for kind, parameter in (("straight", 150.0), ("hairpin", 40.0)):
        out.append(("%s(%g), speed box 20 under a 28 m/s profile" % (kind, parameter), "monza",
                    wl.racing_widths(wl.family_path(kind, parameter, H)), 1.0, {"build_v_max": 20.0}))
    return out


# measured (round 5, 30 scenarios, 19 of them with the refinement triggered): warm-4 excess median 1.1e-4, p90 0.0065, worst
# 0.0113 (the restated OSQP's own slack at 1e-5: its equality rows are met to the tolerance, a plan's exactly), least -0.0032;
# cold (first solve of a fresh controller) median 1.8e-5, p90 0.0065, worst 0.0113; the returned plan's summed squared
# state-row excess at most 1.5e-7 (the t >= 0.01 row of x_1 at 32 m/s, control.py:134 vs :67).  Thresholds <= 5x measured
RACING_EXCESS_MEDIAN_MAX, RACING_EXCESS_P90_MAX, RACING_EXCESS_WORST_MAX = 0.0004, 0.03, 0.055
RACING_VIOLATION_WORST_MAX = 7.5e-7     # sum of squared state-row excess of the returned plan [m^2 / s^2]


def test_racing_configuration_with_active_box_rows():
    """Per scenario: a fresh controller (cold solve, then four warm ones of the same pose, the live v_max rewritten
    before every solve as the control process does, controller.py:241-243); the returned plan's tracking cost against
    the restated-OSQP optimum's, and the plan's OWN summed squared box-row excess - the reference returns a feasible
    dec.x or keeps the previous plan (spatial_mpc.py:193-217)."""
    import time
    from acmpc_amd.mpc import build_mpc
    from test_support import RACING
    rows = []
    for name, track, path_in, offset, extra in _racing_scenarios():
        cfg = copy.deepcopy(RACING[track])
        live_v_max = float(cfg["unlocalised_max_speed"])
        if "build_v_max" in extra:
            cfg["speed_profile_constraints"]["v_max"] = extra["build_v_max"]
        mpc = build_mpc(cfg, PlaceholderVehicle())
        assert mpc._control_solver._lq_candidate == 2      # the shipped default
        failures, plans, stats, took, problem = 0, [], [], [], None
        for solve in range(5):
            mpc.speed_profile_constraints["v_max"] = live_v_max
            start = time.perf_counter()
            mpc.get_control(path_in, offset=offset)
            took.append(time.perf_counter() - start)
            failures += mpc.infeasibility_counter > 0
            stats.append(mpc._control_solver._engine.lq_box_stats())
            if mpc.infeasibility_counter == 0 and solve in (0, 4):
                if problem is None:
                    problem = _qp_optimum(mpc, cfg, offset, max_iter=20000)
                plans.append(_plan_objective(mpc, cfg, problem))
        assert failures == 0 and len(plans) == 2, "%s/%s: %d of 5 solves rejected" % (track, name, failures)
        tracking = problem["j_qp"] + problem["constant"]
        rows.append({"scenario": "%s: %s" % (track, name), "tracking_cost_qp": tracking, "qp_eps": problem["eps"],
                     "cold_excess": (plans[0][0] - problem["j_qp"]) / tracking, "cold_violation": plans[0][1],
                     "warm_4_excess": (plans[1][0] - problem["j_qp"]) / tracking, "warm_4_violation": plans[1][1],
                     "box_iterations": [s["iterations"] for s in stats], "box_chosen": [s["chosen"] for s in stats],
                     "box_triggered": [int(s["triggered"]) for s in stats], "solve_us": [round(1e6 * t, 1) for t in took]})
    print("\n%-58s %9s %9s %9s %9s %9s  %s" % ("scenario", "track qp", "cold ex", "cold V", "warm4 ex", "warm4 V", "box iterations"))
    for r in rows:
        print("%-58s %9.4f %9.4f %9.1e %9.4f %9.1e  %s" % (r["scenario"], r["tracking_cost_qp"], r["cold_excess"],
                                                         r["cold_violation"], r["warm_4_excess"], r["warm_4_violation"],
                                                         r["box_iterations"]))
    excess = np.array([r["warm_4_excess"] for r in rows])
    violation = np.array([r["warm_4_violation"] for r in rows])
    cold = np.array([r["cold_excess"] for r in rows])
    summary = {"warm_4_excess": {"median": float(np.median(excess)), "p90": float(np.percentile(excess, 90)),
                                 "max": float(excess.max()), "min": float(excess.min())},
               "cold_excess": {"median": float(np.median(cold)), "p90": float(np.percentile(cold, 90)), "max": float(cold.max())},
               "warm_4_violation_max": float(violation.max()),
               "cold_violation_max": float(max(r["cold_violation"] for r in rows)),
               "triggered_scenarios": int(sum(any(r["box_triggered"]) for r in rows)), "scenarios": len(rows)}
    print(json.dumps(summary))
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "qp_gap_racing.json"), "w") as handle:
            json.dump({"summary": summary, "rows": rows}, handle, indent=1)
    assert summary["triggered_scenarios"] >= 15, summary     # the family does exercise the box rows
    # a plan that rolls out feasibly cannot beat the QP's optimum by more than the restated solver's own slack (~1 %)
    assert excess.min() > -0.03 and cold.min() > -0.03, summary
    assert np.median(excess) <= RACING_EXCESS_MEDIAN_MAX, summary
    assert np.percentile(excess, 90) <= RACING_EXCESS_P90_MAX, summary
    assert excess.max() <= RACING_EXCESS_WORST_MAX, summary
    assert violation.max() <= RACING_VIOLATION_WORST_MAX, summary


def test_an_infeasible_corridor_keeps_the_previous_plan():
    """A hairpin of radius 10 m entered at 30 degrees under the racing corridor has no feasible plan (the restated OSQP
    does not converge; every plan leaves the corridor by metres).  The reference then keeps its previous plan and counts
    the failure (spatial_mpc.py:212-217) - so does the controller, whatever candidate the box refinement offered."""
    from acmpc_amd import workloads as wl
    from acmpc_amd.mpc import build_mpc
    from test_support import RACING
    cfg = copy.deepcopy(RACING["monza"])
    mpc = build_mpc(cfg, PlaceholderVehicle())
    H = cfg["horizon"]
    mpc.speed_profile_constraints["v_max"] = 28.0
    mpc.get_control(wl.racing_widths(wl.family_path("hairpin", 40.0, H)), offset=0.5)
    assert mpc.infeasibility_counter == 0
    kept = mpc.projected_control.copy()
    for attempt in range(3):
        mpc.get_control(wl.racing_widths(wl.family_path("hairpin", 10.0, H, angle=-np.pi / 6)), offset=0.0)
        assert mpc.infeasibility_counter == attempt + 1
        np.testing.assert_array_equal(mpc.projected_control, kept)
    mpc.get_control(wl.racing_widths(wl.family_path("hairpin", 40.0, H)), offset=0.5)
    assert mpc.infeasibility_counter == 0
