#!/usr/bin/env python3
"""Frozen outputs of the BUILD-DEFINED composition (SURVEY.md 8c, G12 "end-to-end CPU-restatement outputs"):
sample -> rollout -> cost -> argmin of oracle/acmpc_oracle.py on the BASELINE shapes, at reduced N for file size,
plus SHA-256 digests of the full-N cost vectors (computed with the oracle's C restatement, which
tests/test_oracle_c_vs_numpy.py holds bit-identical to the NumPy one).

The reference has no such path, so these vectors do not come from it; they pin the specification ("spec order",
DESIGN.md section 2) against accidental drift: any change to the oracle, the C restatement or the kernels that moves
one bit of a finite cost shows up here.  Inputs are stored in the file; the full-N cases regenerate theirs from the
seeds of tests/test_support.py::make_problem.

    python tests/golden/gen_composition.py        # rewrites tests/golden/composition.npz
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import acmpc_oracle as orc  # noqa: E402
import c_oracle  # noqa: E402
from test_support import full_size_controls, make_problem  # noqa: E402

# (name, track, H, N stored, mode, nn_window)
CASES = [
    ("config1_monza_H20_S", "monza", 20, 128, 0, None),          # BASELINE configs[0] at full size
    ("config1_monza_H20_T", "monza", 20, 128, 1, None),
    ("config2_monza_H50_S", "monza", 50, 512, 0, None),          # configs[1] shape, reduced N
    ("config2_monza_H50_T", "monza", 50, 256, 1, None),
    ("config2_monza_H50_T_window", "monza", 50, 256, 1, (2, 5)),
    ("config3_spa_H50_S", "spa", 50, 512, 0, None),              # configs[2] shape, reduced N
    ("config4_nordschleife_H80_S", "nordschleife", 80, 256, 0, None),
]
# (name, track, H, N, mode, nn_window, seed): digests only
FULL = [
    ("full_config2_monza_H50_N4096_S", "monza", 50, 4096, 0, None, 4242),
    ("full_config3_spa_H50_N65536_S", "spa", 50, 65536, 0, None, 4242),
    ("full_config3_spa_H50_N65536_T_window", "spa", 50, 65536, 1, (2, 5), 4242),
    ("full_config4_nordschleife_H80_N262144_S", "nordschleife", 80, 262144, 0, None, 4242),
]


# (name, mode, nn_window, seed): ONE WHOLE SOLVE at the closed-loop shape - 1 problem x 16 384 candidates x horizon 50, two
# rounds at the controller's warm spread, the LQ plan as the last round's candidate 2 - by oracle.optimize_restated: what
# acmpc_optimize / acmpc_control_tick must return bit for bit (tests/test_gpu_restated_solve.py)
SOLVES = [
    ("solve_monza_H50_S", 0, None, 0x5EED0001),
    ("solve_monza_H50_T", 1, None, 0x5EED0002),
    ("solve_monza_H50_T_window", 1, (2, 5), 0x5EED0003),
]
SOLVE_N, SOLVE_ROUNDS, SOLVE_SIGMA, SOLVE_SHRINK = 16384, 2, (0.5, 1.0e-3), 0.5


def solves(out):
    for name, mode, window, seed in SOLVES:
        prob = make_problem(orc, "monza", 50, 4, seed=8100 + mode)
        cfg = prob["cfg"]
        margin = prob["limits"].margin
        coef = orc.coefficients_spatial(prob["table"], margin) if mode == 0 else orc.coefficients_temporal(prob["table"], margin)
        start = prob["x0"] if mode == 0 else prob["pose0"]
        u_ref = np.stack([prob["table"][orc.ROW_V], prob["table"][orc.ROW_KAPPA]], axis=1).astype(np.float32)
        centre = (u_ref + np.array([-0.6, 0.0015], dtype=np.float32)).astype(np.float32)
        frenet = start.astype(np.float64) if mode == 0 else orc.frenet_start(prob["table"], start.astype(np.float64))
        plan = orc.lq_plan(prob["table"], frenet, cfg["step_cost"], cfg["r_term"], cfg["final_cost"], prob["u_lo"], prob["u_hi"])
        won = orc.optimize_restated(mode, start, coef, centre, u_ref, SOLVE_N, SOLVE_ROUNDS, SOLVE_SIGMA, SOLVE_SHRINK, seed,
                                    cfg["step_cost"], cfg["r_term"], cfg["final_cost"], prob["u_lo"], prob["u_hi"], 1.0e6, 0.05,
                                    window, extra=plan)
        k = name + "/"
        out[k + "table"], out[k + "start"], out[k + "centre"], out[k + "u_ref"], out[k + "coef"] = (
            prob["table"], start, centre, u_ref, coef)
        out[k + "weights"] = np.array(list(cfg["step_cost"]) + list(cfg["r_term"]) + list(cfg["final_cost"]))
        out[k + "box"] = np.array(list(prob["u_lo"]) + list(prob["u_hi"]))
        out[k + "margin"], out[k + "mode"] = np.array(margin), np.array(mode)
        out[k + "window"] = np.array(window if window is not None else (-1, -1))
        out[k + "n_candidates"], out[k + "rounds"] = np.array(SOLVE_N), np.array(SOLVE_ROUNDS)
        out[k + "sigma"], out[k + "shrink"], out[k + "seed"] = np.array(SOLVE_SIGMA), np.array(SOLVE_SHRINK), np.array(seed)
        out[k + "plan"] = plan
        out[k + "cost"], out[k + "violation"] = np.array(won["cost"]), np.array(won["violation"])
        out[k + "n_feasible"], out[k + "index"] = np.array(won["n_feasible"]), np.array(won["index"])
        out[k + "winners"] = np.array(won["winners"])
        out[k + "u"], out[k + "x"] = won["u"], won["x"]
        print("%s: winners %s cost %.6f" % (name, won["winners"], won["cost"]))


def main():
    out = {"cases": np.array([c[0] for c in CASES]), "full_cases": np.array([c[0] for c in FULL]),
           "solves": np.array([c[0] for c in SOLVES])}
    solves(out)
    for name, track, H, N, mode, window in CASES:
        prob = make_problem(orc, track, H, N, seed=7000 + H + N + mode)
        cfg = prob["cfg"]
        args = (prob["U"], cfg["step_cost"], cfg["r_term"], cfg["final_cost"], prob["u_lo"], prob["u_hi"], 1.0e6)
        if mode == 0:
            coef = orc.coefficients_spatial(prob["table"], prob["limits"].margin)
            start = prob["x0"]
            cost, viol, X = orc.rollout_spatial(start, coef, *args, dtype=np.float32, return_states=True)
        else:
            coef = orc.coefficients_temporal(prob["table"], prob["limits"].margin)
            start = prob["pose0"]
            cost, viol, X, _ = orc.rollout_temporal(start, coef, *args, 0.05, dtype=np.float32, return_states=True,
                                                    nn_window=window)
        best, best_cost = orc.pick_best(cost)
        k = name + "/"
        out[k + "table"], out[k + "start"], out[k + "U"] = prob["table"], start, prob["U"]
        out[k + "weights"] = np.array(list(cfg["step_cost"]) + list(cfg["r_term"]) + list(cfg["final_cost"]))
        out[k + "box"] = np.array(list(prob["u_lo"]) + list(prob["u_hi"]))
        out[k + "margin"] = np.array(prob["limits"].margin)
        out[k + "mode"] = np.array(mode)
        out[k + "window"] = np.array(window if window is not None else (-1, -1))
        out[k + "coef"] = coef
        out[k + "cost"], out[k + "violation"] = cost, viol
        out[k + "best"], out[k + "best_cost"] = np.array(best), np.array(best_cost, dtype=np.float32)
        out[k + "best_u"], out[k + "best_x"] = prob["U"][best], X[best]
    for name, track, H, N, mode, window, seed in FULL:
        n = H - 1
        prob = make_problem(orc, track, H, 16, seed=seed)
        U = full_size_controls(orc, prob, N, n)
        cfg = prob["cfg"]
        w = c_oracle.make_weights(cfg["step_cost"], cfg["r_term"], cfg["final_cost"], prob["u_lo"], prob["u_hi"], 1.0e6,
                                  nn_window=window)
        if mode == 0:
            coef, start = orc.coefficients_spatial(prob["table"], prob["limits"].margin), prob["x0"]
        else:
            coef, start = orc.coefficients_temporal(prob["table"], prob["limits"].margin), prob["pose0"]
        cost, viol = c_oracle.rollout(mode, start, coef, U, 0, w)
        k = name + "/"
        out[k + "spec"] = np.array([H, N, mode, seed] + list(window if window is not None else (-1, -1)))
        out[k + "track"] = np.array(track)
        out[k + "controls_sha256"] = np.array(hashlib.sha256(U.tobytes()).hexdigest())
        out[k + "cost_sha256"] = np.array(hashlib.sha256(cost.tobytes()).hexdigest())
        out[k + "best"] = np.array(orc.pick_best(cost)[0])
        out[k + "n_feasible"] = np.array(int(np.count_nonzero(viol == 0)))
    path = os.path.join(HERE, "composition.npz")
    np.savez_compressed(path, **out)
    print("wrote %s: %d arrays, %.1f kB" % (path, len(out), os.path.getsize(path) / 1e3))


if __name__ == "__main__":
    main()
