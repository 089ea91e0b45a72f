"""Development tool (GPU box): the QP-gap distribution of tests/test_gpu_qp_gap.py for several sampling schedules.
usage: python3 tools/qp_gap_sweep.py   (edit SCHEDULES)"""
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("ac-mpc_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))

import numpy as np  # noqa: E402

import test_gpu_qp_gap as T  # noqa: E402
from test_gpu_controller import REFERENCE_SCRIPT_CONFIG  # noqa: E402
from test_support import PlaceholderVehicle  # noqa: E402
from acmpc_amd import workloads  # noqa: E402
from acmpc_amd.mpc import build_mpc  # noqa: E402

SCHEDULES = {
    "shipped": {},
    "cold 6 x (3, 0.05)": {"sampling_cold_rounds": 6, "sampling_cold_sigma": [3.0, 0.05]},
    "cold 8 x (3, 0.05)": {"sampling_cold_rounds": 8, "sampling_cold_sigma": [3.0, 0.05]},
    "cold 6 x (3, 0.05), warm 2 x (0.5, 4e-3)": {"sampling_cold_rounds": 6, "sampling_cold_sigma": [3.0, 0.05],
                                                  "sampling_sigma": [0.5, 4e-3]},
    "cold 6 x (3, 0.05), warm 3 x (1.0, 4e-3)": {"sampling_cold_rounds": 6, "sampling_cold_sigma": [3.0, 0.05],
                                                  "sampling_sigma": [1.0, 4e-3], "sampling_rounds": 3},
}
problems = {}
for name, extra in SCHEDULES.items():
    rows = []
    for kind, parameters, angle in T.FAMILIES:
        for parameter in parameters:
            cfg = copy.deepcopy(REFERENCE_SCRIPT_CONFIG)
            cfg.update(extra)
            mpc = build_mpc(cfg, PlaceholderVehicle())
            path_in = workloads.family_path(kind, float(parameter), cfg["horizon"], angle=angle, width=100.0)
            plans, failed = [], 0
            for solve in range(9):
                mpc.get_control(path_in, offset=0.0)
                failed += mpc.infeasibility_counter > 0
                if solve == 0:
                    key = (kind, float(parameter))
                    if key not in problems:
                        problems[key] = T._qp_optimum(mpc, cfg)
                    problem = problems[key]
                if solve in (0, 1, 4, 8):
                    plans.append(T._plan_objective(mpc, cfg, problem)[0])
            j_qp = problem["j_qp"]
            track = j_qp + problem["constant"]
            rows.append([(p - j_qp) / (abs(j_qp) + 1.0) for p in plans] + [(p - j_qp) / track for p in plans] + [failed])
    rows = np.array(rows)
    print("%s" % name)
    for col, label in enumerate(("cold", "warm 1", "warm 4", "warm 8")):
        print("   %-7s gap/(|J|+1): median %.5f p90 %.5f max %.5f    excess/tracking cost: median %.3f p90 %.3f max %.3f"
              % (label, np.median(rows[:, col]), np.percentile(rows[:, col], 90), rows[:, col].max(),
                 np.median(rows[:, 4 + col]), np.percentile(rows[:, 4 + col], 90), rows[:, 4 + col].max()))
    print("   infeasible solves: %d" % rows[:, 8].sum(), flush=True)
