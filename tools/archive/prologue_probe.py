"""Development probe (GPU box): where the per-tick prologue kernel spends its time.  Needs an A/B build with phase stamps:
    lib=$(tools/ab_build.sh stamps "-DACMPC_STAMPS") && ACMPC_HIP_LIBRARY=$lib python3 tools/prologue_probe.py"""
import copy
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "ac-mpc_amd"))
from acmpc_amd import _capi, workloads  # noqa: E402
from acmpc_amd.mpc import build_mpc  # noqa: E402

lib = _capi.load_library()
read = lib.acmpc_debug_prologue_stamps
read.argtypes = [C.c_void_p]
H = int(os.environ.get("TICK_H", "50"))
track = workloads.synthetic_track("silverstone")
cfg = copy.deepcopy(workloads.RACING_CONTROL["silverstone"])
cfg["horizon"] = H
cfg["speed_profile_constraints"]["v_max"] = float(cfg["unlocalised_max_speed"])
mpc = build_mpc(cfg, workloads.PlaceholderVehicle())
use_map = os.environ.get("TICK_MAP", "0") == "1"
if use_map:
    mpc.bind_map(track)
paths = [workloads.reference_path_from_centreline(workloads.local_centreline(track, (i * 2) % len(track["centre"])), H) for i in range(220)]
rows = []
stamps = np.zeros(16, dtype=np.uint64)
for i, path in enumerate(paths):
    if use_map:
        mpc.get_control_at(map_index=(i * 2) % len(track["centre"]))
    else:
        mpc.get_control(path)
    if i >= 20:
        assert read(stamps.ctypes.data) == 0
        rows.append(stamps[:8].astype(np.int64).copy())
rows = np.array(rows)
names = ["header read", "previous plan / warm state requested, map window", "construct_waypoints", "velocity ceiling, warm iterate into LDS",
         "speed-profile QP (ADMM)", "iterate kept, velocities", "t2s, linearise, reference controls, centre", "7 x n table to the host"]
delta = np.diff(rows, axis=1) * 0.01   # 100 MHz wall clock -> us
print("prologue phases, median us over %d ticks (H = %d, path from %s):" % (len(rows), H, "the map" if use_map else "the caller"))
for k in range(7):
    print("  %-62s %6.2f" % (names[k + 1], np.median(delta[:, k])))
print("  %-62s %6.2f" % ("total, first to last stamp", np.median(rows[:, 7] - rows[:, 0]) * 0.01))
