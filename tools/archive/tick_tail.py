"""Development probe (GPU box): where the tail of the closed-loop latency distribution comes from - percentiles of
get_control with the garbage collector on and off, and the positions of the slowest calls."""
import copy
import gc
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ac-mpc_amd"))
from acmpc_amd import workloads  # noqa: E402
from acmpc_amd.mpc import build_mpc  # noqa: E402

track = workloads.synthetic_track("silverstone")
paths = [workloads.reference_path_from_centreline(workloads.local_centreline(track, (i * 2) % len(track["centre"])), 50)
         for i in range(3020)]
cfg = copy.deepcopy(workloads.RACING_CONTROL["silverstone"])
cfg["speed_profile_constraints"]["v_max"] = float(cfg["unlocalised_max_speed"])
mpc = build_mpc(cfg, workloads.PlaceholderVehicle())
for p in paths[:20]:
    mpc.get_control(p)
for label, collect in (("gc on", True), ("gc off", False)):
    gc.enable() if collect else gc.disable()
    t = []
    for p in paths[20:]:
        t0 = time.perf_counter()
        mpc.get_control(p)
        t.append(time.perf_counter() - t0)
    t = np.array(t) * 1e6
    slow = np.nonzero(t > np.percentile(t, 98))[0]
    print("%-7s p50 %.1f  p90 %.1f  p95 %.1f  p99 %.1f  p99.9 %.1f  max %.1f us; gaps between the slowest 2%%: %s"
          % (label, *(np.percentile(t, q) for q in (50, 90, 95, 99, 99.9)), t.max(), np.diff(slow)[:24]), flush=True)
gc.enable()
