"""Development probe (GPU box): how the closed-loop get_control divides between Python and the library call."""
import copy
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ac-mpc_amd"))
from acmpc_amd import _capi, workloads  # noqa: E402
from acmpc_amd.mpc import build_mpc  # noqa: E402

track = workloads.synthetic_track("silverstone")
paths = [workloads.reference_path_from_centreline(workloads.local_centreline(track, (i * 2) % len(track["centre"])), 50)
         for i in range(2020)]
cfg = copy.deepcopy(workloads.RACING_CONTROL["silverstone"])
cfg["speed_profile_constraints"]["v_max"] = float(cfg["unlocalised_max_speed"])
mpc = build_mpc(cfg, workloads.PlaceholderVehicle())
inner = []
original = _capi._TickBuffers.call


def timed(self, tick, coords, centre):
    t0 = time.perf_counter()
    rc = original(self, tick, coords, centre)
    inner.append(time.perf_counter() - t0)
    return rc


for p in paths[:20]:
    mpc.get_control(p)
t = []
for p in paths[20:1020]:
    t0 = time.perf_counter()
    mpc.get_control(p)
    t.append(time.perf_counter() - t0)
_capi._TickBuffers.call = timed
w = []
for p in paths[1020:]:
    t0 = time.perf_counter()
    mpc.get_control(p)
    w.append(time.perf_counter() - t0)
print("get_control p50 %.1f us (untouched); with the timer in place p50 %.1f us, of which the library call p50 %.1f us"
      % (np.percentile(t, 50) * 1e6, np.percentile(w, 50) * 1e6, np.percentile(inner, 50) * 1e6))
