"""Development probe (GPU box): closed-loop get_control with the device prologue vs the host prologue, consecutive
poses 1 m apart.  For the prologue kernel's own time run it under the profiler with the interpreter named after `--`
(a script started through its shebang would go through `env`, which replaces the process after the profiler's preloaded
library has initialised the GPU - not allowed on this pool):
    rocprofv3 --kernel-trace --stats -- python3 tools/time_tick.py"""
import copy
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
         for i in range(1020)]
for device in (True, False):
    for check in ((10, 5, 3, 2, 1) if device else (10,)):
        cfg = copy.deepcopy(workloads.RACING_CONTROL["silverstone"])
        cfg["speed_profile_constraints"]["v_max"] = float(cfg["unlocalised_max_speed"])
        cfg.update(device_prologue=device, speed_profile_check_every=check)
        mpc = build_mpc(cfg, workloads.PlaceholderVehicle())
        for p in paths[:20]:
            mpc.get_control(p)
        t = []
        for p in paths[20:]:
            t0 = time.perf_counter()
            mpc.get_control(p)
            t.append(time.perf_counter() - t0)
        t = np.array(t) * 1e6
        print("device prologue %-5s check_every %2d: p50 %.1f us  p99 %.1f us  max %.1f us  infeasible %d"
              % (device, check, np.percentile(t, 50), np.percentile(t, 99), t.max(), mpc.infeasibility_counter), flush=True)
