"""Development probe: wall time of SpatialMPC.get_control and of its stages on the GPU box."""
import copy
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ac-mpc_amd"))
from acmpc_amd import workloads  # noqa: E402
from acmpc_amd.mpc import build_mpc  # noqa: E402

cfg = copy.deepcopy(workloads.RACING_CONTROL["monza"])
cfg["speed_profile_constraints"]["v_max"] = 28.0
mpc = build_mpc(cfg, workloads.PlaceholderVehicle())
track = workloads.synthetic_track("monza")
paths = [workloads.reference_path_from_centreline(workloads.local_centreline(track, i * 40), 50) for i in range(60)]
for p in paths[:5]:
    mpc.get_control(p)
t = []
for p in paths:
    t0 = time.perf_counter()
    mpc.get_control(p)
    t.append(time.perf_counter() - t0)
t = np.array(t) * 1e3
print("get_control: p50 %.2f ms  p99 %.2f ms  (infeasible %d)" % (np.percentile(t, 50), np.percentile(t, 99), mpc.infeasibility_counter))
# stages
path = mpc.construct_waypoints(paths[7])
t0 = time.perf_counter()
for _ in range(20):
    mpc.compute_speed_profile(path, False, end_vel=14.0)
print("speed profile: %.2f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
solver = mpc._control_solver
x0 = mpc.model.t2s(path.get_state(0), np.array([0.0, 0.0, np.pi / 2]))
t0 = time.perf_counter()
for _ in range(20):
    solver.solve(x0, path)
print("control solve (%d rounds x %d candidates): %.2f ms" % (solver._rounds, solver._n_candidates, (time.perf_counter() - t0) / 20 * 1e3))

# finer breakdown
import cProfile
import pstats
pr = cProfile.Profile()
pr.enable()
for p in paths:
    mpc.get_control(p)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
