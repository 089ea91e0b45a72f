"""Development probe: where the host time of SpatialMPC.get_control goes (perf_counter around its stages)."""
import copy, math, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ac-mpc_amd"))
from acmpc_amd import workloads
from acmpc_amd.mpc import build_mpc

cfg = copy.deepcopy(workloads.RACING_CONTROL["silverstone"]); cfg["speed_profile_constraints"]["v_max"] = 32.0
mpc = build_mpc(cfg, workloads.PlaceholderVehicle())
track = workloads.synthetic_track("silverstone")
paths = [workloads.reference_path_from_centreline(workloads.local_centreline(track, (i * 2) % 11000), 50) for i in range(1020)]
for p in paths[:20]: mpc.get_control(p)
T = np.zeros(6); n = 49
for coords in paths[20:]:
    t0 = time.perf_counter()
    path = mpc.construct_waypoints(coords); t1 = time.perf_counter()
    path = mpc.compute_speed_profile(path, False, end_vel=mpc.speed_profile_constraints["end_velocity"]); t2 = time.perf_counter()
    state = mpc.model.t2s(path.get_state(0), np.array([0.0, 0.0, math.pi / 2])); t3 = time.perf_counter()
    dec = mpc._control_solver.solve(state, path); t4 = time.perf_counter()
    from acmpc_amd import _capi
    out = _capi.unpack_decision(dec.x, n, path.table, mpc.model.length)
    t5 = time.perf_counter()
    T += [t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t5 - t0]
T = T / 1000 * 1e6
print("waypoints %.1f us | speed profile %.1f | t2s %.1f | solve %.1f | unpack %.1f | total %.1f" % tuple(T))
