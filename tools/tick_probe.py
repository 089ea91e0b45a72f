"""Development probe (GPU box): closed-loop get_control latency on consecutive poses (TICK_H = horizon, default 50;
TICK_MODE = S | T, the controller's rollout_mode; TICK_WINDOW = "back,ahead" or "none" = mode T's nn_window; unset: the controller's default, the nearest of all)."""
import copy, os, sys, time
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "ac-mpc_amd"))
from acmpc_amd import workloads
from acmpc_amd.mpc import build_mpc
H = int(os.environ.get("TICK_H", "50"))
track = workloads.synthetic_track("silverstone")
paths = [workloads.reference_path_from_centreline(workloads.local_centreline(track, (i * 2) % len(track["centre"])), H) for i in range(1520)]
cfg = copy.deepcopy(workloads.RACING_CONTROL["silverstone"])
cfg["horizon"] = H
cfg["rollout_mode"] = os.environ.get("TICK_MODE", "S")
if "TICK_WINDOW" in os.environ:
    cfg["nn_window"] = None if os.environ["TICK_WINDOW"] == "none" else tuple(int(v) for v in os.environ["TICK_WINDOW"].split(","))
if "TICK_CHECK" in os.environ:
    cfg["speed_profile_check_every"] = int(os.environ["TICK_CHECK"])
cfg["speed_profile_constraints"]["v_max"] = float(cfg["unlocalised_max_speed"])
mpc = build_mpc(cfg, workloads.PlaceholderVehicle())
for p in paths[:20]: mpc.get_control(p)
t = []
for p in paths[20:]:
    t0 = time.perf_counter(); mpc.get_control(p); t.append(time.perf_counter() - t0)
t = np.array(t) * 1e6
print("tick p50 %.1f us  p99 %.1f us infeasible %d" % (np.percentile(t, 50), np.percentile(t, 99), mpc.infeasibility_counter), flush=True)
