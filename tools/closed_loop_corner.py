"""Development tool (GPU box): the drop-in controller in closed loop round a circuit whose corners bind the QP's box rows.

A stadium - two straights joined by half circles of radius R - under the racing corridor (widths linspace(10, 6, H),
controller.py:256-267).  Every 10 ms: the next 150 m of centre line moved into the vehicle frame (perception's job in the
reference), `get_control`, the first command of the plan applied to the kinematic bicycle of localiser.py:66-95 for 10 ms.
Prints per-lap statistics: rejected solves, lateral error, how often and how long the box-constrained refinement ran, solve time.
usage: python3 tools/closed_loop_corner.py [radius 9.0] [laps 2] [track monza] [lq_candidate 2]"""
import copy
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sub in ("ac-mpc_amd", "tests"):
    sys.path.insert(0, os.path.join(ROOT, sub))

import numpy as np  # noqa: E402

from acmpc_amd import workloads  # noqa: E402
from acmpc_amd.mpc import build_mpc  # noqa: E402
from test_support import RACING, PlaceholderVehicle, WHEELBASE  # noqa: E402


def stadium(radius, straight=60.0, spacing=0.5):
    pts = []
    s = np.arange(0.0, straight, spacing)
    pts.append(np.stack([np.zeros_like(s), s], axis=1))
    a = np.arange(0.0, np.pi, spacing / radius)
    pts.append(np.stack([-radius + radius * np.cos(a), straight + radius * np.sin(a)], axis=1))
    pts.append(np.stack([np.full_like(s, -2 * radius), straight - s], axis=1))
    pts.append(np.stack([-radius - radius * np.cos(a), -radius * np.sin(a)], axis=1))
    return np.concatenate(pts)


def run(radius=9.0, laps=2, track="monza", lq_candidate=2, ticks_per_lap=None, verbose=True):
    cfg = copy.deepcopy(RACING[track])
    cfg["lq_candidate"] = lq_candidate
    H = cfg["horizon"]
    centre = stadium(radius)
    M = len(centre)
    mpc = build_mpc(cfg, PlaceholderVehicle())
    pose = np.array([0.3, 0.0, np.pi / 2])          # 0.3 m off the centre line on the first straight
    dt = 0.01
    count = int(round(150.0 / 0.5)) + 1
    out = []
    lap_len = M * 0.5
    travelled, tick, nearest = 0.0, 0, 0
    while travelled < laps * lap_len and tick < 20000:
        search = (nearest + np.arange(-10, 60)) % M
        nearest = int(search[np.argmin(((centre[search] - pose[:2]) ** 2).sum(1))])
        window = centre[(nearest + np.arange(count)) % M]
        rot = np.pi / 2 - pose[2]
        c, s = np.cos(rot), np.sin(rot)
        local = (window - pose[:2]) @ np.array([[c, s], [-s, c]])
        t = np.linspace(0, count - 1, 500)
        line = np.stack([np.interp(t, np.arange(count), local[:, 0]), np.interp(t, np.arange(count), local[:, 1])], axis=1).astype(np.float32)
        path = workloads.reference_path_from_centreline(line, H)
        mpc.speed_profile_constraints["v_max"] = float(cfg["unlocalised_max_speed"])
        t0 = time.perf_counter()
        mpc.get_control(path, False, 0.0)
        solve_us = (time.perf_counter() - t0) * 1e6
        stats = mpc._control_solver._engine.lq_box_stats() if lq_candidate == 2 else dict(iterations=0, chosen=0, triggered=False)
        v, delta = mpc.projected_control[0][0], mpc.projected_control[1][0]
        lateral = float(np.sqrt(((centre[nearest] - pose[:2]) ** 2).sum()))
        out.append((solve_us, mpc.infeasibility_counter > 0, lateral, stats["iterations"], stats["chosen"], stats["triggered"], v, delta))
        pose = pose + dt * np.array([v * np.cos(pose[2]), v * np.sin(pose[2]), v * np.tan(delta) / WHEELBASE])
        travelled += v * dt
        tick += 1
    a = np.array(out, dtype=np.float64)
    report = {"radius": radius, "track": track, "lq_candidate": lq_candidate, "ticks": int(len(a)), "rejected_solves": int(a[:, 1].sum()),
              "lateral_max_m": float(a[:, 2].max()), "lateral_p50_m": float(np.median(a[:, 2])),
              "refinement_triggered_share": float(a[:, 5].mean()), "iterations_when_triggered_p50": float(np.median(a[a[:, 5] > 0, 3])) if a[:, 5].any() else 0.0,
              "iterations_max": float(a[:, 3].max()), "refined_plan_taken_share": float((a[:, 4] > 0).mean()),
              "solve_us_p50": float(np.median(a[:, 0])), "solve_us_p50_when_triggered": float(np.median(a[a[:, 5] > 0, 0])) if a[:, 5].any() else 0.0,
              "solve_us_p99": float(np.percentile(a[:, 0], 99)), "speed_p50": float(np.median(a[:, 6])), "steer_at_limit_share": float((np.abs(a[:, 7]) > 0.299).mean())}
    if verbose:
        print(report, flush=True)
    return report


if __name__ == "__main__":
    radius = float(sys.argv[1]) if len(sys.argv) > 1 else 9.0
    laps = float(sys.argv[2]) if len(sys.argv) > 2 else 2
    track = sys.argv[3] if len(sys.argv) > 3 else "monza"
    lq = int(sys.argv[4]) if len(sys.argv) > 4 else 2
    run(radius, laps, track, lq)
