#!/usr/bin/env python3
"""Development probe (GPU box): one localisation update of the device-resident particle filter (step + update: scoring,
resampling, estimate - the particles never leave the GPU) at the reference's 500 particles, and the same through the
host-resampling filter.   usage: python3 tools/time_pf_filter.py [particles]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ac-mpc_amd"))
from acmpc_amd import workloads  # noqa: E402
from acmpc_amd.particle_filter import DeviceParticleFilter, ParticleFilter  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
CFG = dict(n_particles=n, n_converged_particles=n, sampling_noise=dict(x=1.1, y=1.1, yaw=3.0),
           control_noise=dict(velocity=0.25, yaw=2.0), thresholds=dict(offset=10, rotation=90, minimum_particles=20, track_limit=20.0),
           score_distribution=dict(mean=0, sigma=10), convergence_criteria=dict(maximum_distance=50, maximum_angle=90))
track = workloads.synthetic_track("monza")
centre, left, right = track["centre"], track["left"], track["right"]
M = len(centre)
rng = np.random.default_rng(1)
tangent = np.roll(centre, -1, axis=0) - np.roll(centre, 1, axis=0)
heading = np.arctan2(tangent[:, 1], tangent[:, 0])


def observe(limit, count, at, pose):
    pts = limit[(at + np.arange(count)) % M] - pose[:2]
    a = np.pi / 2 - pose[2]
    rot = np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]])
    return (pts @ rot.T + rng.normal(0, 0.15, (count, 2))).astype(np.float32)


for name, make in (("device-resident", lambda: DeviceParticleFilter(CFG, dict(centre=centre, left=left, right=right), seed=3)),
                   ("host resampling", lambda: ParticleFilter(CFG, dict(centre=centre, left=left, right=right), rng=np.random.default_rng(2)))):
    pf = make()
    start = np.array([centre[0, 0], centre[0, 1], heading[0]])
    states = (start + rng.normal(0, [1.0, 1.0, 0.05], (n, 3))).astype(np.float32)
    if hasattr(pf, "set_particles"):
        pf.set_particles(states, np.full(n, 1.0 / n, dtype=np.float32))
    else:
        pf.states = states
    t_step, t_update = [], []
    for tick in range(1200):
        at = (tick * 1) % M
        pose = np.array([centre[at, 0], centre[at, 1], heading[at]])
        obs = {"left": observe(left, 230, at, pose), "right": observe(right, 210, at, pose)}
        t0 = time.perf_counter()
        pf.step(0.0, 20.0, 0.05)
        t1 = time.perf_counter()
        pf.update(obs)
        t2 = time.perf_counter()
        if tick >= 200:
            t_step.append(t1 - t0)
            t_update.append(t2 - t1)
    print("%-16s %d particles: step p50 %.1f us, update p50 %.1f us p99 %.1f us" % (name, n, np.median(t_step) * 1e6, np.median(t_update) * 1e6,
                                                                              np.percentile(t_update, 99) * 1e6), flush=True)
