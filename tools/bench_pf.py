"""Timing of the particle-scoring seam (development tool): GPU `ParticleScorer.update_particles` vs the oracle's NumPy
restatement of the reference (KD-tree replaced by brute force there, so the CPU figure is indicative only)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ac-mpc_amd"))
from acmpc_amd import workloads  # noqa: E402
from acmpc_amd.particle_filter import ParticleScorer  # noqa: E402

track = workloads.synthetic_track("monza")
cfg = dict(n_particles=100000, score_distribution=dict(mean=0, sigma=10),
           thresholds=dict(offset=10, rotation=90, minimum_particles=20, track_limit=20.0))
scorer = ParticleScorer(cfg, track)
rng = np.random.default_rng(0)
centre = track["centre"]
left = np.stack([-4.7 + rng.normal(0, 0.1, 100), np.linspace(0, 49, 100)], axis=1).astype(np.float32)
right = np.stack([4.7 + rng.normal(0, 0.1, 100), np.linspace(0, 49, 100)], axis=1).astype(np.float32)
SIZES = [int(a) for a in sys.argv[1:]] or [500, 5000, 100000]   # (tools/profile_pf.sh profiles one size per run)
for P in SIZES:
    seeds = rng.integers(0, len(centre), P)
    states = np.concatenate([centre[seeds] + rng.normal(0, 2.0, (P, 2)), rng.uniform(-3, 3, (P, 1))], axis=1).astype(np.float32)
    scorer.update_particles(states, [left, right])
    t = []
    for _ in range(10):
        t0 = time.perf_counter()
        scorer.update_particles(states, [left, right])
        t.append(time.perf_counter() - t0)
    print("P=%6d  map %d pts x3, %d obs pts: %.3f ms per scoring call (%.3g particles/s)"
          % (P, len(centre), 200, np.median(t) * 1e3, P / np.median(t)), flush=True)
