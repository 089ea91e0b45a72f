"""Development probe: wall time of one acmpc_optimize call (the closed-loop solve's GPU part) on the GPU box."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ac-mpc_amd"))
from acmpc_amd import Engine, workloads  # noqa: E402

H, N, rounds = 50, 4096, 4
n = H - 1
mode = 1 if "--mode-t" in sys.argv else 0
batch = workloads.problem_batch("silverstone", 1, H, seed=0)
window = (2, 5) if mode == 1 else None
engine = Engine(**workloads.engine_kwargs(batch, mode, N, nn_window=window))
engine.set_paths(batch.tables)
u_ref = np.stack([batch.tables[:, 6, :], batch.tables[:, 3, :]], axis=2).astype(np.float32)
x0 = (batch.x0 if mode == 0 else batch.pose0).astype(np.float32)
for r in (1, rounds):
    for _ in range(20):
        engine.optimize(x0, u_ref, u_ref, N, r, (3.0, 0.01), shrink=0.5, seed=1)
    t = np.empty(500)
    for i in range(500):
        t0 = time.perf_counter()
        out = engine.optimize(x0, u_ref, u_ref, N, r, (3.0, 0.01), shrink=0.5, seed=i)
        t[i] = time.perf_counter() - t0
    print("optimize mode %s rounds=%d N=%d: p50 %.1f us  p99 %.1f us  cost %.6g" % (
        "ST"[mode], r, N, np.percentile(t, 50) * 1e6, np.percentile(t, 99) * 1e6, out["cost"][0]))
