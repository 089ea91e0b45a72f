"""Kernel-level sweep (development tool): times rollout_kernel alone with HIP events for a list of shapes.

    python tools/sweep.py "S,1,256,4096,50" "T,1,64,4096,50" ...      spec = mode,layout,P,N,H
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ac-mpc_amd"))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from acmpc_amd import Engine, workloads  # noqa: E402
from bench import make_controls  # noqa: E402


def run(spec, iters=30):
    parts = spec.split(",")
    window = (int(parts[5]), int(parts[6])) if len(parts) > 5 else None
    mode_s, layout, P, N, H = parts[:5]
    if mode_s not in ("S", "T"):
        raise SystemExit("spec %r: mode must be S or T" % spec)
    mode, layout, P, N, H = (0 if mode_s == "S" else 1), int(layout), int(P), int(N), int(H)
    n = H - 1
    device = torch.device("cuda", 0)
    base = workloads.problem_batch("monza", min(P, 16), H, seed=0)
    reps = (P + base.tables.shape[0] - 1) // base.tables.shape[0]
    base.tables = np.tile(base.tables, (reps, 1, 1))[:P]
    base.x0 = np.tile(base.x0, (reps, 1))[:P]
    base.pose0 = np.tile(base.pose0, (reps, 1))[:P]
    eng = Engine(**workloads.engine_kwargs(base, mode, N, device=0, nn_window=window))
    eng.set_paths(base.tables)
    stream = torch.cuda.current_stream().cuda_stream
    eng.sync_tables(stream)
    x0 = torch.tensor(base.x0 if mode == 0 else base.pose0, device=device)
    nbuf = 2 if P * N * n * 8 < 2e9 else 1
    Us = [make_controls(base, P, N, n, layout, device, seed=b) for b in range(nbuf)]
    costs = torch.empty(P, N, device=device)
    for i in range(3):
        eng.rollout_device(x0.data_ptr(), Us[i % nbuf].data_ptr(), P, N, n, layout, 0, costs.data_ptr(), 0, stream)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for i, (a, b) in enumerate(ev):
        a.record()
        eng.rollout_device(x0.data_ptr(), Us[i % nbuf].data_ptr(), P, N, n, layout, 0, costs.data_ptr(), 0, stream)
        b.record()
    torch.cuda.synchronize()
    ms = np.array([a.elapsed_time(b) for a, b in ev])
    byts = P * N * (8 * n + 4)
    print("%-22s kernel median %.1f us  min %.1f us  -> %.2f TB/s algorithmic (%.1f%% of 8 TB/s), %.3g traj/s"
          % (spec, np.median(ms) * 1e3, ms.min() * 1e3, byts / np.median(ms) / 1e9, byts / np.median(ms) / 1e9 / 8 * 100,
             P * N / np.median(ms) * 1e3), flush=True)
    eng.close()


if __name__ == "__main__":
    for spec in sys.argv[1:]:
        run(spec)
