"""rollout_kernel (mode S, step-major) alone on ONE problem, by launch shape (ACMPC_SHAPE = threads, candidates per lane):
what a lane's second candidate costs when the launch leaves SIMDs to spare.  GPU box: python3 tools/time_shapes_small.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ac-mpc_amd"))
import numpy as np
import torch
import bench
from acmpc_amd import Engine, workloads
device = torch.device("cuda", 0); torch.cuda.set_device(0)
H, n = 50, 49
batch = workloads.problem_batch("spa", 1, H, seed=0)
stream = torch.cuda.current_stream().cuda_stream
for N in (16384, 32768, 65536, 131072):
    row = []
    for shape in ("64,1", "256,1", "256,2", "256,4"):
        os.environ["ACMPC_SHAPE"] = shape
        eng = Engine(**workloads.engine_kwargs(batch, 0, N, device=0))
        eng.set_paths(batch.tables)
        eng.sync_tables(stream)
        u_ref = torch.tensor(np.stack([batch.tables[:, 6, :], batch.tables[:, 3, :]], axis=2), dtype=torch.float32, device=device).contiguous()
        U = torch.empty((1, n, 2, N), device=device)
        eng.sample_device(u_ref.data_ptr(), 2 * n, u_ref.data_ptr(), 1, N, n, 1, 0, bench.SAMPLE_SIGMA, 77, 0, U.data_ptr(), stream)
        x0 = torch.tensor(batch.x0, device=device)
        costs = torch.empty(1, N, device=device)
        for _ in range(5):
            eng.rollout_device(x0.data_ptr(), U.data_ptr(), 1, N, n, 1, 0, costs.data_ptr(), 0, stream)
        eng.profile_enable(100)
        for _ in range(100):
            eng.rollout_device(x0.data_ptr(), U.data_ptr(), 1, N, n, 1, 0, costs.data_ptr(), 0, stream)
        torch.cuda.synchronize()
        row.append("%s %.2f" % (shape, float(np.median(eng.profile_collect())) * 1e3))
        eng.close()
    print("N %6d  kernel us: %s" % (N, "   ".join(row)), flush=True)
