"""Mode T rollout kernel alone on the bench's secondary workload (256 poses x 4 096 candidates x H = 50, smooth sampled
controls): exhaustive semantics, window (2,5), window (1,2).  Run on the GPU box:  python3 tools/time_modeT.py [P]"""
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ac-mpc_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from acmpc_amd import Engine, workloads  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 256
device = torch.device("cuda", 0)
torch.cuda.set_device(0)
H, N, n = 50, 4096, 49
base = workloads.problem_batch("monza", 256, H, seed=0)
stream = torch.cuda.current_stream().cuda_stream
for name, window in (("exhaustive", None), ("window_2_5", (2, 5)), ("window_1_2", (1, 2))):
    batch = copy.copy(base)
    reps = P // 256
    batch.tables, batch.x0, batch.pose0 = (np.tile(base.tables, (reps, 1, 1)), np.tile(base.x0, (reps, 1)),
                                           np.tile(base.pose0, (reps, 1)))
    u_ref = torch.tensor(np.stack([batch.tables[:, 6, :], batch.tables[:, 3, :]], axis=2), dtype=torch.float32,
                         device=device).contiguous()
    costs = torch.empty(P, N, device=device)
    eng = Engine(**workloads.engine_kwargs(batch, 1, N, device=0, nn_window=window))
    eng.set_paths(batch.tables)
    eng.sync_tables(stream)
    x0 = torch.tensor(batch.pose0, device=device)
    U = torch.empty((P, n, 2, N), device=device)
    eng.sample_device(u_ref.data_ptr(), 2 * n, u_ref.data_ptr(), P, N, n, 1, 0, bench.SAMPLE_SIGMA, 77, 0, U.data_ptr(), stream)
    for _ in range(max(3, 4096 // P)):   # (a small launch is over before the clocks have come up from idle: warm up for ~2 ms)
        eng.rollout_device(x0.data_ptr(), U.data_ptr(), P, N, n, 1, 0, costs.data_ptr(), 0, stream)
    eng.set_option("ACMPC_START_CLOCKS", "1")
    eng.profile_enable(20)
    late = []
    for _ in range(20):
        eng.rollout_device(x0.data_ptr(), U.data_ptr(), P, N, n, 1, 0, costs.data_ptr(), 0, stream)
        starts = eng.rollout_start_clocks()     # (synchronises: the launches of this loop do not queue up behind each other)
        late.append(int((starts > 10.0).sum()))
    torch.cuda.synchronize()
    times = eng.profile_collect()
    ms = float(np.median(times))
    byts = P * N * (8 * n + 4)
    print("%-12s %8.1f us  %.3f of the HBM roofline  (min %.1f max %.1f us; workgroups started > 10 us late per launch: median %d, "
          "max %d of %d; checksum %.6e)" % (name, ms * 1e3, byts / (ms * 1e-3) / 1e9 / 8000.0, times.min() * 1e3, times.max() * 1e3,
                                           int(np.median(late)), max(late), len(starts), float(costs.double().sum())), flush=True)
    eng.close()
