"""Development driver (GPU box): a few fused sample + rollout launches at the scale of bench.py's `sampled_fused_16M`
(1 024 problems x 16 384 candidates x horizon 50), for the profiler.
    rocprofv3 --pmc ... --kernel-trace -- python3 tools/run_sampled_fused.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ac-mpc_amd"))

import torch  # noqa: E402

import bench  # noqa: E402
from acmpc_amd import Engine, workloads  # noqa: E402

torch.cuda.set_device(0)
out = bench.sampled_fused_at_scale(workloads, Engine, "monza", 50, torch.device("cuda", 0), iters=int(sys.argv[1]) if len(sys.argv) > 1 else 4)
print({k: v for k, v in out.items() if k != "workload"})
