"""Latency of one device-resident solve per call against the number of candidates (horizon 50): where the one-launch
form's time goes up.  Run on the GPU box:  python3 tools/time_solo_sizes.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ac-mpc_amd"))

import torch  # noqa: E402

import bench  # noqa: E402
from acmpc_amd import Engine, workloads  # noqa: E402

device = torch.device("cuda", 0)
torch.cuda.set_device(0)
for layout in (1, 0):
    for N in (4096, 16384, 32768, 49152, 65536, 98304, 131072):
        p50 = sorted(bench.single_solve(workloads, Engine, "spa", 50, N, 0, layout, device, iters=300,
                                        host_pointer=False)["device_resident_us_p50"] for _ in range(3))
        print("layout %d N %6d (%4d workgroups of 64)  %s us" % (layout, N, N // 64, "/".join("%.1f" % v for v in p50)), flush=True)
