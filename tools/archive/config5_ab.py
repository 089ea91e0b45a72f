#!/usr/bin/env python3
"""Development aid (GPU box): bench.py's config5_host_pointer (consecutive poses, one host-pointer solve each, the CPU
building the next control matrix in between) with the transfers in place (default) and as copies (ACMPC_NO_ZERO_COPY=1),
pageable and page-locked control matrix, interleaved.   usage: python3 tools/config5_ab.py [poses]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ac-mpc_amd")]
import bench  # noqa: E402
from acmpc_amd import Engine, workloads  # noqa: E402

poses = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
device = torch.device("cuda", 0)
for rep in range(2):
    for copies in ("0", "1"):
        os.environ["ACMPC_NO_ZERO_COPY"] = copies     # (read when the handle is created)
        for pinned in (False, True):
            out = bench.config5_host_pointer(workloads, Engine, device, poses=poses, pinned=pinned)
            print("%-10s %-12s p50 %.1f us  p99 %.1f us" % ("copies" if copies == "1" else "in place", "page-locked" if pinned else "pageable",
                                                          out["solve_us_p50"], out["solve_us_p99"]), flush=True)
