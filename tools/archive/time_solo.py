"""Latency of one device-resident solve per call (bench.py's single_solve blocks) in the forms ACMPC_SOLO_SPLIT /
ACMPC_NO_SOLO select.  Run on the GPU box:  python3 tools/time_solo.py"""
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
CASES = [("config2", "monza", 50, 4096), ("config3", "spa", 50, 65536), ("config4_share", "nordschleife", 80, 32768),
         ("N16384", "monza", 50, 16384)]
FORMS = [("default", {}), ("registers", {"ACMPC_SOLO_REGISTERS": "1"}), ("lds_trace", {"ACMPC_SOLO_REGISTERS": "0"}), ("one_wave", {"ACMPC_SOLO_SPLIT": "0"}),
         ("two_launches", {"ACMPC_NO_SOLO": "1"})]
KEYS = ("ACMPC_SOLO_SPLIT", "ACMPC_NO_SOLO", "ACMPC_SOLO_REGISTERS")
REPEATS = 3
for layout in (1, 0):
    for name, track, H, N in CASES:
        row = []
        for form, env in FORMS:
            p50 = []
            for _ in range(REPEATS):   # (a fresh engine each time: run-to-run spread of the p50 is ~1 us)
                for key in KEYS:
                    os.environ.pop(key, None)
                os.environ.update(env)
                out = bench.single_solve(workloads, Engine, track, H, N, 0, layout, device, iters=300, host_pointer=False)
                p50.append(out["device_resident_us_p50"])
            row.append("%s %s" % (form, "/".join("%.1f" % v for v in sorted(p50))))
        print("layout %d %-14s %s" % (layout, name, "  ".join(row)), flush=True)
