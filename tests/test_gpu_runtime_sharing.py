"""One HIP runtime per process, whichever of the library and PyTorch comes first.  PyTorch-ROCm bundles its own
libamdhip64 / libhsa-runtime64; loaded after the system's copies (which this library links by SONAME) they would be
a second HSA runtime that finds no GPU.  `_capi._share_torchs_hip_runtime` maps torch's copies first."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import os, sys
sys.path.insert(0, os.path.join(%(root)r, "ac-mpc_amd"))
import numpy as np
%(first)s
from acmpc_amd import Engine, workloads
batch = workloads.problem_batch("monza", 1, 20, seed=0)
eng = Engine(**workloads.engine_kwargs(batch, 0, 64))
eng.set_paths(batch.tables)
u_ref = np.stack([batch.tables[:, 6, :], batch.tables[:, 3, :]], axis=2).astype(np.float32)
out = eng.optimize(batch.x0.astype(np.float32), u_ref, u_ref, 64, 1, (0.5, 0.001), seed=1)
assert np.isfinite(out["cost"][0])
import torch
torch.cuda.init()
x = torch.ones(8, device="cuda") * 2
assert float(x.sum()) == 16.0
maps = open("/proc/self/maps").read()
runtimes = sorted({line.split()[-1] for line in maps.splitlines() if "libamdhip64" in line})
assert len(runtimes) == 1, runtimes
print("ok", runtimes[0])
'''


@pytest.mark.parametrize("first", ["pass  # the library first, torch afterwards", "import torch"])
def test_library_and_torch_share_one_hip_runtime(first):
    proc = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT, "first": first}], capture_output=True,
                          text=True, timeout=300)
    assert proc.returncode == 0, proc.stdout + proc.stderr
    assert proc.stdout.strip().startswith("ok")
