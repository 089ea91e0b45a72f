#!/usr/bin/env python3
"""Development aid (GPU box): the batched finalize of the headline's batch (4 096 problems) on its own - back to back (warm
caches and address translations) and behind a 6.6 GB stream through HBM (as it runs in the headline step).
usage: python3 tools/finalize_probe.py   (ACMPC_NO_GROUP_FINALIZE=1 for a wavefront per problem)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "ac-mpc_amd")]
from acmpc_amd import Engine, _capi, workloads  # noqa: E402


def main():
    P, N, H = 4096, 4096, 50
    n = H - 1
    dev = torch.device("cuda", 0)
    batch = workloads.problem_batch("monza", 256, H, seed=0)
    batch.tables = np.tile(batch.tables, (16, 1, 1))[:P]
    batch.x0 = np.tile(batch.x0, (16, 1))[:P]
    eng = Engine(**workloads.engine_kwargs(batch, 0, N, device=0))
    eng.set_paths(batch.tables)
    s = torch.cuda.current_stream().cuda_stream
    eng.sync_tables(s)
    x0 = torch.tensor(batch.x0, device=dev)
    u_ref = torch.tensor(np.stack([batch.tables[:, 6, :], batch.tables[:, 3, :]], axis=2), dtype=torch.float32, device=dev).contiguous()
    rng = np.random.default_rng(0)
    costs = rng.random(P).astype(np.float32)
    idx = rng.integers(0, N, P).astype(np.uint32)
    keys = torch.tensor(np.array([_capi.pack_key(float(c), int(i)) for c, i in zip(costs, idx)], dtype=np.int64), device=dev)
    rec = torch.empty(P, _capi.record_floats(n), device=dev)
    sigma = (2.0, 0.01)

    def fin():
        eng.finalize_sampled_device(keys.data_ptr(), x0.data_ptr(), u_ref.data_ptr(), 2 * n, u_ref.data_ptr(), P, N, n, sigma, 1000, 0,
                                    rec.data_ptr(), s)

    for _ in range(10):
        fin()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        fin()
    e1.record()
    torch.cuda.synchronize()
    print("back to back: %.2f us per launch (launch gaps included)" % (e0.elapsed_time(e1) * 1e3 / 200))
    big = torch.empty(int(3.3e9) // 4, device=dev)
    other = torch.empty_like(big)
    cold = []
    for _ in range(30):
        other.copy_(big)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fin()
        b.record()
        torch.cuda.synchronize()
        cold.append(a.elapsed_time(b) * 1e3)
    print("behind a 6.6 GB copy: median %.2f us (event pair round one launch)" % float(np.median(cold)))
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    b.record()
    torch.cuda.synchronize()
    print("empty event pair: %.2f us" % (a.elapsed_time(b) * 1e3))


if __name__ == "__main__":
    main()
