#!/usr/bin/env python3
"""Development aid (GPU box): what the previous batch's finalize costs the rollout kernel that carries it.  One process, two
handles on the headline's batch (4 096 x 4 096 x horizon 50): one issues a stream of batches (rollout_chained_kernel from the
second call on), the other the plain rollout kernel; launches alternate, every launch carries an event pair.
usage: [ACMPC_HIP_LIBRARY=/tmp/ab_<name>/libacmpc_hip.so] python3 tools/chained_ab.py [launches per kind]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ac-mpc_amd")]
import bench  # noqa: E402
from acmpc_amd import Engine, _capi, workloads  # noqa: E402


def main():
    launches = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    P, N, H = 4096, 4096, 50
    n = H - 1
    dev = torch.device("cuda", 0)
    batch = workloads.problem_batch("monza", 256, H, seed=0)
    batch.tables = np.tile(batch.tables, (16, 1, 1))[:P]
    batch.x0 = np.tile(batch.x0, (16, 1))[:P]
    s = torch.cuda.current_stream().cuda_stream
    engines = []
    for _ in range(2):
        eng = Engine(**workloads.engine_kwargs(batch, 0, N, device=0))
        eng.set_paths(batch.tables)
        eng.sync_tables(s)
        engines.append(eng)
    chained, plain = engines
    x0 = torch.tensor(batch.x0, device=dev)
    u_ref = torch.tensor(np.stack([batch.tables[:, 6, :], batch.tables[:, 3, :]], axis=2), dtype=torch.float32, device=dev).contiguous()
    Us = []
    for b in range(2):
        U = torch.empty(P, n, 2, N, device=dev)
        chained.sample_device(u_ref.data_ptr(), 2 * n, u_ref.data_ptr(), P, N, n, 1, 0, bench.SAMPLE_SIGMA, 1000 + b, 0, U.data_ptr(), s)
        Us.append(U)
    rec = torch.empty(P, _capi.record_floats(n), device=dev)
    keys = torch.empty(P, dtype=torch.int64, device=dev)

    def one(i):
        U = Us[i % 2]
        chained.solve_stream_device(x0.data_ptr(), U.data_ptr(), u_ref.data_ptr(), 2 * n, u_ref.data_ptr(), P, N, n, 1, bench.SAMPLE_SIGMA,
                                    1000 + i % 2, 0, 0, keys.data_ptr(), rec.data_ptr(), s)
        plain.rollout_device(x0.data_ptr(), U.data_ptr(), P, N, n, 1, 0, 0, 0, s)

    for i in range(10):
        one(i)
    torch.cuda.synchronize()
    chained.profile_enable(launches)
    plain.profile_enable(launches)
    for i in range(launches):
        one(i)
    chained.solve_stream_flush(s)
    torch.cuda.synchronize()
    a, b = chained.profile_collect() * 1e3, plain.profile_collect() * 1e3
    print("%s: chained median %.1f us  plain median %.1f us  difference of medians %.1f us, median of differences %.1f us"
          % (os.environ.get("ACMPC_HIP_LIBRARY", "tree"), np.median(a), np.median(b), np.median(a) - np.median(b), np.median(a - b)))


if __name__ == "__main__":
    main()
