"""Helper launched by tests/test_gpu_baseline_forms.py under torch.distributed.run (not collected by pytest):
BASELINE config 4 (Nordschleife, 262 144 candidates x horizon 80) split over the ranks of the process group, counter-
based candidates, ONE all-reduce(MIN) of the keys (gloo carries it; all ranks share cuda:0), every rank re-draws the
winner.  Every rank's record must equal the unsharded solve over the same global indices."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [os.path.join(ROOT, "ac-mpc_amd"), os.path.join(ROOT, "oracle"), HERE]
import acmpc_oracle as orc  # noqa: E402
from acmpc_amd import Engine, _capi  # noqa: E402
from acmpc_amd.sharding import ShardedRollout, shard_range  # noqa: E402
from test_support import engine_kwargs, make_problem  # noqa: E402


def main():
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    H, N = 80, 262144
    n = H - 1
    offset, local = shard_range(N, rank, world)
    prob = make_problem(orc, "nordschleife", H, 16, seed=4243)
    stream = torch.cuda.current_stream().cuda_stream
    x0 = torch.tensor(prob["x0"][None], device=dev)
    u_ref = torch.tensor(np.stack([prob["table"][orc.ROW_V], prob["table"][orc.ROW_KAPPA]], axis=1)[None],
                         dtype=torch.float32, device=dev).contiguous()
    sigma, seed = (2.0, 0.01), 99
    eng = Engine(**engine_kwargs(prob, 0, 1, local, n))
    eng.set_paths(prob["table"])
    shard = ShardedRollout(eng, 1, local, n, 1, index_offset=offset, device=dev, host_collectives=True)
    shard.use_sampler(u_ref, u_ref, sigma, seed, 0)
    U = torch.empty(1, n, 2, local, device=dev)
    shard.sample(U, stream)
    rec = shard.step(x0, U, stream)
    torch.cuda.synchronize()
    rec = rec.cpu().numpy().copy()
    rec[:, _capi.REC_NFEASIBLE] = 0
    # the unsharded solve of the same global indices, on this rank's own handle
    full = Engine(**engine_kwargs(prob, 0, 1, N, n))
    full.set_paths(prob["table"])
    U_full = torch.empty(1, n, 2, N, device=dev)
    full.sample_device(u_ref.data_ptr(), 2 * n, u_ref.data_ptr(), 1, N, n, 1, 0, sigma, seed, 0, U_full.data_ptr(), stream)
    want = torch.empty(1, _capi.record_floats(n), device=dev)
    keys = torch.empty(1, dtype=torch.int64, device=dev)
    full.solve_device(x0.data_ptr(), U_full.data_ptr(), 1, N, n, 1, 0, keys.data_ptr(), want.data_ptr(), stream)
    torch.cuda.synchronize()
    want = want.cpu().numpy().copy()
    want[:, _capi.REC_NFEASIBLE] = 0
    if not np.array_equal(rec, want):
        raise SystemExit("rank %d: the sharded record differs from the unsharded solve" % rank)
    if int(shard.keys[0]) != int(keys[0]):
        raise SystemExit("rank %d: the reduced key differs from the unsharded key" % rank)
    gathered = [None] * world
    dist.all_gather_object(gathered, rec.tobytes())
    if any(g != gathered[0] for g in gathered):
        raise SystemExit("ranks disagree")
    # softmin variant: the weighted mean over BOTH ranks' candidates, one all-gather; the same bits on every
    # rank, and the unsharded kernel's mean to float32 rounding
    mean, wsum = shard.softmin(U, stream)
    want_mean = torch.empty(1, n, 2, device=dev)
    want_wsum = torch.empty(1, dtype=torch.float64, device=dev)
    costs_full = torch.empty(1, N, device=dev)
    full.rollout_device(x0.data_ptr(), U_full.data_ptr(), 1, N, n, 1, 0, costs_full.data_ptr(), keys.data_ptr(), stream)
    full.softmin_device(costs_full.data_ptr(), keys.data_ptr(), U_full.data_ptr(), 1, N, n, 1, want_mean.data_ptr(),
                        want_wsum.data_ptr(), stream)
    torch.cuda.synchronize()
    if not torch.allclose(mean, want_mean, rtol=2e-6, atol=1e-7):
        raise SystemExit("rank %d: the sharded softmin mean differs from the unsharded one by %g"
                         % (rank, float((mean - want_mean).abs().max())))
    if abs(float(wsum[0]) - float(want_wsum[0])) > 1e-9 * float(want_wsum[0]):
        raise SystemExit("rank %d: softmin weight sums differ" % rank)
    dist.all_gather_object(gathered, mean.cpu().numpy().tobytes())
    if any(g != gathered[0] for g in gathered):
        raise SystemExit("ranks disagree on the softmin mean")
    dist.barrier()
    if rank == 0:
        print("config 4 in two ranks ok")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
