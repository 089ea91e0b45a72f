"""Helper launched by tests/test_gpu_sharded.py under torch.distributed.run (not collected by pytest): every rank
draws and rolls out its slice of the candidates of each round (ShardedOptimizer, gloo collectives, all ranks on
cuda:0), and the records every rank ends with must equal the unsharded acmpc_optimize over the same global indices."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [os.path.join(ROOT, "ac-mpc_amd"), os.path.join(ROOT, "oracle"), HERE]
import acmpc_oracle as orc  # noqa: E402
from acmpc_amd import Engine  # noqa: E402
from acmpc_amd.sharding import ShardedOptimizer  # noqa: E402
from test_support import engine_kwargs, make_problem  # noqa: E402


def main():
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    for mode, window in ((0, None), (1, (2, 5))):
        P, H, N, rounds = 3, 50, 4096, 3
        n, local = H - 1, N // world
        problems = [make_problem(orc, "silverstone", H, 4, seed=700 + p) for p in range(P)]
        tables = np.stack([p["table"] for p in problems])
        u_ref = np.stack([np.stack([p["table"][orc.ROW_V], p["table"][orc.ROW_KAPPA]], axis=1)
                          for p in problems]).astype(np.float32)
        x0 = np.stack([p["x0"] if mode == 0 else p["pose0"] for p in problems])
        eng = Engine(**engine_kwargs(problems[0], mode, P, local, n, nn_window=window))
        eng.set_paths(tables)
        opt = ShardedOptimizer(eng, P, local, n, index_offset=rank * local, device=dev, host_collectives=True)
        d_x0, d_ref = torch.tensor(x0, device=dev), torch.tensor(u_ref, device=dev)
        sigma, seed = (0.5, 0.001), 31
        rec = opt.solve(d_x0, d_ref, d_ref, rounds, sigma, shrink=0.5, seed=seed,
                        stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        rec = rec.cpu().numpy().copy()
        rec[:, 2] = 0                                            # n_feasible is each rank's own count
        full = Engine(**engine_kwargs(problems[0], mode, P, N, n, nn_window=window))
        full.set_paths(tables)
        want = full.optimize(x0, u_ref, u_ref, N, rounds, sigma, shrink=0.5, seed=seed)["records"].copy()
        want[:, 2] = 0
        if not np.array_equal(rec, want):
            raise SystemExit("rank %d mode %d: sharded records differ from the unsharded solve" % (rank, mode))
        gathered = [None] * world
        dist.all_gather_object(gathered, rec.tobytes())
        if any(g != gathered[0] for g in gathered):
            raise SystemExit("ranks disagree")
    dist.barrier()
    if rank == 0:
        print("sharded optimizer ok")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
