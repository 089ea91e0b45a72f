"""Development probe (GPU box): where a host-pointer acmpc_solve of 4 096 x 49 candidates spends its time - pageable vs
page-locked control matrix, and the bare copies through torch for comparison."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ac-mpc_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from acmpc_amd import Engine, _capi, workloads  # noqa: E402

N, H = 4096, 50
n = H - 1
batch = workloads.problem_batch("silverstone", 1, H, seed=1)
eng = Engine(**workloads.engine_kwargs(batch, 0, N, device=0))
eng.set_paths(batch.tables)
rng = np.random.default_rng(0)
U_page = (np.stack([batch.tables[0, 6], batch.tables[0, 3]], axis=1)[None, None] + rng.standard_normal((1, N, n, 2)) * [2.0, 0.01]).astype(np.float32)
U_pin = _capi.pinned_empty(U_page.shape, np.float32)
U_pin[...] = U_page
x0 = batch.x0.astype(np.float32)


def timed(fn, iters=300):
    for _ in range(20):
        fn()
    t = []
    for _ in range(iters):
        t0 = time.perf_counter()
        fn()
        t.append(time.perf_counter() - t0)
    return np.percentile(np.array(t) * 1e6, [50, 99])


print("acmpc_solve pageable U   p50 %.1f us p99 %.1f" % tuple(timed(lambda: eng.solve(x0, U_page, layout=0, want_costs=False))))
print("acmpc_solve pinned U     p50 %.1f us p99 %.1f" % tuple(timed(lambda: eng.solve(x0, U_pin, layout=0, want_costs=False))))


def with_new_path(U):   # what a controller does per pose: new tables, then the solve (the upload rides in the solve)
    eng.set_paths(batch.tables)
    eng.solve(x0, U, layout=0, want_costs=False)


print("set_paths + solve, pageable U   p50 %.1f us p99 %.1f" % tuple(timed(lambda: with_new_path(U_page))))
print("set_paths + solve, pinned U     p50 %.1f us p99 %.1f" % tuple(timed(lambda: with_new_path(U_pin))))
dev = torch.empty(U_page.shape, device="cuda")
t_page, t_pin = torch.from_numpy(U_page), torch.from_numpy(U_pin)


def copy(src):
    dev.copy_(src, non_blocking=True)
    torch.cuda.synchronize()


print("torch H2D 1.6 MB pageable p50 %.1f us p99 %.1f" % tuple(timed(lambda: copy(t_page))))
print("torch H2D 1.6 MB pinned   p50 %.1f us p99 %.1f  (is_pinned says %s)" % (tuple(timed(lambda: copy(t_pin))) + (t_pin.is_pinned(),)))
small = torch.empty(252, device="cuda")
host_small = torch.empty(252)
print("torch D2H 1 kB pageable   p50 %.1f us p99 %.1f" % tuple(timed(lambda: host_small.copy_(small))))
eng.close()
