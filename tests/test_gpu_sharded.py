"""Sharded entry points on one GPU (needs an MI355X).  The RCCL collectives themselves need >1 GPU; what can be
proven on one card is everything around them: keys carry global indices, finalize writes the record only on the
owning shard, and the reduction of per-shard results equals the unsharded solve bit for bit."""
import os

import numpy as np
import pytest

import acmpc_oracle as orc
from test_support import engine_kwargs, make_problem

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("layout", [0, 1])
def test_two_shards_on_one_gpu_equal_the_full_solve(mode, layout):
    import torch
    from acmpc_amd import Engine, _capi
    P, H, total = 3, 50, 1536
    n = H - 1
    problems = [make_problem(orc, "silverstone", H, total, seed=90 + p) for p in range(P)]
    dev = torch.device("cuda", 0)
    x0 = torch.tensor(np.stack([p["x0"] if mode == 0 else p["pose0"] for p in problems]), device=dev)
    U_full = np.stack([p["U"] for p in problems])  # [P,N,n,2]
    stream = torch.cuda.current_stream().cuda_stream
    R = _capi.record_floats(n)

    full = Engine(**engine_kwargs(problems[0], mode, P, total, n))
    full.set_paths(np.stack([p["table"] for p in problems]))
    want = full.solve(x0.cpu().numpy(), U_full if layout == 0 else np.ascontiguousarray(U_full.transpose(0, 2, 3, 1)),
                      layout=layout)

    shards = [(0, 1000), (1000, 536)]  # uneven on purpose
    keys, recs, costs = [], [], []
    engines = []
    for offset, count in shards:
        eng = Engine(**engine_kwargs(problems[0], mode, P, count, n))
        eng.set_paths(np.stack([p["table"] for p in problems]))
        U = U_full[:, offset:offset + count]
        U = torch.tensor(U if layout == 0 else np.ascontiguousarray(U.transpose(0, 2, 3, 1)), device=dev)
        k = torch.empty(P, dtype=torch.int64, device=dev)
        c = torch.empty(P, count, dtype=torch.float32, device=dev)
        eng.rollout_device(x0.data_ptr(), U.data_ptr(), P, count, n, layout, offset, c.data_ptr(), k.data_ptr(), stream)
        engines.append((eng, U, offset, count))
        keys.append(k)
        costs.append(c)
    gkeys = torch.minimum(keys[0], keys[1])  # what all-reduce(MIN) computes
    for eng, U, offset, count in engines:
        r = torch.empty(P, R, dtype=torch.float32, device=dev)
        eng.finalize_device(gkeys.data_ptr(), x0.data_ptr(), U.data_ptr(), P, count, n, layout, offset, r.data_ptr(),
                            stream)
        recs.append(r)
    torch.cuda.synchronize()
    owners = torch.stack([r[:, _capi.REC_OWNER] for r in recs]).cpu().numpy()
    assert (owners.sum(axis=0) == 1).all(), "exactly one shard owns each winner"
    total_rec = (recs[0] + recs[1]).cpu().numpy()  # what all-reduce(SUM) computes
    np.testing.assert_array_equal(total_rec, want["records"])
    np.testing.assert_array_equal(np.concatenate([c.cpu().numpy() for c in costs], axis=1), want["costs"])
    for p in range(P):
        assert _capi.key_index(int(gkeys[p])) == want["best_idx"][p]
        assert _capi.key_cost(int(gkeys[p])) == want["cost"][p]


def test_sharded_rollout_class_with_a_one_rank_process_group():
    """ShardedRollout end to end over torch.distributed's nccl (= RCCL) backend with world_size 1: the collective
    calls execute on the GPU tensors, the result must equal Engine.solve."""
    import os
    import torch
    import torch.distributed as dist
    from acmpc_amd import Engine
    from acmpc_amd.sharding import ShardedRollout
    P, H, N = 4, 50, 2048
    n = H - 1
    problems = [make_problem(orc, "monza", H, N, seed=120 + p) for p in range(P)]
    dev = torch.device("cuda", 0)
    import socket
    with socket.socket() as sock:          # a free port, so that a stale listener cannot make this flaky
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        eng = Engine(**engine_kwargs(problems[0], 0, P, N, n))
        eng.set_paths(np.stack([p["table"] for p in problems]))
        x0_h = np.stack([p["x0"] for p in problems])
        U_h = np.ascontiguousarray(np.stack([p["U"] for p in problems]).transpose(0, 2, 3, 1))
        want = eng.solve(x0_h, U_h, layout=1)
        shard = ShardedRollout(eng, P, N, n, 1, index_offset=0, device=dev)
        shard.distributed = True  # force the collective path although world_size == 1
        x0, U = torch.tensor(x0_h, device=dev), torch.tensor(U_h, device=dev)
        rec = shard.step(x0, U, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(rec.cpu().numpy(), want["records"])
        np.testing.assert_array_equal(shard.costs.cpu().numpy(), want["costs"])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("layout", [0, 1])
@pytest.mark.parametrize("N,H", [(5000, 50), (300, 20)])
def test_softmin_weighted_mean(layout, N, H):
    import torch
    from acmpc_amd import Engine
    P, n = 2, H - 1
    problems = [make_problem(orc, "monza", H, N, seed=200 + p) for p in range(P)]
    problems[1]["U"][7, 2, 0] = np.nan  # a non-finite candidate must get weight 0
    dev = torch.device("cuda", 0)
    eng = Engine(**engine_kwargs(problems[0], 0, P, N, n, softmin_lambda=0.5))
    eng.set_paths(np.stack([p["table"] for p in problems]))
    U_h = np.stack([p["U"] for p in problems])
    U = torch.tensor(U_h if layout == 0 else np.ascontiguousarray(U_h.transpose(0, 2, 3, 1)), device=dev)
    x0 = torch.tensor(np.stack([p["x0"] for p in problems]), device=dev)
    costs = torch.empty(P, N, device=dev)
    keys = torch.empty(P, dtype=torch.int64, device=dev)
    mean = torch.empty(P, n, 2, device=dev)
    wsum = torch.empty(P, dtype=torch.float64, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    eng.rollout_device(x0.data_ptr(), U.data_ptr(), P, N, n, layout, 0, costs.data_ptr(), keys.data_ptr(), s)
    eng.softmin_device(costs.data_ptr(), keys.data_ptr(), U.data_ptr(), P, N, n, layout, mean.data_ptr(),
                       wsum.data_ptr(), s)
    torch.cuda.synchronize()
    for p in range(P):
        c = costs[p].cpu().numpy()
        Up = np.where(np.isfinite(U_h[p]), U_h[p], 0.0)  # weight 0 x NaN control: excluded, as on the device
        w = orc.softmin_weights(c, 0.5).astype(np.float64)
        want = np.tensordot(w, Up.astype(np.float64), axes=(0, 0)) / w.sum()
        got = mean[p].cpu().numpy()
        np.testing.assert_allclose(got, want, rtol=2e-5, atol=1e-6)
        np.testing.assert_allclose(wsum[p].item(), w.sum(), rtol=1e-5)


def test_softmin_over_shards_equals_the_unsharded_mean():
    """SURVEY 8e's softmin variant on one card: three uneven shards, the keys' minimum as the all-reduce would leave it,
    each shard's softmin kernels on its own slice, the payloads combined in shard order - equal to the unsharded
    kernel's mean to float32 rounding and to the oracle; a problem with no finite cost falls back to the plain mean."""
    import torch
    from acmpc_amd import Engine
    from acmpc_amd.sharding import combine_softmin, softmin_payload
    P, H, N, lam = 3, 30, 5000, 0.5
    n = H - 1
    problems = [make_problem(orc, "monza", H, N, seed=610 + p) for p in range(P)]
    problems[2]["U"][:, 0, 0] = np.nan   # problem 2: every candidate non-finite -> every weight zero
    dev = torch.device("cuda", 0)
    s = torch.cuda.current_stream().cuda_stream
    U_h = np.ascontiguousarray(np.stack([p["U"] for p in problems]).transpose(0, 2, 3, 1))   # step-major
    x0 = torch.tensor(np.stack([p["x0"] for p in problems]), device=dev)
    tables = np.stack([p["table"] for p in problems])

    def run(offset, count, gkeys=None):
        eng = Engine(**engine_kwargs(problems[0], 0, P, count, n, softmin_lambda=lam))
        eng.set_paths(tables)
        U = torch.tensor(np.ascontiguousarray(U_h[..., offset:offset + count]), device=dev)
        costs = torch.empty(P, count, device=dev)
        keys = torch.empty(P, dtype=torch.int64, device=dev)
        eng.rollout_device(x0.data_ptr(), U.data_ptr(), P, count, n, 1, offset, costs.data_ptr(), keys.data_ptr(), s)
        return eng, U, costs, keys

    full = run(0, N)
    mean_full = torch.empty(P, n, 2, device=dev)
    wsum_full = torch.empty(P, dtype=torch.float64, device=dev)
    full[0].softmin_device(full[2].data_ptr(), full[3].data_ptr(), full[1].data_ptr(), P, N, n, 1, mean_full.data_ptr(),
                           wsum_full.data_ptr(), s)
    shards = [run(0, 2100), run(2100, 1900), run(4000, 1000)]
    gkeys = torch.stack([sh[3] for sh in shards]).min(dim=0).values.contiguous()
    assert torch.equal(gkeys, full[3])
    payloads = []
    for (eng, U, costs, _), count in zip(shards, (2100, 1900, 1000)):
        mean = torch.empty(P, n, 2, device=dev)
        wsum = torch.empty(P, dtype=torch.float64, device=dev)
        eng.softmin_device(costs.data_ptr(), gkeys.data_ptr(), U.data_ptr(), P, count, n, 1, mean.data_ptr(),
                           wsum.data_ptr(), s)
        payloads.append(softmin_payload(mean, wsum, count))
    mean, wsum = combine_softmin(payloads, n)
    again = combine_softmin(payloads, n)
    torch.cuda.synchronize()
    assert torch.equal(mean.view(torch.int32), again[0].view(torch.int32)) and torch.equal(wsum, again[1])   # bits
    np.testing.assert_allclose(mean.cpu().numpy(), mean_full.cpu().numpy(), rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(wsum.cpu().numpy(), wsum_full.cpu().numpy(), rtol=1e-12)
    assert wsum[2].item() == 0.0 and wsum[0].item() > 0.0
    for p in range(2):
        want = orc.softmin_mean(full[2][p].cpu().numpy(), np.moveaxis(U_h[p], -1, 0), lam)
        np.testing.assert_allclose(mean[p].cpu().numpy(), want, rtol=2e-5, atol=1e-6)


@pytest.mark.parametrize("layout", [0, 1])
def test_device_sampler_matches_its_restatement(layout):
    """acmpc_sample_device == the oracle's restatement bit for bit (round 4: the Box-Muller transform is specified -
    polynomial logarithm, correctly rounded square root, sin / cos reduced in turns - and restated with exact fmaf),
    structural guarantees, and regeneration of a shard from indices alone."""
    import torch
    from acmpc_amd import Engine
    P, H, N = 2, 50, 1000
    n = H - 1
    problems = [make_problem(orc, "monza", H, 4, seed=400 + p) for p in range(P)]
    dev = torch.device("cuda", 0)
    eng = Engine(**engine_kwargs(problems[0], 0, P, N, n))
    eng.set_paths(np.stack([p["table"] for p in problems]))
    u_ref = np.stack([np.stack([p["table"][orc.ROW_V], p["table"][orc.ROW_KAPPA]], axis=1) for p in problems]).astype(np.float32)
    centre = (u_ref + np.array([-1.0, 0.002], dtype=np.float32)).astype(np.float32)
    d_centre, d_ref = torch.tensor(centre, device=dev), torch.tensor(u_ref, device=dev)
    shape = (P, N, n, 2) if layout == 0 else (P, n, 2, N)
    U = torch.empty(shape, device=dev)
    sigma, seed, rnd = (3.0, 0.01), 0x1234567899, 2
    s = torch.cuda.current_stream().cuda_stream
    eng.sample_device(d_centre.data_ptr(), 2 * n, d_ref.data_ptr(), P, N, n, layout, 0, sigma, seed, rnd, U.data_ptr(), s)
    torch.cuda.synchronize()
    got = U.cpu().numpy() if layout == 0 else U.cpu().numpy().transpose(0, 3, 1, 2)
    lo, hi = problems[0]["u_lo"], problems[0]["u_hi"]
    for p in range(P):
        want = orc.sample_candidates(centre[p], u_ref[p], N, 0, p, rnd, seed, sigma, lo, hi)
        np.testing.assert_array_equal(got[p], want)
        np.testing.assert_array_equal(got[p, 0], np.clip(centre[p], lo.astype(np.float32), hi.astype(np.float32)))
        np.testing.assert_array_equal(got[p, 1], np.clip(u_ref[p], lo.astype(np.float32), hi.astype(np.float32)))
        assert (got[p] >= lo.astype(np.float32)).all() and (got[p] <= hi.astype(np.float32)).all()
    # perturbations are smooth along the horizon and grow with the amplitude level
    dev_k = got[0, 8:, :, 1] - np.clip(centre[0, :, 1], lo[1], hi[1])
    assert np.abs(np.diff(dev_k, axis=1)).mean() < 0.25 * np.abs(dev_k).mean()
    # a shard regenerates exactly the candidates its global indices name
    part = torch.empty((P, 300, n, 2) if layout == 0 else (P, n, 2, 300), device=dev)
    eng.sample_device(d_centre.data_ptr(), 2 * n, d_ref.data_ptr(), P, 300, n, layout, 500, sigma, seed, rnd,
                      part.data_ptr(), s)
    torch.cuda.synchronize()
    got_part = part.cpu().numpy() if layout == 0 else part.cpu().numpy().transpose(0, 3, 1, 2)
    np.testing.assert_array_equal(got_part, got[:, 500:800])


@pytest.mark.parametrize("mode,window,P,H,N,rounds", [
    (0, None, 3, 50, 2048, 4), (1, None, 3, 50, 2048, 4), (1, (2, 5), 3, 50, 2048, 4),
    (0, None, 2, 3, 65, 2),        # shortest horizon, a ragged last wave
    (0, None, 1, 100, 4096, 3),    # the mapping horizon: two 64-step chunks in the finalize
    (0, None, 5, 20, 1, 2),        # a single candidate: one workgroup per problem draws the first and last ticket
    (1, (1, 2), 2, 66, 130, 1),    # one round only: the full record straight from the fused launch
    (0, None, 1, 1025, 256, 2),    # the longest horizon the handle accepts
    (1, (2, 5), 1, 1025, 256, 2),  # ... where mode T's tables + record image exceed one workgroup's LDS: the
                                   # finalize runs as its own launch again
])
def test_optimize_equals_the_manual_round_loop(mode, window, P, H, N, rounds):
    """acmpc_optimize (one fused sample + rollout + finalize launch per round, the last workgroup of a problem
    writing its record) == sample_device -> solve_device per round with the incumbent fed back, bit for bit; and
    the winner's cost never increases from round to round."""
    import torch
    from acmpc_amd import Engine, _capi
    n = H - 1
    problems = [make_problem(orc, "silverstone", H, 4, seed=500 + p) for p in range(P)]
    dev = torch.device("cuda", 0)
    eng = Engine(**engine_kwargs(problems[0], mode, P, N, n, nn_window=window))
    eng.set_paths(np.stack([p["table"] for p in problems]))
    u_ref = np.stack([np.stack([p["table"][orc.ROW_V], p["table"][orc.ROW_KAPPA]], axis=1) for p in problems]).astype(np.float32)
    x0 = np.stack([p["x0"] if mode == 0 else p["pose0"] for p in problems])
    sigma, seed = (3.0, 0.01), 42
    out = eng.optimize(x0, u_ref, u_ref, N, rounds, sigma, shrink=0.5, seed=seed)
    R = _capi.record_floats(n)
    d_x0, d_ref = torch.tensor(x0, device=dev), torch.tensor(u_ref, device=dev)
    U = torch.empty(P, n, 2, N, device=dev)
    rec = torch.empty(P, R, device=dev)
    keys = torch.empty(P, dtype=torch.int64, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    costs = []
    for r in range(rounds):
        centre_ptr, stride = (d_ref.data_ptr(), 2 * n) if r == 0 else (rec.data_ptr() + 4 * _capi.REC_HEADER, R)
        eng.sample_device(centre_ptr, stride, d_ref.data_ptr(), P, N, n, 1, 0, (sigma[0] * 0.5**r, sigma[1] * 0.5**r),
                          seed, r, U.data_ptr(), s)
        eng.solve_device(d_x0.data_ptr(), U.data_ptr(), P, N, n, 1, 0, keys.data_ptr(), rec.data_ptr(), s)
        torch.cuda.synchronize()
        costs.append(rec[:, 0].cpu().numpy().copy())
    np.testing.assert_array_equal(out["records"], rec.cpu().numpy())
    for a, b in zip(costs, costs[1:]):
        assert (b <= a).all()


@pytest.mark.parametrize("mode,window,P,H,N,rounds", [
    (0, None, 1, 50, 16384, 3),     # the closed-loop shape: 256 workgroups, chained rounds
    (0, None, 2, 150, 4096, 3),     # horizon beyond the register-staged operands (the loop behind them)
    (1, (2, 5), 2, 50, 4096, 3),    # mode T: tables + trace need more than the default 64 kB of LDS
    (0, None, 1, 50, 16448, 2),     # 257 workgroups: traced, but one too many for the chained form
    (0, None, 3, 80, 1000, 4),      # ragged last wave, several problems
])
def test_optimize_forms_agree(mode, window, P, H, N, rounds, monkeypatch):
    """The fused rounds come in three forms - the record copied out of the winning workgroup's trace with the rounds
    chained through the previous launch's keys (default), traced but every round finalized, and the winner re-drawn and
    re-rolled by the last workgroup - and all three write the same records, bit for bit."""
    from acmpc_amd import Engine
    n = H - 1
    problems = [make_problem(orc, "silverstone", H, 4, seed=900 + p) for p in range(P)]
    u_ref = np.stack([np.stack([p["table"][orc.ROW_V], p["table"][orc.ROW_KAPPA]], axis=1) for p in problems]).astype(np.float32)
    x0 = np.stack([p["x0"] if mode == 0 else p["pose0"] for p in problems])

    def solve():
        eng = Engine(**engine_kwargs(problems[0], mode, P, N, n, nn_window=window))
        eng.set_paths(np.stack([p["table"] for p in problems]))
        first = eng.optimize(x0, u_ref, u_ref, N, rounds, (3.0, 0.01), shrink=0.5, seed=7)["records"].copy()
        again = eng.optimize(x0, u_ref, None, N, rounds, (1.0, 0.005), shrink=0.5, seed=8)["records"].copy()
        eng.close()
        return first, again

    default = solve()
    monkeypatch.setenv("ACMPC_NO_CHAINED_ROUNDS", "1")
    unchained = solve()
    monkeypatch.setenv("ACMPC_NO_TRACED_FINALIZE", "1")
    rerolled = solve()
    for a, b, c in zip(default, unchained, rerolled):
        assert np.isfinite(a[:, 0]).all()
        np.testing.assert_array_equal(a, b)
        np.testing.assert_array_equal(a, c)


def test_softmin_centre_update_equals_the_manual_round_loop():
    """centre_update="softmin": each round samples round the softmin-weighted mean of the previous round and keeps
    the previous winner as candidate 1; equals the loop built from sample/solve/softmin device calls bit for bit, the
    winner's cost never increases, and the mean itself is checked against the oracle's weighted reduction."""
    import torch
    from acmpc_amd import Engine, _capi
    P, H, N, rounds = 3, 50, 2048, 4
    n = H - 1
    problems = [make_problem(orc, "silverstone", H, 4, seed=520 + p) for p in range(P)]
    dev = torch.device("cuda", 0)
    eng = Engine(**engine_kwargs(problems[0], 0, P, N, n, centre_update="softmin"))
    eng.set_paths(np.stack([p["table"] for p in problems]))
    u_ref = np.stack([np.stack([p["table"][orc.ROW_V], p["table"][orc.ROW_KAPPA]], axis=1) for p in problems]).astype(np.float32)
    x0 = np.stack([p["x0"] for p in problems])
    sigma, seed = (3.0, 0.01), 43
    out = eng.optimize(x0, u_ref, u_ref, N, rounds, sigma, shrink=0.5, seed=seed)
    R = _capi.record_floats(n)
    d_x0, d_ref = torch.tensor(x0, device=dev), torch.tensor(u_ref, device=dev)
    U = torch.empty(P, n, 2, N, device=dev)
    rec = torch.empty(P, R, device=dev)
    keys = torch.empty(P, dtype=torch.int64, device=dev)
    cost = torch.empty(P, N, device=dev)
    mean = d_ref.clone()
    s = torch.cuda.current_stream().cuda_stream
    best = []
    for r in range(rounds):
        ref = d_ref if r == 0 else rec[:, _capi.REC_HEADER:_capi.REC_HEADER + 2 * n].contiguous()
        eng.sample_device(mean.data_ptr(), 2 * n, ref.data_ptr(), P, N, n, 1, 0, (sigma[0] * 0.5**r, sigma[1] * 0.5**r),
                          seed, r, U.data_ptr(), s)
        eng.solve_device(d_x0.data_ptr(), U.data_ptr(), P, N, n, 1, cost.data_ptr(), keys.data_ptr(), rec.data_ptr(), s)
        eng.softmin_device(cost.data_ptr(), keys.data_ptr(), U.data_ptr(), P, N, n, 1, mean.data_ptr(), 0, s)
        torch.cuda.synchronize()
        best.append(rec[:, 0].cpu().numpy().copy())
        if r == 0:
            Uh, ch = U.cpu().numpy(), cost.cpu().numpy()
            for p in range(P):
                want = orc.softmin_mean(ch[p], np.moveaxis(Uh[p], -1, 0), 0.5)
                np.testing.assert_allclose(mean[p].cpu().numpy(), want, rtol=1e-5, atol=1e-6)
    np.testing.assert_array_equal(out["records"], rec.cpu().numpy())
    for a, b in zip(best, best[1:]):
        assert (b <= a).all()
    # the eager (no graph) form runs the same launches
    os.environ["ACMPC_NO_GRAPH"] = "1"
    try:
        again = eng.optimize(x0, u_ref, u_ref, N, rounds, sigma, shrink=0.5, seed=seed)
    finally:
        del os.environ["ACMPC_NO_GRAPH"]
    np.testing.assert_array_equal(again["records"], out["records"])


def test_profile_hooks_and_pipelined_rollout():
    """acmpc_profile_enable/collect time exactly the launches they were armed for, and the pipelined (side-stream)
    driver produces the same records as plain stream order."""
    import torch
    from acmpc_amd import Engine
    from acmpc_amd.sharding import PipelinedRollout, ShardedRollout
    P, H, N = 8, 50, 4096
    n = H - 1
    problems = [make_problem(orc, "monza", H, 4, seed=600 + p) for p in range(P)]
    dev = torch.device("cuda", 0)
    tables = np.stack([p["table"] for p in problems])
    x0 = torch.tensor(np.stack([p["x0"] for p in problems]), device=dev)
    rng = np.random.default_rng(0)
    Us = [torch.tensor((rng.standard_normal((P, n, 2, N)) * np.array([2.0, 0.01])[None, None, :, None]
                        + np.array([20.0, 0.0])[None, None, :, None]).astype(np.float32), device=dev) for _ in range(3)]
    engines = []
    for _ in range(2):
        eng = Engine(**engine_kwargs(problems[0], 0, P, N, n))
        eng.set_paths(tables)
        engines.append(eng)
    stream = torch.cuda.current_stream()
    serial = ShardedRollout(engines[0], P, N, n, 1, 0, dev)
    engines[0].profile_enable(2)
    want = []
    for U in Us:
        want.append(serial.step(x0, U, stream.cuda_stream).clone())
    torch.cuda.synchronize()
    times = engines[0].profile_collect()
    assert times.shape == (2,) and (times > 0).all() and (times < 5.0).all()  # armed for 2 of the 3 launches
    assert engines[0].profile_collect().shape == (0,)
    engines[0].profile_enable(0)
    pipe = PipelinedRollout(engines, P, N, n, 1, 0, dev)
    pipe.bind_stream(stream)
    got = []
    for U in Us:
        slot = pipe.step(x0, U)
        pipe.drain()
        torch.cuda.synchronize()
        got.append(slot.records.clone())
    for a, b in zip(want, got):
        assert torch.equal(a, b)


@pytest.mark.parametrize("mode", [0, 1])
def test_regenerated_finalize_equals_finalize_from_the_matrix(mode):
    """Single-collective design: for sampled candidates, finalize that re-draws the winner from its index gives
    the same record as finalize that reads the control matrix - on the full batch and on two shards of it, where
    BOTH shards must produce the complete record without owning the winner."""
    import torch
    from acmpc_amd import Engine, _capi
    P, H, total = 3, 50, 3000
    n = H - 1
    problems = [make_problem(orc, "monza", H, 4, seed=700 + p) for p in range(P)]
    dev = torch.device("cuda", 0)
    tables = np.stack([p["table"] for p in problems])
    u_ref = torch.tensor(np.stack([np.stack([p["table"][orc.ROW_V], p["table"][orc.ROW_KAPPA]], axis=1)
                                   for p in problems]).astype(np.float32), device=dev)
    centre = (u_ref + torch.tensor([-0.7, 0.001], device=dev)).contiguous()
    x0 = torch.tensor(np.stack([p["x0"] if mode == 0 else p["pose0"] for p in problems]), device=dev)
    s = torch.cuda.current_stream().cuda_stream
    R = _capi.record_floats(n)
    sigma, seed, rnd = (3.0, 0.01), 99, 1

    def run(offset, count):
        eng = Engine(**engine_kwargs(problems[0], mode, P, count, n))
        eng.set_paths(tables)
        U = torch.empty(P, n, 2, count, device=dev)
        eng.sample_device(centre.data_ptr(), 2 * n, u_ref.data_ptr(), P, count, n, 1, offset, sigma, seed, rnd, U.data_ptr(), s)
        keys = torch.empty(P, dtype=torch.int64, device=dev)
        eng.rollout_device(x0.data_ptr(), U.data_ptr(), P, count, n, 1, offset, 0, keys.data_ptr(), s)
        return eng, U, keys

    eng, U, keys = run(0, total)
    from_matrix = torch.empty(P, R, device=dev)
    eng.finalize_device(keys.data_ptr(), x0.data_ptr(), U.data_ptr(), P, total, n, 1, 0, from_matrix.data_ptr(), s)
    regenerated = torch.empty(P, R, device=dev)
    eng.finalize_sampled_device(keys.data_ptr(), x0.data_ptr(), centre.data_ptr(), 2 * n, u_ref.data_ptr(), P, total, n,
                                sigma, seed, rnd, regenerated.data_ptr(), s)
    torch.cuda.synchronize()
    assert torch.equal(from_matrix, regenerated)
    shards = [run(0, 1800), run(1800, 1200)]
    gkeys = torch.minimum(shards[0][2], shards[1][2])
    assert torch.equal(gkeys, keys)
    for eng_s, _, _ in shards:
        rec = torch.empty(P, R, device=dev)
        eng_s.finalize_sampled_device(gkeys.data_ptr(), x0.data_ptr(), centre.data_ptr(), 2 * n, u_ref.data_ptr(), P,
                                      eng_s.params.max_candidates, n, sigma, seed, rnd, rec.data_ptr(), s)
        torch.cuda.synchronize()
        rec[:, _capi.REC_NFEASIBLE] = from_matrix[:, _capi.REC_NFEASIBLE]  # feasible counts are per shard
        assert torch.equal(rec, from_matrix)


def test_sharded_optimizer_two_shards_equal_one():
    """ShardedOptimizer on one GPU: a single 'rank' holding all 4 096 candidates vs two emulated ranks holding 2 048
    each whose keys are min-reduced by hand - identical records after every round, hence identical plans; and the
    single-rank result equals acmpc_optimize."""
    import torch
    from acmpc_amd import Engine, _capi
    from acmpc_amd.sharding import ShardedOptimizer
    P, H, N, rounds = 2, 50, 4096, 3
    n = H - 1
    problems = [make_problem(orc, "monza", H, 4, seed=800 + p) for p in range(P)]
    dev = torch.device("cuda", 0)
    tables = np.stack([p["table"] for p in problems])
    u_ref_h = np.stack([np.stack([p["table"][orc.ROW_V], p["table"][orc.ROW_KAPPA]], axis=1) for p in problems]).astype(np.float32)
    x0_h = np.stack([p["x0"] for p in problems])
    u_ref, x0 = torch.tensor(u_ref_h, device=dev), torch.tensor(x0_h, device=dev)
    sigma, seed = (3.0, 0.01), 5
    s = torch.cuda.current_stream().cuda_stream

    def engine(count):
        e = Engine(**engine_kwargs(problems[0], 0, P, count, n))
        e.set_paths(tables)
        return e

    whole = ShardedOptimizer(engine(N), P, N, n, 0, dev)
    rec_whole = whole.solve(x0, u_ref, u_ref, rounds, sigma, seed=seed, stream=s).clone()
    want = engine(N).optimize(x0_h, u_ref_h, u_ref_h, N, rounds, sigma, shrink=0.5, seed=seed)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(rec_whole.cpu().numpy(), want["records"])

    # two emulated ranks, advanced round by round with the all-reduce(MIN) done by hand
    halves = [ShardedOptimizer(engine(N // 2), P, N // 2, n, off, dev) for off in (0, N // 2)]
    for h in halves:
        h.shard.distributed = True
    R = _capi.record_floats(n)
    scale = 1.0
    for r in range(rounds):
        sig = (sigma[0] * scale, sigma[1] * scale)
        for h in halves:
            sh = h.shard
            cptr, stride = (u_ref.data_ptr(), 2 * n) if r == 0 else (sh.records.data_ptr() + 4 * _capi.REC_HEADER, R)
            sh.engine.sample_device(cptr, stride, u_ref.data_ptr(), P, sh.N, n, 1, sh.offset, sig, seed, r, h.U.data_ptr(), s)
            sh.rollout(x0, h.U, s)
        gkeys = torch.minimum(halves[0].shard.keys, halves[1].shard.keys)
        for h in halves:
            sh = h.shard
            cptr, stride = (u_ref.data_ptr(), 2 * n) if r == 0 else (sh.records.data_ptr() + 4 * _capi.REC_HEADER, R)
            sh.engine.finalize_sampled_device(gkeys.data_ptr(), x0.data_ptr(), cptr, stride, u_ref.data_ptr(), P, sh.N,
                                              n, sig, seed, r, sh.records.data_ptr(), s)
        torch.cuda.synchronize()
        a, b = halves[0].shard.records.clone(), halves[1].shard.records.clone()
        a[:, _capi.REC_NFEASIBLE] = 0
        b[:, _capi.REC_NFEASIBLE] = 0
        assert torch.equal(a, b)
        scale *= 0.5
    final = halves[0].shard.records.clone()
    final[:, _capi.REC_NFEASIBLE] = rec_whole[:, _capi.REC_NFEASIBLE]
    assert torch.equal(final, rec_whole)


def test_reduce_across_ranks_with_a_caller_owned_rccl_communicator():
    """acmpc_reduce_across_ranks: the C-ABI form of the step's one collective, for hosts that hold an `ncclComm_t`
    themselves.  One rank is all a single GPU allows: the communicator is created straight on RCCL's C API, the keys
    of a rollout go through ncclAllReduce(MIN, int64) inside the library and must come back unchanged, and the
    finalize that follows must equal the plain solve."""
    import ctypes as C
    import torch
    from acmpc_amd import Engine, EngineError
    P, H, N = 3, 50, 1024
    n = H - 1
    problems = [make_problem(orc, "spa", H, N, seed=880 + p) for p in range(P)]
    dev = torch.device("cuda", 0)
    eng = Engine(**engine_kwargs(problems[0], 0, P, N, n))
    eng.set_paths(np.stack([p["table"] for p in problems]))
    x0 = torch.tensor(np.stack([p["x0"] for p in problems]), device=dev)
    U = torch.tensor(np.ascontiguousarray(np.stack([p["U"] for p in problems]).transpose(0, 2, 3, 1)), device=dev)
    want = eng.solve(x0.cpu().numpy(), U.cpu().numpy(), layout=1)

    rccl = C.CDLL("librccl.so.1")

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]

    uid, comm = UniqueId(), C.c_void_p()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        keys = torch.empty(P, dtype=torch.int64, device=dev)
        rec = torch.empty(P, want["records"].shape[1], device=dev)
        s = torch.cuda.current_stream().cuda_stream
        eng.rollout_device(x0.data_ptr(), U.data_ptr(), P, N, n, 1, 0, 0, keys.data_ptr(), s)
        before = keys.clone()
        eng.reduce_across_ranks(comm.value, keys.data_ptr(), P, s)
        eng.finalize_device(keys.data_ptr(), x0.data_ptr(), U.data_ptr(), P, N, n, 1, 0, rec.data_ptr(), s)
        torch.cuda.synchronize()
        assert torch.equal(keys, before)
        np.testing.assert_array_equal(rec.cpu().numpy(), want["records"])
        with pytest.raises(EngineError) as e:
            eng.reduce_across_ranks(0, keys.data_ptr(), P, s)
        assert e.value.code == -1
    finally:
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)



def test_the_library_makes_its_own_communicator_and_bench_can_use_it():
    """`LibraryCommunicator` (acmpc_rccl_unique_id / acmpc_rccl_comm_create: the RCCL acmpc_reduce_across_ranks resolves) with
    the one rank a single GPU allows, `ShardedRollout.use_library_collective`: the step then takes the multi-rank path -
    rollout, keys, all-reduce(MIN) through the C ABI, finalize - and returns the plain step's records; and
    `bench.py --collective capi` runs its timed steps that way and says so in its line."""
    import json
    import subprocess
    import sys
    import torch
    from acmpc_amd import Engine, _capi
    from acmpc_amd.sharding import LibraryCommunicator, ShardedRollout
    P, H, N = 5, 50, 2048
    n = H - 1
    problems = [make_problem(orc, "monza", H, 4, seed=930 + p) for p in range(P)]
    dev = torch.device("cuda", 0)
    s = torch.cuda.current_stream().cuda_stream
    x0 = torch.tensor(np.stack([p["x0"] for p in problems]), device=dev)
    u_ref = torch.tensor(np.stack([np.stack([p["table"][orc.ROW_V], p["table"][orc.ROW_KAPPA]], axis=1) for p in problems]),
                         dtype=torch.float32, device=dev).contiguous()
    records = []
    comm = LibraryCommunicator(0)
    assert comm.world == 1 and comm.handle
    for library in (False, True):
        eng = Engine(**engine_kwargs(problems[0], 0, P, N, n))
        eng.set_paths(np.stack([p["table"] for p in problems]))
        slot = ShardedRollout(eng, P, N, n, 1, index_offset=0, device=dev)
        slot.use_sampler(u_ref, u_ref, (2.0, 0.01), 31, 0)
        if library:
            slot.use_library_collective(comm.handle)
            assert slot.distributed
        U = torch.empty(P, n, 2, N, device=dev)
        slot.sample(U, s)
        slot.step(x0, U, s)
        torch.cuda.synchronize()
        records.append(slot.records.cpu().numpy().copy())
        eng.close()
    comm.close()
    np.testing.assert_array_equal(records[0], records[1])
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    proc = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--collective", "capi", "--poses", "64", "--steps", "10",
                           "--warmup", "3", "--no-cpu-baseline", "--no-single-solve"], capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stderr[-2000:]
    line = json.loads([l for l in proc.stdout.splitlines() if l.startswith("{")][-1])
    assert line["collective"]["entry"] == "acmpc_reduce_across_ranks" and line["collective"]["world_size"] == 1
    assert line["value"] > 1e8 and line["roofline"]["kernel"] == "rollout_kernel"     # (no stream form: the reduce sits between)


def test_sharded_optimizer_two_ranks_equal_the_unsharded_solve():
    """The multi-GPU closed-loop solve rehearsed with two ranks on the one card (gloo carries the key all-reduce):
    `tests/sharded_optimizer_ranks.py` under torch.distributed.run; both modes, three rounds."""
    import socket
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(here, "sharded_optimizer_ranks.py")]
    proc = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, (proc.stdout + proc.stderr)[-3000:]
    assert "sharded optimizer ok" in proc.stdout


def test_mode_t_softmin_optimize_on_a_fresh_handle_and_after_new_paths():
    """Mode T with the exhaustive (verified-window) search and centre_update="softmin": acmpc_optimize's captured graph
    runs the three-kernel rounds, whose rollout reads the verified-search thresholds - they must be on the device when
    the graph runs on a handle that has never made another device call, and be the NEW paths' thresholds after a
    second set_paths.  Oracle: the same rounds through sample / solve / softmin device calls."""
    import torch
    from acmpc_amd import Engine, _capi
    P, H, N, rounds = 2, 50, 1024, 3
    n = H - 1
    dev = torch.device("cuda", 0)
    sigma, seed = (3.0, 0.01), 47

    def manual(eng, x0, u_ref):
        R = _capi.record_floats(n)
        d_x0, d_ref = torch.tensor(x0, device=dev), torch.tensor(u_ref, device=dev)
        U = torch.empty(P, n, 2, N, device=dev)
        rec = torch.empty(P, R, device=dev)
        keys = torch.empty(P, dtype=torch.int64, device=dev)
        cost = torch.empty(P, N, device=dev)
        mean = d_ref.clone()
        s = torch.cuda.current_stream().cuda_stream
        for r in range(rounds):
            ref = d_ref if r == 0 else rec[:, _capi.REC_HEADER:_capi.REC_HEADER + 2 * n].contiguous()
            eng.sample_device(mean.data_ptr(), 2 * n, ref.data_ptr(), P, N, n, 1, 0,
                              (sigma[0] * 0.5**r, sigma[1] * 0.5**r), seed, r, U.data_ptr(), s)
            eng.solve_device(d_x0.data_ptr(), U.data_ptr(), P, N, n, 1, cost.data_ptr(), keys.data_ptr(), rec.data_ptr(), s)
            eng.softmin_device(cost.data_ptr(), keys.data_ptr(), U.data_ptr(), P, N, n, 1, mean.data_ptr(), 0, s)
            torch.cuda.synchronize()
        return rec.cpu().numpy()

    def inputs(seed0):
        problems = [make_problem(orc, "silverstone", H, 4, seed=seed0 + p) for p in range(P)]
        tables = np.stack([p["table"] for p in problems])
        u_ref = np.stack([np.stack([p["table"][orc.ROW_V], p["table"][orc.ROW_KAPPA]], axis=1)
                          for p in problems]).astype(np.float32)
        return problems, tables, u_ref, np.stack([p["pose0"] for p in problems])

    problems, tables, u_ref, x0 = inputs(560)
    fresh = Engine(**engine_kwargs(problems[0], 1, P, N, n, centre_update="softmin"))   # exhaustive search: thresholds
    fresh.set_paths(tables)
    first = fresh.optimize(x0, u_ref, u_ref, N, rounds, sigma, shrink=0.5, seed=seed)["records"].copy()   # FIRST device call
    checker = Engine(**engine_kwargs(problems[0], 1, P, N, n, centre_update="softmin"))
    checker.set_paths(tables)
    np.testing.assert_array_equal(first, manual(checker, x0, u_ref))
    # new paths on the same handle: the graph is replayed, the thresholds must be the new ones
    problems2, tables2, u_ref2, x02 = inputs(590)
    fresh.set_paths(tables2)
    second = fresh.optimize(x02, u_ref2, u_ref2, N, rounds, sigma, shrink=0.5, seed=seed)["records"].copy()
    checker.set_paths(tables2)
    np.testing.assert_array_equal(second, manual(checker, x02, u_ref2))
    assert not np.array_equal(first, second)
