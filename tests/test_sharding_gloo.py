"""N > 1 path on CPU: two gloo ranks each hold half of the candidates; the all-reduce(MIN) of packed keys plus
the all-reduce(SUM) of owner-masked records must reproduce the single-process answer.  The local evaluator is the
oracle here (there is no GPU in this container); on the GPU the same `global_select` runs over RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, out_dir):
    for p in (os.path.join(ROOT, "ac-mpc_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import acmpc_oracle as orc
    from acmpc_amd import _capi
    from acmpc_amd.sharding import global_select, shard_range
    from test_support import make_problem

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    P, H = 3, 20
    n = H - 1
    R = _capi.record_floats(n)
    problems = [make_problem(orc, "monza", H, total, seed=50 + p) for p in range(P)]
    offset, count = shard_range(total, rank, world)
    local = []
    for prob in problems:
        cfg = prob["cfg"]
        coef = orc.coefficients_spatial(prob["table"], prob["limits"].margin)
        U = prob["U"][offset:offset + count]
        cost, viol, X = orc.rollout_spatial(prob["x0"], coef, U, cfg["step_cost"], cfg["r_term"], cfg["final_cost"],
                                            prob["u_lo"], prob["u_hi"], 1e6, dtype=np.float32, return_states=True)
        local.append((cost, viol, X, U))
    keys = torch.tensor([_capi.pack_key(c[0][orc.pick_best(c[0])[0]], offset + orc.pick_best(c[0])[0]) for c in local],
                        dtype=torch.int64)

    def make_records(global_keys):
        rec = torch.zeros(P, R)
        for p, (cost, viol, X, U) in enumerate(local):
            rec[p, _capi.REC_NFEASIBLE] = float(np.count_nonzero(viol == 0))
            j = _capi.key_index(int(global_keys[p])) - offset
            if 0 <= j < count:
                rec[p, _capi.REC_COST], rec[p, _capi.REC_VIOLATION] = float(cost[j]), float(viol[j])
                rec[p, _capi.REC_OWNER] = 1.0
                rec[p, _capi.REC_HEADER:_capi.REC_HEADER + 2 * n] = torch.from_numpy(U[j].ravel())
                rec[p, _capi.REC_HEADER + 2 * n:] = torch.from_numpy(X[j].ravel())
        return rec

    gkeys, records = global_select(keys, make_records)
    np.save(os.path.join(out_dir, "keys_%d.npy" % rank), gkeys.numpy())
    np.save(os.path.join(out_dir, "records_%d.npy" % rank), records.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("world,total", [(2, 256), (2, 257), (8, 1003)])
def test_sharded_selection_equals_single_process(tmp_path, world, total):
    """Two ranks, and the real width: eight (BASELINE configs[3]'s GPU count; ragged shares of 126 / 125 candidates)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import acmpc_oracle as orc
    from acmpc_amd import _capi
    from test_support import make_problem

    mp.spawn(_worker, args=(world, _free_port(), total, str(tmp_path)), nprocs=world, join=True)
    keys = [np.load(tmp_path / ("keys_%d.npy" % r)) for r in range(world)]
    recs = [np.load(tmp_path / ("records_%d.npy" % r)) for r in range(world)]
    for r in range(1, world):
        np.testing.assert_array_equal(keys[0], keys[r])
        np.testing.assert_array_equal(recs[0], recs[r])
    P, H = 3, 20
    n = H - 1
    for p in range(P):
        prob = make_problem(orc, "monza", H, total, seed=50 + p)
        cfg = prob["cfg"]
        coef = orc.coefficients_spatial(prob["table"], prob["limits"].margin)
        cost, viol, X = orc.rollout_spatial(prob["x0"], coef, prob["U"], cfg["step_cost"], cfg["r_term"],
                                            cfg["final_cost"], prob["u_lo"], prob["u_hi"], 1e6, dtype=np.float32,
                                            return_states=True)
        best, _ = orc.pick_best(cost)
        assert _capi.key_index(int(keys[0][p])) == best
        out = _capi.split_record(recs[0][p], n)
        assert out["owner"] == 1.0 and out["cost"] == cost[best] and out["n_feasible"] == np.count_nonzero(viol == 0)
        np.testing.assert_array_equal(out["u"], prob["U"][best])
        np.testing.assert_array_equal(out["x"], X[best])


def test_shard_range_partitions_exactly():
    from acmpc_amd.sharding import shard_range
    for total in (1, 7, 4096, 262144, 262145):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == total
            for (o1, c1), (o2, _) in zip(spans, spans[1:]):
                assert o1 + c1 == o2


def _worker_sampled(rank, world, port, total, out_dir):
    for p in (os.path.join(ROOT, "ac-mpc_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import acmpc_oracle as orc
    from acmpc_amd import _capi
    from acmpc_amd.sharding import global_select_sampled, shard_range
    from test_support import make_problem

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    P, H = 2, 20
    n = H - 1
    R = _capi.record_floats(n)
    sigma, seed, rnd = (2.0, 0.01), 7, 3
    problems = [make_problem(orc, "spa", H, 4, seed=60 + p) for p in range(P)]
    offset, count = shard_range(total, rank, world)

    def candidates(p, first, number):
        prob = problems[p]
        u_ref = np.stack([prob["table"][orc.ROW_V], prob["table"][orc.ROW_KAPPA]], axis=1)
        return orc.sample_candidates(u_ref, u_ref, number, first, p, rnd, seed, sigma, prob["u_lo"],
                                     prob["u_hi"]).astype(np.float32)

    def evaluate(p, U):
        prob, cfg = problems[p], problems[p]["cfg"]
        coef = orc.coefficients_spatial(prob["table"], prob["limits"].margin)
        return orc.rollout_spatial(prob["x0"], coef, U, cfg["step_cost"], cfg["r_term"], cfg["final_cost"], prob["u_lo"],
                                   prob["u_hi"], 1e6, dtype=np.float32, return_states=True)

    keys = []
    for p in range(P):
        cost, _, _ = evaluate(p, candidates(p, offset, count))   # this rank's slice, generated from indices alone
        j = orc.pick_best(cost)[0]
        keys.append(_capi.pack_key(cost[j], offset + j))
    keys = torch.tensor(keys, dtype=torch.int64)

    def regenerate(global_keys):
        rec = torch.zeros(P, R)
        for p in range(P):
            idx = _capi.key_index(int(global_keys[p]))
            U = candidates(p, idx, 1)                             # any rank can rebuild the winner
            cost, viol, X = evaluate(p, U)
            rec[p, _capi.REC_COST], rec[p, _capi.REC_VIOLATION] = float(cost[0]), float(viol[0])
            rec[p, _capi.REC_OWNER] = 1.0
            rec[p, _capi.REC_HEADER:_capi.REC_HEADER + 2 * n] = torch.from_numpy(U[0].ravel())
            rec[p, _capi.REC_HEADER + 2 * n:] = torch.from_numpy(X[0].ravel())
        return rec

    gkeys, records = global_select_sampled(keys, regenerate)
    np.save(os.path.join(out_dir, "skeys_%d.npy" % rank), gkeys.numpy())
    np.save(os.path.join(out_dir, "srecords_%d.npy" % rank), records.numpy())
    if rank == 0:   # single-process answer over all candidates, for the parent to compare with
        full = []
        for p in range(P):
            cost, _, _ = evaluate(p, candidates(p, 0, total))
            full.append(orc.pick_best(cost)[0])
        np.save(os.path.join(out_dir, "sfull.npy"), np.array(full))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,total", [(2, 301), (8, 2049)])
def test_single_collective_protocol_for_counter_based_candidates(tmp_path, world, total):
    """gloo ranks, ONE all-reduce(MIN): all ranks end with identical complete records and the global argmin."""
    from acmpc_amd import _capi
    mp.spawn(_worker_sampled, args=(world, _free_port(), total, str(tmp_path)), nprocs=world, join=True)
    keys = [np.load(tmp_path / ("skeys_%d.npy" % r)) for r in range(world)]
    recs = [np.load(tmp_path / ("srecords_%d.npy" % r)) for r in range(world)]
    for r in range(1, world):
        np.testing.assert_array_equal(keys[0], keys[r])
        np.testing.assert_array_equal(recs[0], recs[r])
    full = np.load(tmp_path / "sfull.npy")
    assert [_capi.key_index(int(k)) for k in keys[0]] == list(full)
    assert (recs[0][:, _capi.REC_OWNER] == 1.0).all()


def test_softmin_combine_over_shards_is_the_global_weighted_mean():
    """combine_softmin (softmin variant): per-shard (mean, weight sum, count) payloads in rank order give
    sum(w u) / sum(w) over all candidates; with every weight zero, the plain mean.  Host logic only (torch on CPU)."""
    sys.path.insert(0, os.path.join(ROOT, "ac-mpc_amd"))
    from acmpc_amd.sharding import combine_softmin, softmin_payload
    rng = np.random.default_rng(8)
    P, n, counts = 3, 7, (40, 25, 60)
    U = [rng.standard_normal((P, c, n, 2)) for c in counts]
    w = [rng.random((P, c)) for c in counts]
    for k in range(len(counts)):
        w[k][2] = 0.0                                     # problem 2: no finite cost anywhere
    w[1][0] = 0.0                                          # problem 0: one shard contributes nothing
    payloads = []
    for Uk, wk, c in zip(U, w, counts):
        ws = wk.sum(axis=1)
        local = np.where(ws[:, None, None] > 0, np.einsum("pc,pcij->pij", wk, Uk) / np.where(ws > 0, ws, 1)[:, None, None],
                         Uk.mean(axis=1))
        payloads.append(softmin_payload(torch.tensor(local, dtype=torch.float32), torch.tensor(ws), c))
    mean, wsum = combine_softmin(payloads, n)
    Uall, wall = np.concatenate(U, axis=1), np.concatenate(w, axis=1)
    for p in range(2):
        want = np.einsum("c,cij->ij", wall[p], Uall[p]) / wall[p].sum()
        np.testing.assert_allclose(mean[p].numpy(), want, rtol=2e-6, atol=1e-6)
        np.testing.assert_allclose(wsum[p].item(), wall[p].sum(), rtol=1e-12)
    np.testing.assert_allclose(mean[2].numpy(), Uall[2].mean(axis=0), rtol=2e-6, atol=1e-6)
    assert wsum[2].item() == 0.0 and mean.dtype == torch.float32 and tuple(mean.shape) == (P, n, 2)


# ---- BASELINE configs[3] at its real width: Nordschleife, 262 144 candidates x horizon 80 as 8 x 32 768 --------------------
def _worker_config4(rank, world, port, out_dir):
    """One rank of the 8-way split: its 32 768 counter-based candidates (global indices from its offset), rolled by the
    oracle (no GPU here); then all three things the ranks exchange - the keys (all-reduce MIN) with every rank re-drawing
    the winner, the owner-masked records (all-reduce SUM), and the softmin payloads (all-gather) combined in rank order."""
    for p in (os.path.join(ROOT, "ac-mpc_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import acmpc_oracle as orc
    from acmpc_amd import _capi
    from acmpc_amd.sharding import combine_softmin, global_select, global_select_sampled, shard_range, softmin_payload
    from test_support import make_problem

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    H, N = 80, 262144
    n = H - 1
    R = _capi.record_floats(n)
    sigma, seed, rnd, lam = (2.0, 0.01), 99, 0, 0.5
    prob = make_problem(orc, "nordschleife", H, 16, seed=4243)
    cfg = prob["cfg"]
    coef = orc.coefficients_spatial(prob["table"], prob["limits"].margin)
    u_ref = np.stack([prob["table"][orc.ROW_V], prob["table"][orc.ROW_KAPPA]], axis=1)
    offset, count = shard_range(N, rank, world)
    assert count == N // world and offset == rank * count            # 8 x 32 768, contiguous

    def candidates(first, number):
        return orc.sample_candidates(u_ref, u_ref, number, first, 0, rnd, seed, sigma, prob["u_lo"], prob["u_hi"])

    def evaluate(U):
        return orc.rollout_spatial(prob["x0"], coef, U, cfg["step_cost"], cfg["r_term"], cfg["final_cost"], prob["u_lo"],
                                   prob["u_hi"], 1e6, dtype=np.float32, return_states=True)

    U = candidates(offset, count)
    cost, viol, X = evaluate(U)
    j = orc.pick_best(cost)[0]
    local_key = torch.tensor([_capi.pack_key(cost[j], offset + j)], dtype=torch.int64)

    def regenerate(global_keys):                                       # protocol 1: every rank re-draws the winner
        idx = _capi.key_index(int(global_keys[0]))
        Uw = candidates(idx, 1)
        c, v, Xw = evaluate(Uw)
        rec = torch.zeros(1, R)
        rec[0, _capi.REC_COST], rec[0, _capi.REC_VIOLATION], rec[0, _capi.REC_OWNER] = float(c[0]), float(v[0]), 1.0
        rec[0, _capi.REC_HEADER:_capi.REC_HEADER + 2 * n] = torch.from_numpy(Uw[0].ravel())
        rec[0, _capi.REC_HEADER + 2 * n:] = torch.from_numpy(Xw[0].ravel())
        return rec

    def own_record(global_keys):                                       # protocol 2: the owner's record, zeros elsewhere
        rec = torch.zeros(1, R)
        rec[0, _capi.REC_NFEASIBLE] = float(np.count_nonzero(viol == 0))
        k = _capi.key_index(int(global_keys[0])) - offset
        if 0 <= k < count:
            rec[0, _capi.REC_COST], rec[0, _capi.REC_VIOLATION], rec[0, _capi.REC_OWNER] = float(cost[k]), float(viol[k]), 1.0
            rec[0, _capi.REC_HEADER:_capi.REC_HEADER + 2 * n] = torch.from_numpy(U[k].ravel())
            rec[0, _capi.REC_HEADER + 2 * n:] = torch.from_numpy(X[k].ravel())
        return rec

    gkeys, redrawn = global_select_sampled(local_key.clone(), regenerate)
    gkeys2, summed = global_select(local_key.clone(), own_record)
    assert int(gkeys[0]) == int(gkeys2[0])
    # softmin variant: weights against the GLOBAL minimum, one all-gather of [2n + 2] float64 per rank
    cost_min = _capi.key_cost(int(gkeys[0]))
    finite = np.isfinite(cost)
    w = np.where(finite, np.exp(-(np.where(finite, cost, 0.0).astype(np.float64) - cost_min) / lam), 0.0)
    wsum = w.sum()
    mean = np.einsum("c,cij->ij", w, U.astype(np.float64)) / wsum if wsum > 0 else U.mean(axis=0)
    payload = softmin_payload(torch.tensor(mean[None], dtype=torch.float32), torch.tensor([wsum]), count)
    gathered = [torch.zeros_like(payload) for _ in range(world)]
    dist.all_gather(gathered, payload)
    soft_mean, soft_wsum = combine_softmin(gathered, n)
    np.save(os.path.join(out_dir, "c4_key_%d.npy" % rank), gkeys.numpy())
    np.save(os.path.join(out_dir, "c4_redrawn_%d.npy" % rank), redrawn.numpy())
    np.save(os.path.join(out_dir, "c4_summed_%d.npy" % rank), summed.numpy())
    np.save(os.path.join(out_dir, "c4_soft_%d.npy" % rank), soft_mean.numpy())
    np.save(os.path.join(out_dir, "c4_local_%d.npy" % rank), np.array([cost.min(), float(np.count_nonzero(viol == 0)), wsum]))
    np.save(os.path.join(out_dir, "c4_wmean_%d.npy" % rank), mean * wsum)
    dist.destroy_process_group()


def test_config4_as_eight_gloo_ranks(tmp_path):
    """BASELINE configs[3] in its literal partition - 8 ranks x 32 768 candidates x horizon 80 with global index offsets -
    rehearsed on the CPU over gloo, the oracle standing in for the kernels: one all-reduce(MIN) of the packed keys and the
    winner re-drawn on every rank; the owner-masked records summed; the softmin payloads all-gathered and combined in rank
    order.  Every rank ends with the same key, the same record (both protocols) and the same mean, and they are the
    unsharded answers.  What an 8-GPU node adds to this is RCCL in gloo's place (unmeasured on hardware: DESIGN.md section 7)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import acmpc_oracle as orc
    from acmpc_amd import _capi
    from test_support import make_problem
    world = 8
    mp.spawn(_worker_config4, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    load = lambda name, r: np.load(tmp_path / ("c4_%s_%d.npy" % (name, r)))   # noqa: E731
    for name in ("key", "redrawn", "summed", "soft"):
        for r in range(1, world):
            np.testing.assert_array_equal(load(name, 0), load(name, r), err_msg="%s differs on rank %d" % (name, r))
    key = int(load("key", 0)[0])
    local = np.stack([load("local", r) for r in range(world)])
    n = 79
    redrawn, summed = load("redrawn", 0)[0], load("summed", 0)[0]
    # the global minimum is the smallest local one; its owner is the rank whose slice holds the index
    assert _capi.key_cost(key) == np.float32(local[:, 0].min())
    owner = _capi.key_index(key) // 32768
    assert local[owner, 0] == local[:, 0].min()
    # both protocols hand every rank the same plan; the summed one also carries the global feasible count
    np.testing.assert_array_equal(redrawn[_capi.REC_HEADER:], summed[_capi.REC_HEADER:])
    assert redrawn[_capi.REC_COST] == summed[_capi.REC_COST] == np.float32(_capi.key_cost(key))
    assert summed[_capi.REC_NFEASIBLE] == local[:, 1].sum() and summed[_capi.REC_OWNER] == 1.0
    # the unsharded answer for the winner: re-drawn from its global index alone
    prob = make_problem(orc, "nordschleife", 80, 16, seed=4243)
    cfg = prob["cfg"]
    u_ref = np.stack([prob["table"][orc.ROW_V], prob["table"][orc.ROW_KAPPA]], axis=1)
    Uw = orc.sample_candidates(u_ref, u_ref, 1, _capi.key_index(key), 0, 0, 99, (2.0, 0.01), prob["u_lo"], prob["u_hi"])
    coef = orc.coefficients_spatial(prob["table"], prob["limits"].margin)
    c, v, Xw = orc.rollout_spatial(prob["x0"], coef, Uw, cfg["step_cost"], cfg["r_term"], cfg["final_cost"], prob["u_lo"],
                                   prob["u_hi"], 1e6, dtype=np.float32, return_states=True)
    assert c[0] == redrawn[_capi.REC_COST]
    np.testing.assert_array_equal(redrawn[_capi.REC_HEADER:_capi.REC_HEADER + 2 * n].reshape(n, 2), Uw[0])
    np.testing.assert_array_equal(redrawn[_capi.REC_HEADER + 2 * n:].reshape(n + 1, 3), Xw[0])
    # softmin: the combined mean is sum_r (w u)_r / sum_r w_r over all 262 144 candidates
    want = sum(load("wmean", r) for r in range(world)) / local[:, 2].sum()
    np.testing.assert_allclose(load("soft", 0)[0], want, rtol=2e-6, atol=1e-6)
