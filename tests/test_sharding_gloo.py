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


@pytest.mark.parametrize("total", [256, 257])
def test_two_rank_selection_equals_single_process(tmp_path, total):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import acmpc_oracle as orc
    from acmpc_amd import _capi
    from test_support import make_problem

    world = 2
    mp.spawn(_worker, args=(world, _free_port(), total, str(tmp_path)), nprocs=world, join=True)
    keys = [np.load(tmp_path / ("keys_%d.npy" % r)) for r in range(world)]
    recs = [np.load(tmp_path / ("records_%d.npy" % r)) for r in range(world)]
    np.testing.assert_array_equal(keys[0], keys[1])
    np.testing.assert_array_equal(recs[0], recs[1])
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


def test_single_collective_protocol_for_counter_based_candidates(tmp_path):
    """Two gloo ranks, ONE all-reduce(MIN): both ranks end with identical complete records and the global argmin."""
    from acmpc_amd import _capi
    world, total = 2, 301
    mp.spawn(_worker_sampled, args=(world, _free_port(), total, str(tmp_path)), nprocs=world, join=True)
    keys = [np.load(tmp_path / ("skeys_%d.npy" % r)) for r in range(world)]
    recs = [np.load(tmp_path / ("srecords_%d.npy" % r)) for r in range(world)]
    np.testing.assert_array_equal(keys[0], keys[1])
    np.testing.assert_array_equal(recs[0], recs[1])
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
