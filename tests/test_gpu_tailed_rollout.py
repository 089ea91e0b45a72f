"""The batched solve's finalize in its forms (round 4): a wavefront per problem (finalize_kernel; ACMPC_NO_GROUP_FINALIZE=1
keeps it at any problem count); from 256 problems up sixteen lanes per problem, four problems per wavefront
(finalize_groups_kernel, the default there); inside the rollout's launch (rollout_tailed_kernel: the last workgroup of every
problem takes the argmin and re-rolls the winner while other problems are still streaming); and - a stream of batches -
inside the NEXT batch's rollout launch (rollout_chained_kernel).  The records must be
the two-launch form's (rollout_kernel + finalize_kernel, the default: it is the faster of the two on the headline's batch,
csrc/acmpc_capi.hip solve_batched; ACMPC_TAILED_ROLLOUT=1 selects the one launch) bit for bit, and the oracle's."""
import numpy as np
import pytest

import acmpc_oracle as orc
from test_support import engine_kwargs, make_problem

pytestmark = pytest.mark.gpu


def _solve(monkeypatch, switch, problems, P, N, n, sampled):
    import torch
    from acmpc_amd import Engine, _capi
    for name in ("ACMPC_TAILED_ROLLOUT", "ACMPC_NO_GROUP_FINALIZE", "ACMPC_FINALIZE_WAVES"):
        monkeypatch.delenv(name, raising=False)
    for name in (switch or "").split():
        monkeypatch.setenv(name, "1")
    dev = torch.device("cuda", 0)
    eng = Engine(**engine_kwargs(problems[0], 0, P, N, n))      # (the switch is read when the handle is created)
    eng.set_paths(np.stack([p["table"] for p in problems]))
    s = torch.cuda.current_stream().cuda_stream
    x0 = torch.tensor(np.stack([p["x0"] for p in problems]), device=dev)
    u_ref = torch.tensor(np.stack([np.stack([p["table"][orc.ROW_V], p["table"][orc.ROW_KAPPA]], axis=1) for p in problems]),
                         dtype=torch.float32, device=dev).contiguous()
    U = torch.empty(P, n, 2, N, device=dev)
    sigma, seed, rnd = (2.0, 0.01), 4321, 1
    eng.sample_device(u_ref.data_ptr(), 2 * n, u_ref.data_ptr(), P, N, n, 1, 0, sigma, seed, rnd, U.data_ptr(), s)
    costs = torch.empty(P, N, device=dev)
    keys = torch.empty(P, dtype=torch.int64, device=dev)
    rec = torch.empty(P, _capi.record_floats(n), device=dev)
    outs = []
    for _ in range(3):   # repeated launches: the tickets are left as they were found
        rec.zero_()
        if sampled:
            eng.solve_sampled_device(x0.data_ptr(), U.data_ptr(), u_ref.data_ptr(), 2 * n, u_ref.data_ptr(), P, N, n, 1,
                                     sigma, seed, rnd, costs.data_ptr(), keys.data_ptr(), rec.data_ptr(), s)
        else:
            eng.solve_device(x0.data_ptr(), U.data_ptr(), P, N, n, 1, costs.data_ptr(), keys.data_ptr(), rec.data_ptr(), s)
        torch.cuda.synchronize()
        outs.append((rec.cpu().numpy().copy(), keys.cpu().numpy().copy(), costs.cpu().numpy().copy()))
    eng.close()
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            np.testing.assert_array_equal(a, b)
    return outs[0] + (U.cpu().numpy().transpose(0, 3, 1, 2).copy(),)


@pytest.mark.parametrize("P,N,H", [(64, 4096, 50),      # 4 workgroups of 1 024 candidates per problem (the headline's shape)
                                   (301, 2048, 20),     # two candidates per lane: 4 workgroups of 512; a last wavefront of
                                                        # the finalize with one problem for its four quarters
                                   (3, 65536 + 1024, 50),   # beyond the one-launch solve's 1 024 workgroups of 64
                                   (520, 1000, 33),     # a ragged last workgroup, one candidate per lane
                                   (260, 1024, 80),     # 79 steps: more than one pass of a problem's lanes in the finalize
                                   (257, 512, 130)])    # 129 steps: three passes of sixteen lanes x four, three of 64
@pytest.mark.parametrize("sampled", [True, False])
def test_tailed_rollout_equals_two_launches_and_the_oracle(monkeypatch, P, N, H, sampled):
    from acmpc_amd import _capi
    n = H - 1
    problems = [make_problem(orc, "monza", H, 4, seed=900 + p % 7) for p in range(P)]
    one = _solve(monkeypatch, "ACMPC_TAILED_ROLLOUT", problems, P, N, n, sampled)
    two = _solve(monkeypatch, None, problems, P, N, n, sampled)     # (the default: two launches; from 256 problems sixteen
    waves = _solve(monkeypatch, "ACMPC_NO_GROUP_FINALIZE", problems, P, N, n, sampled)                      # lanes per problem)
    four = _solve(monkeypatch, "ACMPC_FINALIZE_WAVES", problems, P, N, n, sampled)     # (a wavefront per problem, four per workgroup)
    for other in (two, waves, four):
        for a, b, what in zip(one, other, ("records", "keys", "costs", "controls")):
            np.testing.assert_array_equal(a, b, err_msg=what)
    rec, keys, costs, U = one
    for p in (0, P // 2, P - 1):
        prob, cfg = problems[p], problems[p]["cfg"]
        coef = orc.coefficients_spatial(prob["table"], prob["limits"].margin)
        cost, viol, X = orc.rollout_spatial(prob["x0"], coef, U[p], cfg["step_cost"], cfg["r_term"], cfg["final_cost"],
                                            prob["u_lo"], prob["u_hi"], 1.0e6, dtype=np.float32, return_states=True)
        np.testing.assert_array_equal(costs[p], cost)
        best, _ = orc.pick_best(cost)
        assert _capi.key_index(int(keys[p])) == best
        out = _capi.split_record(rec[p], n)
        assert out["cost"] == cost[best] and out["violation"] == viol[best]
        assert int(out["n_feasible"]) == int(np.count_nonzero(viol == 0))
        np.testing.assert_array_equal(out["u"], U[p, best])
        np.testing.assert_array_equal(out["x"], X[best])


@pytest.mark.parametrize("layout", [0, 1])
def test_two_shards_of_many_problems_in_every_finalize_form(monkeypatch, layout):
    """The sharded protocol's finalize (global keys in, the winner's record written by the shard that owns it, a blank
    record with the shard's feasible count by the other) at a problem count that takes the many-problem kernels: all
    forms write the same records, and the two shards' records add up to the unsharded solve's."""
    import torch
    from acmpc_amd import Engine, _capi
    P, H, total = 258, 50, 768
    n = H - 1
    distinct = [make_problem(orc, "silverstone", H, total, seed=190 + p) for p in range(6)]
    problems = [distinct[p % 6] for p in range(P)]
    U_full = np.stack([p["U"] for p in problems])                       # [P,N,n,2]
    for p in range(P):                                                   # (six distinct problems: rotating each one's
        U_full[p] = np.roll(U_full[p], 97 * p, axis=0)                   # candidates lands the winners in both shards)
    dev = torch.device("cuda", 0)
    x0 = torch.tensor(np.stack([p["x0"] for p in problems]), device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    R = _capi.record_floats(n)
    shards = [(0, 500), (500, 268)]
    results = {}
    for switch in ("", "ACMPC_NO_GROUP_FINALIZE", "ACMPC_FINALIZE_WAVES"):
        monkeypatch.delenv("ACMPC_NO_GROUP_FINALIZE", raising=False)
        monkeypatch.delenv("ACMPC_FINALIZE_WAVES", raising=False)
        for name in switch.split():
            monkeypatch.setenv(name, "1")
        full = Engine(**engine_kwargs(problems[0], 0, P, total, n))
        full.set_paths(np.stack([p["table"] for p in problems]))
        want = full.solve(x0.cpu().numpy(), U_full if layout == 0 else np.ascontiguousarray(U_full.transpose(0, 2, 3, 1)),
                          layout=layout)
        full.close()
        keys, recs, engines = [], [], []
        for offset, count in shards:
            eng = Engine(**engine_kwargs(problems[0], 0, P, count, n))
            eng.set_paths(np.stack([p["table"] for p in problems]))
            U = U_full[:, offset:offset + count]
            U = torch.tensor(U if layout == 0 else np.ascontiguousarray(U.transpose(0, 2, 3, 1)), device=dev)
            k = torch.empty(P, dtype=torch.int64, device=dev)
            c = torch.empty(P, count, dtype=torch.float32, device=dev)
            eng.rollout_device(x0.data_ptr(), U.data_ptr(), P, count, n, layout, offset, c.data_ptr(), k.data_ptr(), stream)
            engines.append((eng, U, offset, count))
            keys.append(k)
        gkeys = torch.minimum(keys[0], keys[1])
        for eng, U, offset, count in engines:
            r = torch.empty(P, R, dtype=torch.float32, device=dev)
            eng.finalize_device(gkeys.data_ptr(), x0.data_ptr(), U.data_ptr(), P, count, n, layout, offset, r.data_ptr(), stream)
            recs.append(r)
        torch.cuda.synchronize()
        owners = torch.stack([r[:, _capi.REC_OWNER] for r in recs]).cpu().numpy()
        assert (owners.sum(axis=0) == 1).all() and owners[0].sum() > 0 and owners[1].sum() > 0
        np.testing.assert_array_equal((recs[0] + recs[1]).cpu().numpy(), want["records"])
        results[switch] = [r.cpu().numpy() for r in recs] + [want["records"]]
        for eng, *_ in engines:
            eng.close()
    for switch, got in results.items():
        for a, b in zip(got, results[""]):
            np.testing.assert_array_equal(a, b, err_msg=switch)


@pytest.mark.parametrize("P,N,H,sampled", [(64, 4096, 50, True),      # the headline's shape: chained (4 finalize rows of 4 workgroups)
                                           (301, 2048, 20, True),     # a ragged last finalize workgroup and a ragged last row
                                           (520, 1000, 33, False),    # winners read from the previous batch's matrix
                                           (3, 4096, 50, True),       # fewer problems than a finalize workgroup takes
                                           (40, 2048, 80, True),      # horizon 80: the finalize image would cost the rollout
                                                                      # its occupancy - a launch of its own in front
                                           (1, 64, 3, True),          # one workgroup in all, two steps
                                           (5, 1000, 59, True),       # 58 steps: the longest horizon the one launch takes
                                           (7, 4096, 61, False)])     # 60 steps: the first it does not
@pytest.mark.parametrize("chained", [True, False])
def test_a_stream_of_batches_equals_one_call_per_batch(monkeypatch, P, N, H, sampled, chained):
    """acmpc_solve_stream_device: batch k's argmin and records computed inside batch k + 1's rollout launch (or by the
    flush) are the bits acmpc_solve_sampled_device / acmpc_solve_device return for batch k on its own."""
    import torch
    from acmpc_amd import Engine, _capi
    monkeypatch.delenv("ACMPC_NO_CHAINED_STREAM", raising=False)
    if not chained:
        monkeypatch.setenv("ACMPC_NO_CHAINED_STREAM", "1")
    n = H - 1
    problems = [make_problem(orc, "monza", H, 4, seed=900 + p % 7) for p in range(P)]
    dev = torch.device("cuda", 0)
    eng = Engine(**engine_kwargs(problems[0], 0, P, N, n))
    eng.set_paths(np.stack([p["table"] for p in problems]))
    s = torch.cuda.current_stream().cuda_stream
    x0 = torch.tensor(np.stack([p["x0"] for p in problems]), device=dev)
    u_ref = torch.tensor(np.stack([np.stack([p["table"][orc.ROW_V], p["table"][orc.ROW_KAPPA]], axis=1) for p in problems]),
                         dtype=torch.float32, device=dev).contiguous()
    sigma, K = (2.0, 0.01), 5
    seeds = [777 + 13 * k for k in range(K)]
    Us = []
    for k in range(K):
        U = torch.empty(P, n, 2, N, device=dev)
        eng.sample_device(u_ref.data_ptr(), 2 * n, u_ref.data_ptr(), P, N, n, 1, 0, sigma, seeds[k], k, U.data_ptr(), s)
        Us.append(U)
    R = _capi.record_floats(n)

    def buffers():
        return ([torch.zeros(P, N, device=dev) for _ in range(K)], [torch.zeros(P, dtype=torch.int64, device=dev) for _ in range(K)],
                [torch.zeros(P, R, device=dev) for _ in range(K)])

    costs, keys, recs = buffers()
    for k in range(K):
        if sampled:
            eng.solve_sampled_device(x0.data_ptr(), Us[k].data_ptr(), u_ref.data_ptr(), 2 * n, u_ref.data_ptr(), P, N, n, 1, sigma,
                                     seeds[k], k, costs[k].data_ptr(), keys[k].data_ptr(), recs[k].data_ptr(), s)
        else:
            eng.solve_device(x0.data_ptr(), Us[k].data_ptr(), P, N, n, 1, costs[k].data_ptr(), keys[k].data_ptr(), recs[k].data_ptr(), s)
    torch.cuda.synchronize()
    costs_s, keys_s, recs_s = buffers()
    for repeat in range(2):   # (a second stream on the same handle starts from a clean slate)
        for k in range(K):
            eng.solve_stream_device(x0.data_ptr(), Us[k].data_ptr(), u_ref.data_ptr() if sampled else 0, 2 * n, u_ref.data_ptr(),
                                    P, N, n, 1, sigma, seeds[k], k, costs_s[k].data_ptr(), keys_s[k].data_ptr(),
                                    recs_s[k].data_ptr(), s)
            if k == 1:   # (drawing candidates is allowed while a batch is pending: what a real stream does)
                eng.sample_device(u_ref.data_ptr(), 2 * n, u_ref.data_ptr(), P, N, n, 1, 0, sigma, seeds[k + 1], k + 1,
                                  Us[k + 1].data_ptr(), s)
            if k == 2:   # nothing else runs on the handle while a batch is pending
                with pytest.raises(_capi.EngineError, match="pending"):
                    eng.solve_device(x0.data_ptr(), Us[k].data_ptr(), P, N, n, 1, costs[k].data_ptr(), keys[k].data_ptr(),
                                     recs[k].data_ptr(), s)
        eng.solve_stream_flush(s)
        eng.solve_stream_flush(s)   # (nothing pending: a no-op)
        torch.cuda.synchronize()
        for k in range(K):
            np.testing.assert_array_equal(costs_s[k].cpu().numpy(), costs[k].cpu().numpy(), err_msg="costs of batch %d" % k)
            np.testing.assert_array_equal(keys_s[k].cpu().numpy(), keys[k].cpu().numpy(), err_msg="keys of batch %d" % k)
            np.testing.assert_array_equal(recs_s[k].cpu().numpy(), recs[k].cpu().numpy(), err_msg="records of batch %d" % k)
            recs_s[k].zero_()
            keys_s[k].zero_()
    eng.close()


def test_new_tables_in_the_middle_of_a_stream(monkeypatch):
    """acmpc_set_paths between two batches of a stream: the pending finalize still reads the tables its batch was rolled
    with (the upload goes behind it)."""
    import torch
    from acmpc_amd import Engine, _capi
    monkeypatch.delenv("ACMPC_NO_CHAINED_STREAM", raising=False)
    P, N, H = 64, 2048, 50
    n = H - 1
    sets = [[make_problem(orc, track, H, 4, seed=300 + p % 5) for p in range(P)] for track in ("monza", "silverstone")]
    dev = torch.device("cuda", 0)
    eng = Engine(**engine_kwargs(sets[0][0], 0, P, N, n))
    s = torch.cuda.current_stream().cuda_stream
    R = _capi.record_floats(n)
    sigma = (2.0, 0.01)
    want, got, keep = [], [], []
    for which in (0, 1, 0):
        problems = sets[which]
        x0 = torch.tensor(np.stack([p["x0"] for p in problems]), device=dev)
        u_ref = torch.tensor(np.stack([np.stack([p["table"][orc.ROW_V], p["table"][orc.ROW_KAPPA]], axis=1) for p in problems]),
                             dtype=torch.float32, device=dev).contiguous()
        U = torch.empty(P, n, 2, N, device=dev)
        keep.append((x0, u_ref, U, np.stack([p["table"] for p in problems])))
    for x0, u_ref, U, tables in keep:
        eng.set_paths(tables)
        eng.sample_device(u_ref.data_ptr(), 2 * n, u_ref.data_ptr(), P, N, n, 1, 0, sigma, 5, 0, U.data_ptr(), s)
        rec = torch.zeros(P, R, device=dev)
        eng.solve_sampled_device(x0.data_ptr(), U.data_ptr(), u_ref.data_ptr(), 2 * n, u_ref.data_ptr(), P, N, n, 1, sigma, 5, 0, 0, 0,
                                 rec.data_ptr(), s)
        want.append(rec)
    torch.cuda.synchronize()
    for x0, u_ref, U, tables in keep:
        eng.set_paths(tables)
        rec = torch.zeros(P, R, device=dev)
        eng.solve_stream_device(x0.data_ptr(), U.data_ptr(), u_ref.data_ptr(), 2 * n, u_ref.data_ptr(), P, N, n, 1, sigma, 5, 0, 0, 0,
                                rec.data_ptr(), s)
        got.append(rec)
    eng.solve_stream_flush(s)
    torch.cuda.synchronize()
    for a, b in zip(got, want):
        np.testing.assert_array_equal(a.cpu().numpy(), b.cpu().numpy())
    assert not np.array_equal(want[0].cpu().numpy(), want[1].cpu().numpy())
    eng.close()


@pytest.mark.parametrize("switch", ["", "ACMPC_NO_GROUP_FINALIZE", "ACMPC_FINALIZE_WAVES"])
def test_records_from_reduced_keys_equal_the_one_call_solve(monkeypatch, switch):
    """The multi-rank step's last call - acmpc_finalize_sampled_device on the keys an all-reduce(MIN) left - at the
    headline's problem count per launch shape: the records of the one-call solve, from the keys alone."""
    import torch
    from acmpc_amd import Engine, _capi
    monkeypatch.delenv("ACMPC_NO_GROUP_FINALIZE", raising=False)
    monkeypatch.delenv("ACMPC_FINALIZE_WAVES", raising=False)
    for name in switch.split():
        monkeypatch.setenv(name, "1")
    P, N, H = 333, 2048, 50
    n = H - 1
    problems = [make_problem(orc, "monza", H, 4, seed=900 + p % 7) for p in range(P)]
    dev = torch.device("cuda", 0)
    eng = Engine(**engine_kwargs(problems[0], 0, P, N, n))
    eng.set_paths(np.stack([p["table"] for p in problems]))
    s = torch.cuda.current_stream().cuda_stream
    x0 = torch.tensor(np.stack([p["x0"] for p in problems]), device=dev)
    u_ref = torch.tensor(np.stack([np.stack([p["table"][orc.ROW_V], p["table"][orc.ROW_KAPPA]], axis=1) for p in problems]),
                         dtype=torch.float32, device=dev).contiguous()
    centre = (u_ref + torch.tensor([-0.4, 0.001], device=dev)).contiguous()
    sigma, seed, rnd = (2.0, 0.01), 99, 3
    U = torch.empty(P, n, 2, N, device=dev)
    eng.sample_device(centre.data_ptr(), 2 * n, u_ref.data_ptr(), P, N, n, 1, 0, sigma, seed, rnd, U.data_ptr(), s)
    keys = torch.empty(P, dtype=torch.int64, device=dev)
    one = torch.zeros(P, _capi.record_floats(n), device=dev)
    eng.solve_sampled_device(x0.data_ptr(), U.data_ptr(), centre.data_ptr(), 2 * n, u_ref.data_ptr(), P, N, n, 1, sigma, seed, rnd, 0,
                             keys.data_ptr(), one.data_ptr(), s)
    again = torch.zeros_like(one)
    eng.finalize_sampled_device(keys.data_ptr(), x0.data_ptr(), centre.data_ptr(), 2 * n, u_ref.data_ptr(), P, N, n, sigma, seed, rnd,
                                again.data_ptr(), s)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(again.cpu().numpy(), one.cpu().numpy())
    index = keys.cpu().numpy() & 0xffffffff
    assert (index == 1).sum() < P and len(np.unique(index)) > 10     # (not all the reference controls: real winners)
    eng.close()


def test_a_pending_batch_keeps_the_knot_table_of_its_own_horizon():
    """(advisor, round 4) The pending finalize of a stream re-draws its winners with the sampler's knot table, which lives
    in ONE device buffer rewritten in place when the horizon changes.  With a batch pending, acmpc_sample_device for another
    horizon is refused, and acmpc_solve_stream_device for another horizon runs the pending finalize FIRST: the first batch's
    records are the ones a plain solve of it gives."""
    import torch
    from acmpc_amd import Engine, _capi
    P, N = 6, 2048
    dev = torch.device("cuda", 0)
    s = torch.cuda.current_stream().cuda_stream
    sigma = (2.0, 0.01)

    def inputs(H, seed):
        problems = [make_problem(orc, "monza", H, 4, seed=seed + p) for p in range(P)]
        tables = np.stack([p["table"] for p in problems])
        x0 = torch.tensor(np.stack([p["x0"] for p in problems]), device=dev)
        u_ref = torch.tensor(np.stack([np.stack([p["table"][orc.ROW_V], p["table"][orc.ROW_KAPPA]], axis=1) for p in problems]),
                             dtype=torch.float32, device=dev).contiguous()
        return problems, tables, x0, u_ref

    long_, short = inputs(50, 300), inputs(30, 400)
    eng = Engine(**engine_kwargs(long_[0][0], 0, P, N, 49))
    want = {}
    for name, (problems, tables, x0, u_ref), n in (("long", long_, 49), ("short", short, 29)):
        eng.set_paths(tables)
        U = torch.empty(P, n, 2, N, device=dev)
        eng.sample_device(u_ref.data_ptr(), 2 * n, u_ref.data_ptr(), P, N, n, 1, 0, sigma, 11, 0, U.data_ptr(), s)
        rec = torch.zeros(P, _capi.record_floats(n), device=dev)
        eng.solve_sampled_device(x0.data_ptr(), U.data_ptr(), u_ref.data_ptr(), 2 * n, u_ref.data_ptr(), P, N, n, 1, sigma, 11, 0, 0, 0,
                                 rec.data_ptr(), s)
        torch.cuda.synchronize()
        want[name] = (U, rec.cpu().numpy().copy())
    # the stream: the long-horizon batch pending, then everything about the short one
    eng.set_paths(long_[1])
    rec_long = torch.zeros(P, _capi.record_floats(49), device=dev)
    eng.solve_stream_device(long_[2].data_ptr(), want["long"][0].data_ptr(), long_[3].data_ptr(), 98, long_[3].data_ptr(), P, N, 49, 1,
                            sigma, 11, 0, 0, 0, rec_long.data_ptr(), s)
    eng.set_paths(short[1])
    scratch = torch.empty(P, 29, 2, N, device=dev)
    with pytest.raises(_capi.EngineError, match="pending"):
        eng.sample_device(short[3].data_ptr(), 58, short[3].data_ptr(), P, N, 29, 1, 0, sigma, 11, 0, scratch.data_ptr(), s)
    rec_short = torch.zeros(P, _capi.record_floats(29), device=dev)
    eng.solve_stream_device(short[2].data_ptr(), want["short"][0].data_ptr(), short[3].data_ptr(), 58, short[3].data_ptr(), P, N, 29, 1,
                            sigma, 11, 0, 0, 0, rec_short.data_ptr(), s)
    eng.solve_stream_flush(s)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(rec_long.cpu().numpy(), want["long"][1])
    np.testing.assert_array_equal(rec_short.cpu().numpy(), want["short"][1])
    eng.close()
