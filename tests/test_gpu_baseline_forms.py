"""BASELINE.json configs[3] in the form it is stated: Nordschleife, 262 144 candidates x horizon 80, sharded over 8
GPUs with one all-reduce(MIN) of the per-shard best (cost, index) keys.  One card cannot run eight RCCL ranks, so the
eight shards run one after the other on it with their global index offsets and the collective is played by
`torch.minimum` over the eight key vectors - everything else (kernels, keys, both finalize protocols) is the real
path, at full size, and must equal the unsharded solve and the C oracle bit for bit.  The two-rank rehearsal runs the
same problem through `torch.distributed` (gloo carrying the keys, both ranks on this card)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import acmpc_oracle as orc
from test_support import engine_kwargs, full_size_controls, make_problem

pytestmark = pytest.mark.gpu

SHARDS, PER_SHARD, H = 8, 32768, 80
N, n = SHARDS * PER_SHARD, H - 1


def test_config4_arbitrary_controls_in_eight_shards():
    """Protocol for caller-supplied control matrices: rollout per shard -> MIN of the keys -> finalize on every shard
    (the owner writes the record, the others zeros + their feasible count) -> SUM of the records."""
    import c_oracle
    import torch
    from acmpc_amd import Engine, _capi
    prob = make_problem(orc, "nordschleife", H, 16, seed=4242)
    U = full_size_controls(orc, prob, N, n)                       # [N, n, 2]
    U_sm = np.ascontiguousarray(U.transpose(1, 2, 0))             # [n, 2, N]
    cfg = prob["cfg"]
    full = Engine(**engine_kwargs(prob, 0, 1, N, n))
    full.set_paths(prob["table"])
    want = full.solve(prob["x0"][None], U_sm[None], layout=1)
    w = c_oracle.make_weights(cfg["step_cost"], cfg["r_term"], cfg["final_cost"], prob["u_lo"], prob["u_hi"], 1.0e6)
    oracle_cost, oracle_viol = c_oracle.rollout(0, prob["x0"], full.coefficients(0), U, 0, w)
    np.testing.assert_array_equal(want["costs"][0], oracle_cost)
    assert want["best_idx"][0] == c_oracle.argmin(oracle_cost)

    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    x0 = torch.tensor(prob["x0"][None], device=dev)
    R = _capi.record_floats(n)
    engines, keys, costs, slices = [], [], [], []
    for r in range(SHARDS):
        eng = Engine(**engine_kwargs(prob, 0, 1, PER_SHARD, n))
        eng.set_paths(prob["table"])
        d_U = torch.tensor(np.ascontiguousarray(U_sm[:, :, r * PER_SHARD:(r + 1) * PER_SHARD])[None], device=dev)
        k = torch.empty(1, dtype=torch.int64, device=dev)
        c = torch.empty(1, PER_SHARD, dtype=torch.float32, device=dev)
        eng.rollout_device(x0.data_ptr(), d_U.data_ptr(), 1, PER_SHARD, n, 1, r * PER_SHARD, c.data_ptr(), k.data_ptr(),
                           stream)
        engines.append(eng), keys.append(k), costs.append(c), slices.append(d_U)
    gkeys = torch.stack(keys).min(dim=0).values                    # what the all-reduce(MIN) over xGMI computes
    records = []
    for r, eng in enumerate(engines):
        rec = torch.empty(1, R, dtype=torch.float32, device=dev)
        eng.finalize_device(gkeys.data_ptr(), x0.data_ptr(), slices[r].data_ptr(), 1, PER_SHARD, n, 1, r * PER_SHARD,
                            rec.data_ptr(), stream)
        records.append(rec)
    torch.cuda.synchronize()
    owners = torch.stack([rec[0, _capi.REC_OWNER] for rec in records]).cpu().numpy()
    assert owners.sum() == 1 and owners[int(want["best_idx"][0]) // PER_SHARD] == 1
    np.testing.assert_array_equal(torch.stack(records).sum(dim=0).cpu().numpy(), want["records"])   # all-reduce(SUM)
    np.testing.assert_array_equal(torch.cat(costs, dim=1).cpu().numpy()[0], oracle_cost)
    assert _capi.key_index(int(gkeys[0])) == want["best_idx"][0] and _capi.key_cost(int(gkeys[0])) == want["cost"][0]
    assert want["n_feasible"][0] == np.count_nonzero(oracle_viol == 0)


@pytest.mark.parametrize("mode", [0, 1])
def test_config4_counter_based_candidates_in_eight_shards(mode):
    """The single-collective protocol: every shard draws its slice of the global candidate indices (Philox counters),
    rolls it out, ONE MIN over the keys, and every shard re-draws the winner from the index in the key: eight identical
    complete records, equal to the unsharded solve's."""
    import c_oracle
    import torch
    from acmpc_amd import Engine, _capi
    prob = make_problem(orc, "nordschleife", H, 16, seed=4243)
    cfg = prob["cfg"]
    window = None if mode == 0 else (2, 5)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    x0 = torch.tensor((prob["x0"] if mode == 0 else prob["pose0"])[None], device=dev)
    u_ref = torch.tensor(np.stack([prob["table"][orc.ROW_V], prob["table"][orc.ROW_KAPPA]], axis=1)[None],
                         dtype=torch.float32, device=dev).contiguous()
    sigma, seed, R = (2.0, 0.01), 99, _capi.record_floats(n)

    full = Engine(**engine_kwargs(prob, mode, 1, N, n, nn_window=window))
    full.set_paths(prob["table"])
    U_full = torch.empty(1, n, 2, N, device=dev)
    full.sample_device(u_ref.data_ptr(), 2 * n, u_ref.data_ptr(), 1, N, n, 1, 0, sigma, seed, 0, U_full.data_ptr(), stream)
    cost_full = torch.empty(1, N, device=dev)
    rec_full = torch.empty(1, R, device=dev)
    key_full = torch.empty(1, dtype=torch.int64, device=dev)
    full.solve_device(x0.data_ptr(), U_full.data_ptr(), 1, N, n, 1, cost_full.data_ptr(), key_full.data_ptr(),
                      rec_full.data_ptr(), stream)
    torch.cuda.synchronize()
    w = c_oracle.make_weights(cfg["step_cost"], cfg["r_term"], cfg["final_cost"], prob["u_lo"], prob["u_hi"], 1.0e6,
                              nn_window=window)
    oracle_cost, _ = c_oracle.rollout(mode, x0.cpu().numpy()[0], full.coefficients(0), U_full.cpu().numpy()[0], 1, w)
    np.testing.assert_array_equal(cost_full.cpu().numpy()[0], oracle_cost)

    engines, keys, costs = [], [], []
    for r in range(SHARDS):
        eng = Engine(**engine_kwargs(prob, mode, 1, PER_SHARD, n, nn_window=window))
        eng.set_paths(prob["table"])
        d_U = torch.empty(1, n, 2, PER_SHARD, device=dev)
        eng.sample_device(u_ref.data_ptr(), 2 * n, u_ref.data_ptr(), 1, PER_SHARD, n, 1, r * PER_SHARD, sigma, seed, 0,
                          d_U.data_ptr(), stream)
        torch.cuda.synchronize()
        assert torch.equal(d_U, U_full[:, :, :, r * PER_SHARD:(r + 1) * PER_SHARD]), "a shard draws its own indices"
        k = torch.empty(1, dtype=torch.int64, device=dev)
        c = torch.empty(1, PER_SHARD, dtype=torch.float32, device=dev)
        eng.rollout_device(x0.data_ptr(), d_U.data_ptr(), 1, PER_SHARD, n, 1, r * PER_SHARD, c.data_ptr(), k.data_ptr(),
                           stream)
        engines.append(eng), keys.append(k), costs.append(c)
    gkeys = torch.stack(keys).min(dim=0).values
    assert torch.equal(gkeys, key_full)
    want = rec_full.cpu().numpy().copy()
    want[:, _capi.REC_NFEASIBLE] = 0
    for r, eng in enumerate(engines):
        rec = torch.empty(1, R, dtype=torch.float32, device=dev)
        eng.finalize_sampled_device(gkeys.data_ptr(), x0.data_ptr(), u_ref.data_ptr(), 2 * n, u_ref.data_ptr(), 1,
                                    PER_SHARD, n, sigma, seed, 0, rec.data_ptr(), stream)
        torch.cuda.synchronize()
        got = rec.cpu().numpy().copy()
        got[:, _capi.REC_NFEASIBLE] = 0                              # each shard's own feasible count
        np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(torch.cat(costs, dim=1).cpu().numpy()[0], oracle_cost)


def test_config4_two_ranks_of_131072_over_torch_distributed():
    """Two processes (one per would-be GPU, both on this card), 131 072 candidates each at horizon 80, the keys
    all-reduced by torch.distributed (gloo here, RCCL on a multi-GPU node): tests/config4_ranks.py."""
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "config4_ranks.py")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    proc = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                           "--master-addr", "127.0.0.1", "--master-port", str(port), script],
                          capture_output=True, text=True, timeout=600, env=env)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-4000:]
    assert "config 4 in two ranks ok" in proc.stdout
