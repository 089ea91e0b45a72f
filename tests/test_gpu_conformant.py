"""`acmpc_set_option(ctx, "ACMPC_CONFORMANT_SYNC", "1")` (control config key `conformant_sync`): ONE switch for the forms of
the latency paths that stay inside the HSA memory model and HIP's barrier rule - every solve, round and batch as separate
launches ordered by the stream, rounds on one wave per workgroup, completion by hipStreamSynchronize - instead of the
gfx950-specific ones (csrc/acmpc_kernels.hip, top: values published between workgroups of one launch with relaxed
agent-scope atomics + s_waitcnt vmcnt(0), waves that end while their workgroup still meets at s_barrier, the tick's
completion flag).  The results are the same bits: one solve per call, the closed-loop tick in both rollout modes, a stream
of batches (the headline's step), and the drop-in controller."""
import copy

import numpy as np
import pytest

import acmpc_oracle as orc
from test_support import RACING, PlaceholderVehicle, engine_kwargs, make_problem

pytestmark = pytest.mark.gpu


def _pair(**kwargs):
    from acmpc_amd import Engine
    default, conformant = Engine(**kwargs), Engine(**kwargs)
    conformant.set_option("ACMPC_CONFORMANT_SYNC", "1")
    return default, conformant


@pytest.mark.parametrize("track,H,N,P,layout", [("monza", 50, 4096, 1, 1), ("spa", 50, 1000, 3, 0), ("nordschleife", 80, 2085, 2, 1)])
def test_one_solve_per_call(track, H, N, P, layout):
    n = H - 1
    problems = [make_problem(orc, track, H, N, seed=40 + p) for p in range(P)]
    a, b = _pair(**engine_kwargs(problems[0], 0, P, N, n))
    x0 = np.stack([p["x0"] for p in problems])
    U = np.stack([p["U"] for p in problems])
    if layout == 1:
        U = np.ascontiguousarray(U.transpose(0, 2, 3, 1))
    outs = []
    for eng in (a, b):
        eng.set_paths(np.stack([p["table"] for p in problems]))
        outs.append(eng.solve(x0, U, layout=layout))
        eng.close()
    for key in ("costs", "best_idx", "records"):
        np.testing.assert_array_equal(outs[0][key], outs[1][key], err_msg=key)
    cfg = problems[0]["cfg"]
    cost, _ = orc.rollout_spatial(problems[0]["x0"], a_coef(problems[0]), problems[0]["U"], cfg["step_cost"], cfg["r_term"],
                                  cfg["final_cost"], problems[0]["u_lo"], problems[0]["u_hi"], 1.0e6, dtype=np.float32)
    np.testing.assert_array_equal(outs[1]["costs"][0], cost)          # and they are the oracle's


def a_coef(problem):
    return orc.coefficients_spatial(problem["table"], problem["limits"].margin)


@pytest.mark.parametrize("mode,window", [(0, None), (1, None), (1, (2, 5))])
def test_the_closed_loop_tick(mode, window):
    """Four consecutive ticks (16 384 candidates x 2 rounds, the LQ plan and its box refinement in the last): record,
    tables, unpacked plan and status of the conformant forms equal the default forms' bit for bit."""
    from test_gpu_restated_solve import _tick
    H, N, rounds = 50, 16384, 2
    n = H - 1
    cfg = RACING["monza"]
    cons = dict(cfg["speed_profile_constraints"], v_max=28.0)
    lim = orc.vehicle_limits(2.65, 1.94, 0.30, cons["v_min"], cons["v_max"])
    lo, hi = orc.input_box(lim)
    engines = _pair(mode=mode, max_problems=1, max_candidates=N, max_steps=n, step_cost=cfg["step_cost"], r_term=cfg["r_term"],
                    final_cost=cfg["final_cost"], u_min=lo, u_max=hi, margin=lim.margin, wheelbase=lim.length, nn_window=window,
                    lq_candidate=2, w_bound=1.0e4)
    runs = []
    for eng in engines:
        outs, centre = [], None
        for j in range(4):
            y = np.linspace(0, 120, H)
            coords = np.stack([0.004 * (1.0 + 0.1 * j) * y ** 2, y, np.linspace(10, 6, H)], axis=1)
            t = _tick(H, cons, N, rounds, 0.1 * j, 300 + j)
            t.centre_is_reference = 1 if centre is None else 0
            out = eng.control_tick(t, coords, centre)
            assert out["info"][4] == 0 and out["info"][7] == 0
            centre = out["decision"][3 * (n + 1):].reshape(n, 2).astype(np.float32)
            outs.append(out)
        eng.close()
        runs.append(outs)
    for want, got in zip(*runs):
        for key in ("record", "table", "decision", "projected_control", "prediction", "cum_time", "coords"):
            np.testing.assert_array_equal(want[key], got[key], err_msg=key)
        np.testing.assert_array_equal(want["info"][:6], got["info"][:6])


@pytest.mark.parametrize("sampled", [True, False])
def test_a_stream_of_batches(sampled):
    """The headline's step - acmpc_solve_stream_device, batch k's argmin and records inside batch k + 1's launch by default -
    as rollout + finalize launches per batch under the switch: costs, keys and records equal."""
    import torch
    from acmpc_amd import _capi
    P, N, H, K = 16, 4096, 50, 4
    n = H - 1
    problems = [make_problem(orc, "monza", H, 4, seed=700 + p) for p in range(P)]
    dev = torch.device("cuda", 0)
    s = torch.cuda.current_stream().cuda_stream
    x0 = torch.tensor(np.stack([p["x0"] for p in problems]), device=dev)
    u_ref = torch.tensor(np.stack([np.stack([p["table"][orc.ROW_V], p["table"][orc.ROW_KAPPA]], axis=1) for p in problems]),
                         dtype=torch.float32, device=dev).contiguous()
    sigma = (2.0, 0.01)
    R = _capi.record_floats(n)
    results = []
    for eng in _pair(**engine_kwargs(problems[0], 0, P, N, n)):
        eng.set_paths(np.stack([p["table"] for p in problems]))
        Us = []
        for k in range(K):
            U = torch.empty(P, n, 2, N, device=dev)
            eng.sample_device(u_ref.data_ptr(), 2 * n, u_ref.data_ptr(), P, N, n, 1, 0, sigma, 50 + k, k, U.data_ptr(), s)
            Us.append(U)
        costs = [torch.zeros(P, N, device=dev) for _ in range(K)]
        keys = [torch.zeros(P, dtype=torch.int64, device=dev) for _ in range(K)]
        recs = [torch.zeros(P, R, device=dev) for _ in range(K)]
        for k in range(K):
            eng.solve_stream_device(x0.data_ptr(), Us[k].data_ptr(), u_ref.data_ptr() if sampled else 0, 2 * n, u_ref.data_ptr(), P,
                                    N, n, 1, sigma, 50 + k, k, costs[k].data_ptr(), keys[k].data_ptr(), recs[k].data_ptr(), s)
        eng.solve_stream_flush(s)
        torch.cuda.synchronize()
        results.append([(c.cpu().numpy(), q.cpu().numpy(), r.cpu().numpy()) for c, q, r in zip(costs, keys, recs)])
        eng.close()
    for k, (want, got) in enumerate(zip(*results)):
        for name, w, g in zip(("costs", "keys", "records"), want, got):
            np.testing.assert_array_equal(w, g, err_msg="%s of batch %d" % (name, k))


def test_the_drop_in_controller_takes_the_switch_from_its_config():
    from acmpc_amd import workloads
    from acmpc_amd.mpc import build_mpc
    plans = []
    for conformant in (False, True):
        cfg = copy.deepcopy(RACING["silverstone"])
        cfg["conformant_sync"] = conformant
        mpc = build_mpc(cfg, PlaceholderVehicle())
        circuit = workloads.synthetic_track("silverstone")
        mpc.bind_map(circuit)
        seen = []
        for i in range(12):
            mpc.speed_profile_constraints["v_max"] = 32.0
            mpc.get_control_at(map_index=(40 * i) % len(circuit["centre"]), lateral_offset=0.3, offset=0.3)
            assert mpc.infeasibility_counter == 0
            seen.append((mpc.projected_control.copy(), mpc.cum_time.copy(), mpc.current_prediction.copy()))
        plans.append(seen)
    for want, got in zip(*plans):
        for w, g in zip(want, got):
            np.testing.assert_array_equal(w, g)
