"""One whole solve of the sampling controller, end to end against its restatement (round 4).

`oracle.optimize_restated` is sample -> roll -> argmin -> recentre, round after round, with the sampler's normals
specified bit for bit (`box_muller_spec`) and the LQ plan (`lq_plan`, csrc/acmpc_lq.h) as the last round's candidate 2.
acmpc_optimize and acmpc_control_tick must return exactly its record at the closed-loop shape (1 x 16 384 x H 50, modes S
and T) - the product's real path, what `get_control` runs in place of the reference's solver call
(/root/reference/src/acmpc/control/spatial_mpc.py:185-217) - and both must reproduce the records frozen in
tests/golden/composition.npz."""
import os

import numpy as np
import pytest

import acmpc_oracle as orc
from test_support import RACING, engine_kwargs, make_problem

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "composition.npz")


def _record_equals(rec, want, n):
    assert rec["cost"] == want["cost"] and rec["violation"] == want["violation"], (rec["cost"], want["cost"])
    assert int(rec["n_feasible"]) == want["n_feasible"]
    np.testing.assert_array_equal(rec["u"], want["u"])
    np.testing.assert_array_equal(rec["x"], want["x"])


@pytest.mark.parametrize("mode,window,N,rounds,lq", [(0, None, 16384, 2, True), (0, None, 16384, 6, False), (0, None, 1000, 3, True),
                                                     (1, (2, 5), 16384, 2, True), (1, None, 4096, 2, True), (1, (1, 2), 2048, 3, False)])
def test_optimize_equals_the_restated_solve(mode, window, N, rounds, lq):
    from acmpc_amd import Engine
    H = 50
    n = H - 1
    prob = make_problem(orc, "monza", H, 4, seed=31 + N % 7)
    cfg = prob["cfg"]
    eng = Engine(**engine_kwargs(prob, mode, 1, N, n, nn_window=window, lq_candidate=lq))
    eng.set_paths(prob["table"])
    coef = eng.coefficients(0)
    u_ref = np.stack([prob["table"][orc.ROW_V], prob["table"][orc.ROW_KAPPA]], axis=1).astype(np.float32)
    centre = (u_ref + np.array([-0.7, 0.001], dtype=np.float32)).astype(np.float32)
    start = prob["x0"] if mode == 0 else prob["pose0"]
    sigma, shrink, seed = (0.5, 1.0e-3), 0.5, 0x5EED0123456
    got = eng.optimize(start[None], centre[None], u_ref[None], N, rounds, sigma, shrink=shrink, seed=seed)
    plan = None
    if lq:
        frenet = start.astype(np.float64) if mode == 0 else orc.frenet_start(prob["table"], start.astype(np.float64))
        plan = orc.lq_plan(prob["table"], frenet, cfg["step_cost"], cfg["r_term"], cfg["final_cost"], prob["u_lo"], prob["u_hi"])
        assert plan is not None
    want = orc.optimize_restated(mode, start, coef, centre, u_ref, N, rounds, sigma, shrink, seed, cfg["step_cost"],
                                 cfg["r_term"], cfg["final_cost"], prob["u_lo"], prob["u_hi"], 1.0e6, 0.05, window, extra=plan)
    _record_equals({k: got[k][0] for k in ("cost", "violation", "n_feasible", "u", "x")}, want, n)
    eng.close()


def _tick(H, cons, N, rounds, offset, seed):
    from acmpc_amd import _capi
    t = _capi.Tick()
    t.struct_size = _capi.C.sizeof(_capi.Tick)
    t.horizon, t.localised, t.has_end_velocity = H, 0, 1
    t.n_candidates, t.rounds, t.centre_is_reference = N, rounds, 1
    t.qp_max_iter, t.qp_check_every = 4000, 10
    t.offset = offset
    t.v_min, t.v_max, t.a_min, t.a_max = cons["v_min"], cons["v_max"], cons["a_min"], cons["a_max"]
    t.ay_max, t.ki_min, t.end_velocity = cons["ay_max"], cons["ki_min"], cons["end_velocity"]
    t.sigma[0], t.sigma[1], t.shrink = 0.5, 1e-3, 0.5
    t.qp_eps_abs = t.qp_eps_rel = 1e-3
    t.seed = seed
    return t


@pytest.mark.parametrize("mode,window", [(0, None), (1, None), (1, (2, 5))])
def test_the_tick_equals_the_restated_solve(mode, window):
    """Three consecutive ticks of acmpc_control_tick (16 384 candidates x 2 rounds, the controller's shape): each record
    is the restated solve on the tables the device prologue built, warm-started from the tick before, with the LQ plan for
    this tick's waypoints, pose and speed profile (the host's statements of the prologue's steps: acmpc_velocity_ceiling +
    the QP's exact optimum in two passes, acmpc_speed_profile_exact restated by the oracle) as the last round's candidate 2."""
    from acmpc_amd import Engine, _capi
    from acmpc_amd.mpc import waypoint_table
    H, N, rounds = 50, 16384, 2
    n = H - 1
    cfg = RACING["monza"]
    cons = dict(cfg["speed_profile_constraints"], v_max=28.0)
    lim = orc.vehicle_limits(2.65, 1.94, 0.30, cons["v_min"], cons["v_max"])
    lo, hi = orc.input_box(lim)
    eng = Engine(mode=mode, max_problems=1, max_candidates=N, max_steps=n, step_cost=cfg["step_cost"], r_term=cfg["r_term"],
                 final_cost=cfg["final_cost"], u_min=lo, u_max=hi, margin=lim.margin, wheelbase=lim.length, nn_window=window,
                 lq_candidate=True)
    centre, won = None, []
    for j in range(3):
        y = np.linspace(0, 120, H)
        bend = (1.0, 1.0, 1.1)[j]   # the second tick sees the first one's path again, the third a tighter one
        offset = (0.2, 0.2, 0.3)[j]
        coords = np.stack([0.004 * bend * y ** 2, y, np.linspace(10, 6, H)], axis=1)
        t = _tick(H, cons, N, rounds, offset, 900 + j)
        t.centre_is_reference = 1 if centre is None else 0
        out = eng.control_tick(t, coords, centre)
        assert out["info"][4] == 0 and out["info"][7] == 0
        x0, u_ref, coef = eng.tick_device_tables(n)
        # this tick's waypoints, pose and speed profile - the host's statements of the prologue's steps
        table = waypoint_table(coords, 1e-12)
        ceiling = _capi.velocity_ceiling(table[3], cons["ay_max"], cons["ki_min"], cons["v_min"], cons["v_max"], False,
                                         cons["end_velocity"])
        table[6] = orc.speed_profile_exact(ceiling, table[4], cons["a_min"], cons["a_max"], cons["v_min"])   # (feasible here)
        frenet = orc.frenet_start(table, np.array([offset, 0.0, np.pi / 2]))
        plan = orc.lq_plan(table, frenet, cfg["step_cost"], cfg["r_term"], cfg["final_cost"], lo, hi)
        assert plan is not None
        want = orc.optimize_restated(mode, x0, coef, u_ref if centre is None else centre, u_ref, N, rounds, (0.5, 1e-3), 0.5,
                                     900 + j, cfg["step_cost"], cfg["r_term"], cfg["final_cost"], lo, hi, 1.0e6, 0.05, window,
                                     extra=plan)
        rec = out["record"]
        got = dict(cost=rec[0], violation=rec[1], n_feasible=rec[2], u=rec[4:4 + 2 * n].reshape(n, 2),
                   x=rec[4 + 2 * n:].reshape(n + 1, 3))
        _record_equals(got, want, n)
        won.append(want["winners"][-1])
        centre = got["u"].copy()
    if mode == 0:
        # on the path whose speed profile it was planned with the LQ plan is the QP's optimum: it wins the FIRST tick's last
        # round (round 5: a first tick has a plan), and the second tick - the same path - starts from it as its centre, which
        # ties with the re-planned candidate 2 and takes the lower index
        assert won[0] == 2 and won[1] in (0, 2), "the LQ plan never won a last round: %s" % won
    eng.close()


@pytest.mark.parametrize("mode", [0, 1])
def test_the_tick_with_the_box_constrained_plan_equals_the_restated_solve(mode):
    """`lq_candidate = 2` (the controller's default, round 5) on a path where the QP's box rows are active - a corner of
    radius 8 m after 30 m of straight under the racing corridor: the last round's candidate 2 is the plan of
    csrc/acmpc_lq_box.h, warm-started from tick to tick.  Four consecutive ticks equal `oracle.optimize_restated` with
    `oracle.lq_box_plan` (the host solver's line-by-line restatement, iterate chained the same way) as that candidate,
    exactly - and the refinement did run (acmpc_lq_box_stats)."""
    from acmpc_amd import Engine, _capi
    from acmpc_amd import workloads as wl
    from acmpc_amd.mpc import waypoint_table
    H, N, rounds = 50, 16384, 2
    n = H - 1
    cfg = RACING["monza"]
    cons = dict(cfg["speed_profile_constraints"], v_max=28.0)
    lim = orc.vehicle_limits(2.65, 1.94, 0.30, cons["v_min"], cons["v_max"])
    lo, hi = orc.input_box(lim)
    w_bound = 1.0e4
    eng = Engine(mode=mode, max_problems=1, max_candidates=N, max_steps=n, step_cost=cfg["step_cost"], r_term=cfg["r_term"],
                 final_cost=cfg["final_cost"], u_min=lo, u_max=hi, margin=lim.margin, wheelbase=lim.length, nn_window=None,
                 lq_candidate=2, w_bound=w_bound)
    base = wl.corner_entry_path(8.0, 30.0, H)
    centre, state, ran = None, None, []
    for j in range(4):
        coords = np.array(base)
        coords[:, 1] -= (0.0, 0.0, 0.25, 0.5)[j]          # the car moves on along the straight
        offset = (0.0, 0.0, 0.04, 0.08)[j]
        t = _tick(H, cons, N, rounds, offset, 40 + j)
        t.centre_is_reference = 1 if centre is None else 0
        out = eng.control_tick(t, coords, centre)
        assert out["info"][4] == 0 and out["info"][7] == 0
        stats = eng.lq_box_stats()
        x0, u_ref, coef = eng.tick_device_tables(n)
        table = waypoint_table(coords, 1e-12)
        ceiling = _capi.velocity_ceiling(table[3], cons["ay_max"], cons["ki_min"], cons["v_min"], cons["v_max"], False,
                                         cons["end_velocity"])
        table[6] = orc.speed_profile_exact(ceiling, table[4], cons["a_min"], cons["a_max"], cons["v_min"])   # (feasible here)
        frenet = orc.frenet_start(table, np.array([offset, 0.0, np.pi / 2]))
        restated = orc.lq_box_plan(table, frenet, cfg["step_cost"], cfg["r_term"], cfg["final_cost"], lo, hi, lim.margin, w_bound,
                                   40, state)
        state = restated["state"]
        assert restated["plan"] is not None
        assert (stats["iterations"], stats["chosen"], stats["triggered"]) == (restated["iterations"], restated["chosen"],
                                                                               restated["triggered"]), (j, stats, restated)
        ran.append(stats["iterations"])
        want = orc.optimize_restated(mode, x0, coef, u_ref if centre is None else centre, u_ref, N, rounds, (0.5, 1e-3), 0.5,
                                     40 + j, cfg["step_cost"], cfg["r_term"], cfg["final_cost"], lo, hi, w_bound, 0.05, None,
                                     extra=restated["plan"])
        rec = out["record"]
        got = dict(cost=rec[0], violation=rec[1], n_feasible=rec[2], u=rec[4:4 + 2 * n].reshape(n, 2),
                   x=rec[4 + 2 * n:].reshape(n + 1, 3))
        _record_equals(got, want, n)
        centre = got["u"].copy()
    assert ran[0] >= 10 and all(k >= 1 for k in ran), ran       # cold: tens of iterations; warm: it confirms or continues
    eng.close()


@pytest.mark.parametrize("name", ["solve_monza_H50_S", "solve_monza_H50_T", "solve_monza_H50_T_window"])
def test_optimize_reproduces_the_frozen_solves(name):
    """The records frozen in tests/golden/composition.npz (tests/golden/gen_composition.py: the oracle's restated solve at
    the closed-loop shape) out of acmpc_optimize, bit for bit."""
    from acmpc_amd import Engine
    g = np.load(GOLDEN)
    k = name + "/"
    mode, N, rounds = int(g[k + "mode"]), int(g[k + "n_candidates"]), int(g[k + "rounds"])
    window = tuple(int(v) for v in g[k + "window"])
    window = None if window[0] < 0 else window
    table = g[k + "table"]
    n = table.shape[1]
    w, box = g[k + "weights"], g[k + "box"]
    eng = Engine(mode=mode, max_problems=1, max_candidates=N, max_steps=n, step_cost=w[:3], r_term=w[3:5], final_cost=w[5:],
                 u_min=box[:2], u_max=box[2:], margin=float(g[k + "margin"]), wheelbase=2.65, nn_window=window,
                 lq_candidate=True)
    eng.set_paths(table)
    eng.set_coefficients(g[k + "coef"])   # (the frozen packed table: a host cos / sin may round its last bit otherwise)
    got = eng.optimize(g[k + "start"][None], g[k + "centre"][None], g[k + "u_ref"][None], N, rounds,
                       tuple(g[k + "sigma"]), shrink=float(g[k + "shrink"]), seed=int(g[k + "seed"]))
    assert got["cost"][0] == g[k + "cost"] and got["violation"][0] == g[k + "violation"]
    assert int(got["n_feasible"][0]) == int(g[k + "n_feasible"])
    np.testing.assert_array_equal(got["u"][0], g[k + "u"])
    np.testing.assert_array_equal(got["x"][0], g[k + "x"])
    eng.close()
