"""Shared test inputs: the reference's per-track `racing.control` blocks (configs/<track>.yaml:67-81) and the
builder-chosen placeholder vehicle (the reference's vehicle file is a network asset)."""
from types import SimpleNamespace

RACING = {
    "monza": dict(horizon=50, unlocalised_max_speed=28,
                  speed_profile_constraints=dict(v_min=8.0, v_max=84.0, a_min=-1.3, a_max=1.0, ay_max=5.5,
                                                 ki_min=0.005, end_velocity=14.0),
                  step_cost=[4.0e-3, 5.0e-2, 0.0], r_term=[1.0e-2, 10.0], final_cost=[1.0, 0.0, 0.1]),
    "spa": dict(horizon=50, unlocalised_max_speed=8.0,
                speed_profile_constraints=dict(v_min=5.0, v_max=84.0, a_min=-1.0, a_max=1.0, ay_max=4.0,
                                               ki_min=0.003, end_velocity=20.0),
                step_cost=[1.0e-3, 0.0, 0.0], r_term=[1.0e-2, 10.0], final_cost=[1.0, 0.0, 0.1]),
    "nordschleife": dict(horizon=50, unlocalised_max_speed=20,
                         speed_profile_constraints=dict(v_min=12.0, v_max=84.0, a_min=-1.0, a_max=1.0,
                                                        ay_max=3.0, ki_min=0.0, end_velocity=14.0),
                         step_cost=[2.0e-4, 0.0, 0.0], r_term=[1.0e-2, 10.0], final_cost=[1.0, 0.0, 0.1]),
    "silverstone": dict(horizon=50, unlocalised_max_speed=32.0,
                        speed_profile_constraints=dict(v_min=8.0, v_max=84.0, a_min=-1.0, a_max=1.0,
                                                       ay_max=5.0, ki_min=0.003, end_velocity=20.0),
                        step_cost=[2.0e-3, 5.0e-2, 0.0], r_term=[1.0e-2, 10.0], final_cost=[1.0, 0.0, 0.1]),
}

WHEELBASE, WIDTH, DELTA_MAX = 2.65, 1.94, 0.30


class PlaceholderVehicle:
    """Duck-type of ace.steering.SteeringGeometry as used at dynamics.py:11-13."""

    def __init__(self):
        self.vehicle_data = SimpleNamespace(wheelbase=WHEELBASE, width=WIDTH)

    def max_steering_angle(self):
        return DELTA_MAX


# ---------------------------------------------------------------------------------------------------------
# Seeded synthetic problems shared by the CPU and GPU tests (inputs only; expected values come from the oracle)
# ---------------------------------------------------------------------------------------------------------
import numpy as np  # noqa: E402


def local_path(H: int, seed: int, length: float = 150.0) -> np.ndarray:
    """A smooth H x 3 reference path in the vehicle frame (car at the origin heading +y,
    spatial_mpc.py:187), widths linspace(10, 6, H) as controller.py:256-267 builds them."""
    rng = np.random.default_rng(seed)
    s = np.linspace(0.0, length, H)
    a1, a2 = rng.uniform(-12, 12), rng.uniform(-5, 5)
    x = a1 * (s / length) ** 2 + a2 * np.sin(2 * np.pi * s / length * rng.uniform(0.3, 1.0)) * (s / length)
    return np.stack([x, s, np.linspace(10.0, 6.0, H)], axis=1)


def make_problem(orc, track: str, H: int, N: int, seed: int, sigma=(2.0, 0.01)):
    """Returns dict(table, limits, weights, x0_spatial, pose0, U[N,n,2], u_lo, u_hi) for one problem."""
    cfg = RACING[track]
    # the control process overwrites v_max with the reference speed every tick (controller.py:241-243)
    cons = dict(cfg["speed_profile_constraints"], v_max=float(cfg["unlocalised_max_speed"]))
    rng = np.random.default_rng(1000 + seed)
    coords = local_path(H, seed)
    table = orc.construct_waypoints(coords)
    sp = orc.speed_profile_qp(table, cons, cons["end_velocity"], False)
    table[orc.ROW_V] = np.clip(sp["v_hi"] - 2.0, cons["v_min"], cons["v_max"])  # surrogate speed profile
    limits = orc.vehicle_limits(WHEELBASE, WIDTH, DELTA_MAX, cons["v_min"], cons["v_max"])
    offset = float(rng.uniform(-0.5, 0.5))
    pose0 = np.array([offset, 0.0, np.pi / 2])
    x0 = orc.t2s(table[:3, 0], pose0)
    n = H - 1
    u_ref = np.stack([table[orc.ROW_V], table[orc.ROW_KAPPA]], axis=1)
    spread = rng.uniform(0.02, 1.0, (N, 1, 1))  # per-candidate noise scale: a mix of feasible and infeasible
    U = u_ref[None] + rng.standard_normal((N, n, 2)) * np.asarray(sigma) * spread
    u_lo, u_hi = orc.input_box(limits)
    U = np.clip(U, u_lo, u_hi)
    if N > 3:  # a few candidates outside the box so that the violation branch is exercised
        U[3, n // 2, 0] = u_hi[0] + 1.5
        U[2, 1, 1] = u_lo[1] - 0.02
    U[0] = u_ref
    return dict(table=table, limits=limits, cfg=cfg, x0=x0.astype(np.float32), pose0=pose0.astype(np.float32),
                U=U.astype(np.float32), u_lo=u_lo, u_hi=u_hi, coords=coords)


def engine_kwargs(problem, mode: int, P: int, N: int, n: int, **extra):
    cfg, lim = problem["cfg"], problem["limits"]
    kw = dict(mode=mode, max_problems=P, max_candidates=N, max_steps=n, step_cost=cfg["step_cost"],
              r_term=cfg["r_term"], final_cost=cfg["final_cost"], u_min=problem["u_lo"], u_max=problem["u_hi"],
              margin=lim.margin, wheelbase=lim.length, t_min=0.01, dt=0.05, w_bound=1.0e6, softmin_lambda=0.5)
    kw.update(extra)
    return kw


def full_size_controls(orc, prob, N: int, n: int, seed: int = 99) -> np.ndarray:
    """U [N, n, 2] float32 for the full-size BASELINE shapes: seeded noise round the reference controls with a
    per-candidate spread, clipped to the input box, candidate 0 = the reference controls."""
    rng = np.random.default_rng(seed)
    u_ref = np.stack([prob["table"][orc.ROW_V], prob["table"][orc.ROW_KAPPA]], axis=1)
    U = (u_ref[None] + rng.standard_normal((N, n, 2), dtype=np.float32) * np.array([2.0, 0.01], dtype=np.float32)
         * rng.uniform(0.02, 1.0, (N, 1, 1)).astype(np.float32)).astype(np.float32)
    np.clip(U, prob["u_lo"].astype(np.float32), prob["u_hi"].astype(np.float32), out=U)
    U[0] = u_ref
    return U

