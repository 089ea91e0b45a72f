"""Synthetic inputs for benchmarks and tests.  The reference ships no maps, vehicle data or recordings (they
are network assets, scripts/download_assets.sh), so circuits are seeded closed curves at the reference's map
resolution (0.5 m, mapping/map_maker.py:203; 9.5 m wide, agent.py:288) with builder-supplied nominal lengths,
and the per-track weights/constraints are the reference's `configs/<track>.yaml` `racing.control` blocks.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Dict, Tuple

import numpy as np

from .mpc import waypoint_table
from .reference_path import ReferencePath
from .speed_profile import SpeedProfileSolver

# nominal lap lengths [m] (builder-supplied, not from the reference) and generator seeds
TRACKS = {"monza": (5793.0, 0), "spa": (7004.0, 1), "nordschleife": (25378.0, 2), "silverstone": (5891.0, 3)}

# configs/<track>.yaml:67-81
RACING_CONTROL = {
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

# placeholder vehicle (the reference's vehicle file is a network asset): wheelbase, width [m], max steer [rad]
VEHICLE = SimpleNamespace(wheelbase=2.65, width=1.94, delta_max=0.30)

BEV_LOOKAHEAD_M = 150.0      # perception/tracks.py:14
CENTRELINE_POINTS = 500      # perception centreline length (controller.py:102-108)


class PlaceholderVehicle:
    """Duck-type of ace.steering.SteeringGeometry as the model uses it (dynamics.py:11-13)."""

    def __init__(self, spec=VEHICLE):
        self.vehicle_data = SimpleNamespace(wheelbase=spec.wheelbase, width=spec.width)
        self._delta_max = spec.delta_max

    def max_steering_angle(self) -> float:
        return self._delta_max


def synthetic_track(name: str, spacing: float = 0.5, width: float = 9.5) -> Dict[str, np.ndarray]:
    """Closed circuit {centre, left, right} (each M x 2) in the format of utils/load.py:9-35."""
    length, seed = TRACKS[name]
    rng = np.random.default_rng(seed)
    theta = np.linspace(0.0, 2 * np.pi, 20001)[:-1]
    radius = np.ones_like(theta)
    for k in range(2, 9):
        radius += rng.uniform(0.02, 0.12) / (k - 1) * np.cos(k * theta + rng.uniform(0, 2 * np.pi))
    curve = np.stack([1.35 * radius * np.cos(theta), radius * np.sin(theta)], axis=1)
    closed = np.vstack([curve, curve[:1]])
    arc = np.concatenate([[0.0], np.cumsum(np.linalg.norm(np.diff(closed, axis=0), axis=1))])
    closed *= length / arc[-1]
    arc *= length / arc[-1]
    s = np.arange(0.0, length, spacing)
    centre = np.stack([np.interp(s, arc, closed[:, 0]), np.interp(s, arc, closed[:, 1])], axis=1)
    tangent = np.roll(centre, -1, axis=0) - np.roll(centre, 1, axis=0)
    tangent /= np.linalg.norm(tangent, axis=1, keepdims=True)
    normal = np.stack([-tangent[:, 1], tangent[:, 0]], axis=1)
    return dict(centre=centre, left=centre + 0.5 * width * normal, right=centre - 0.5 * width * normal,
                spacing=spacing)


def local_centreline(track: Dict[str, np.ndarray], index: int, lateral_offset: float = 0.0,
                     points: int = CENTRELINE_POINTS) -> np.ndarray:
    """The next 150 m of centreline seen from the pose at `index`, in the vehicle frame (car at the origin,
    heading +y), resampled to `points` (500) points - what perception publishes."""
    centre = track["centre"]
    count = int(round(BEV_LOOKAHEAD_M / track["spacing"])) + 1
    window = centre[(index + np.arange(count)) % len(centre)]
    heading = np.arctan2(*(window[1] - window[0])[::-1])
    rot = np.pi / 2 - heading
    c, s = np.cos(rot), np.sin(rot)
    local = (window - window[0]) @ np.array([[c, s], [-s, c]])
    local[:, 0] -= lateral_offset
    t = np.linspace(0, count - 1, points)
    return np.stack([np.interp(t, np.arange(count), local[:, 0]), np.interp(t, np.arange(count), local[:, 1])],
                    axis=1).astype(np.float32)


def reference_path_from_centreline(centreline: np.ndarray, horizon: int) -> np.ndarray:
    """500 x 2 centreline -> H x 3 path with widths linspace(10, 6, H) (controller.py:256-267).  Like the
    reference this needs len(centreline)/horizon rows to come out at exactly H (true for its horizons 50, 100)."""
    stride = int(len(centreline) / horizon)
    picked = centreline[0::stride]
    if len(picked) != horizon:
        raise ValueError("a %d-point centreline does not downsample to horizon %d" % (len(centreline), horizon))
    return np.stack([picked[:, 0], picked[:, 1], np.linspace(10.0, 6.0, horizon)]).T


def problem_batch(track_name: str, n_problems: int, horizon: int, seed: int = 0) -> SimpleNamespace:
    """P poses spread round the circuit -> tables [P,7,n] (with a solved speed profile), Frenet start states
    [P,3] (mode S), Cartesian poses [P,3] (mode T) and the model/weights they were built with."""
    from .bicycle_model import SpatialBicycleModel

    cfg = RACING_CONTROL[track_name]
    # the control process overwrites v_max with the reference speed every tick (controller.py:241-243)
    cons = dict(cfg["speed_profile_constraints"], v_max=float(cfg["unlocalised_max_speed"]))
    model = SpatialBicycleModel(PlaceholderVehicle(), {"min": cons["v_min"], "max": cons["v_max"]})
    track = synthetic_track(track_name)
    rng = np.random.default_rng(seed)
    n = horizon - 1
    solver = SpeedProfileSolver({"control_horizon": n, "max_iterations": 4000, "constraints": cons})
    tables = np.zeros((n_problems, 7, n))
    x0 = np.zeros((n_problems, 3), dtype=np.float32)
    pose0 = np.zeros((n_problems, 3), dtype=np.float32)
    starts = (np.linspace(0, len(track["centre"]), n_problems, endpoint=False)).astype(int)
    for p, index in enumerate(starts):
        offset = float(rng.uniform(-0.5, 0.5))
        # horizons that do not divide 500 (e.g. 80) get the nearest centreline length that does downsample
        points = horizon * (CENTRELINE_POINTS // horizon)
        coords = reference_path_from_centreline(local_centreline(track, int(index), points=points), horizon)
        path = ReferencePath.from_table(waypoint_table(coords))
        dec = solver.solve(path, cons["end_velocity"])
        path.velocities = dec.x if dec.info.status == "solved" else np.clip(
            solver.velocity_ceiling(path, cons["end_velocity"]) - 2.0, cons["v_min"], cons["v_max"])
        tables[p] = path.table
        pose = np.array([offset, 0.0, np.pi / 2])
        x0[p] = model.t2s(path.get_state(0), pose)
        pose0[p] = pose
    u_lo = np.array([model.min_u[0] - 0.1, model.min_u[1]])
    u_hi = np.array([model.max_u[0] + 0.1, model.max_u[1]])
    return SimpleNamespace(track=track_name, cfg=cfg, constraints=cons, model=model, tables=tables, x0=x0,
                           pose0=pose0, u_lo=u_lo, u_hi=u_hi, n=n, horizon=horizon)


def engine_kwargs(batch: SimpleNamespace, mode: int, max_candidates: int, **extra) -> Dict:
    kw = dict(mode=mode, max_problems=batch.tables.shape[0], max_candidates=max_candidates, max_steps=batch.n,
              step_cost=batch.cfg["step_cost"], r_term=batch.cfg["r_term"], final_cost=batch.cfg["final_cost"],
              u_min=batch.u_lo, u_max=batch.u_hi, margin=batch.model.margin, wheelbase=batch.model.length,
              t_min=0.01, dt=0.05, w_bound=1.0e6, softmin_lambda=0.5)
    kw.update(extra)
    return kw


def sample_candidates(batch: SimpleNamespace, n_candidates: int, seed: int, sigma: Tuple[float, float] = (2.0, 0.01),
                      layout: int = 0) -> np.ndarray:
    """U = u_ref + sigma * N(0,1) clipped to the input box, candidate 0 = u_ref exactly.
    Returned float32 as [P,N,n,2] (layout 0) or [P,n,2,N] (layout 1)."""
    rng = np.random.default_rng(seed)
    P, n = batch.tables.shape[0], batch.n
    u_ref = np.stack([batch.tables[:, 6, :], batch.tables[:, 3, :]], axis=2)  # [P,n,2]
    U = u_ref[:, None] + rng.standard_normal((P, n_candidates, n, 2)) * np.asarray(sigma)
    np.clip(U, batch.u_lo, batch.u_hi, out=U)
    U[:, 0] = np.clip(u_ref, batch.u_lo, batch.u_hi)
    U = U.astype(np.float32)
    return U if layout == 0 else np.ascontiguousarray(U.transpose(0, 2, 3, 1))


# ---------------------------------------------------------------------------------------------------------
# The four synthetic path families the reference's own MPC script drives the controller with
# (/root/reference/src/acmpc/control/utils.py:11-32, used by tests/test_spatial_mpc.py:45-75): H x 3 paths in the
# vehicle frame, turned by `angle` about the origin (clockwise positive, as there).
# ---------------------------------------------------------------------------------------------------------
def _turned(x: np.ndarray, y: np.ndarray, angle: float, width: float) -> np.ndarray:
    c, s = np.cos(angle), np.sin(angle)
    return np.column_stack([c * x + s * y, c * y - s * x, np.full(x.shape[0], float(width))])


def family_path(kind: str, parameter: float, horizon: int, angle: float = 0.0, width: float = 100.0,
                chicane_width: float = 40.0) -> np.ndarray:
    """`kind`: "hairpin" (three quarters of a circle of radius `parameter` to the left), "chicane" (a logistic
    side-step of `chicane_width` centred `parameter` metres ahead), "curve" (a parabola with coefficient `parameter`
    leaving along +x) or "straight" (`parameter` metres ahead)."""
    if kind == "hairpin":
        phi = np.linspace(0.0, 1.5 * np.pi, horizon)
        return _turned(parameter * np.cos(phi) - parameter, parameter * np.sin(phi), angle, width)
    if kind == "chicane":
        ahead = np.linspace(0.0, 100.0, horizon)
        return _turned(chicane_width / (1.0 + np.exp(-0.1 * (ahead - parameter))), ahead, angle, width)
    if kind == "curve":
        along = np.linspace(0.0, 100.0, horizon)
        return _turned(along, parameter * along**2, angle, width)
    if kind == "straight":
        return _turned(np.zeros(horizon), np.linspace(0.0, parameter, horizon), angle, width)
    raise ValueError("unknown path family %r" % kind)



# ---------------------------------------------------------------------------------------------------------
# Racing-configuration paths on which the reference QP's BOX rows are active (round 5, tests/test_gpu_qp_gap.py,
# tests/test_lq_box.py): the corridor e_y in +-(w/2 - margin) under the widths linspace(10, 6, H) the control process
# hands over (/root/reference/src/acmpc/control/controller.py:256-267; control/solvers/control.py:57-60) and the input
# box (control.py:130-139) - corners at or inside the steering limit kappa_max = tan(delta_max) / L, entered from a
# straight, so that the QP's optimum prepares for them steps ahead.
# ---------------------------------------------------------------------------------------------------------
def curvature_path(kappa_of_s, length: float, horizon: int) -> np.ndarray:
    """H x 3 path in the vehicle frame (from the origin, heading +y) with curvature `kappa_of_s(s)` over `length` metres,
    widths linspace(10, 6, H)."""
    s = np.linspace(0.0, length, 4001)
    k = np.asarray(kappa_of_s(s), dtype=np.float64)
    step = np.diff(s)
    psi = np.pi / 2 + np.concatenate([[0.0], np.cumsum(0.5 * (k[1:] + k[:-1]) * step)])
    x = np.concatenate([[0.0], np.cumsum(0.5 * (np.cos(psi[1:]) + np.cos(psi[:-1])) * step)])
    y = np.concatenate([[0.0], np.cumsum(0.5 * (np.sin(psi[1:]) + np.sin(psi[:-1])) * step)])
    at = np.linspace(0.0, length, horizon)
    return np.column_stack([np.interp(at, s, x), np.interp(at, s, y), np.linspace(10.0, 6.0, horizon)])


def corner_entry_path(radius: float, lead: float, horizon: int, arc_angle: float = 1.2 * np.pi, tail: float = 0.0) -> np.ndarray:
    """`lead` metres straight ahead, then a left-hand arc of `radius` through `arc_angle`, then `tail` metres straight."""
    arc = arc_angle * radius
    return curvature_path(lambda s: np.where((s > lead) & (s <= lead + arc), 1.0 / radius, 0.0), lead + arc + tail, horizon)


def s_bend_path(radius: float, lead: float, horizon: int, angle: float = 0.5 * np.pi, tail: float = 20.0) -> np.ndarray:
    """`lead` metres straight, a left-hand arc of `radius` through `angle`, the same to the right, `tail` metres straight."""
    arc = angle * radius
    return curvature_path(lambda s: np.where(s <= lead, 0.0, np.where(s <= lead + arc, 1.0 / radius,
                                                                      np.where(s <= lead + 2 * arc, -1.0 / radius, 0.0))),
                          lead + 2 * arc + tail, horizon)


def racing_widths(path: np.ndarray) -> np.ndarray:
    """The same path with the corridor the control process gives it: widths linspace(10, 6, H) (controller.py:256-267)."""
    out = np.array(path, dtype=np.float64)
    out[:, 2] = np.linspace(10.0, 6.0, out.shape[0])
    return out
