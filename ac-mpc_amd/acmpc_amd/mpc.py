"""`SpatialMPC` and `build_mpc` with the reference's Python surface
(/root/reference/src/acmpc/control/spatial_mpc.py:20-217, control/controller.py:19-29): same constructor, same
`get_control(reference_path, is_localised=False, offset=0.0) -> None`, same read-after-call attributes
(`projected_control`, `current_prediction`, `cum_time`, `times`, `accelerations`, `steer_rates`, `reference_path`,
`speed_profile`, `infeasibility_counter`, `MPC_horizon`, `delta_max`, `model`, `speed_profile_constraints`), so it
drops into the ACI control loop's `ControlProcess` unchanged.  The optimisation inside is the GPU
rollout-and-cost engine (`sampling_solver.ControlSolver`) instead of OSQP.
"""
from __future__ import annotations

import copy
import logging
import math
from typing import Dict

import numpy as np

from . import _capi
from .bicycle_model import SpatialBicycleModel, wrap_angle
from .reference_path import ReferencePath
from .sampling_solver import SOLVED, ControlSolver
from .speed_profile import LocalisedSpeedProfileSolver, SpeedProfileSolver

logger = logging.getLogger("acmpc_amd")

MAX_SOLVER_ITERATIONS_MAP = 40000
MAX_SOLVER_ITERATIONS = 4000


def waypoint_table(waypoint_coordinates: np.ndarray, eps: float = 1e-12) -> np.ndarray:
    """H x 3 `[x, y, width]` -> 7 x n table with n = H - 1 (what spatial_mpc.py:125-154 computes), by the library's
    host routine (`acmpc_waypoint_table`: the same operations as `waypoint_table_numpy` below without ten NumPy
    dispatches on 50-element arrays)."""
    return _capi.waypoint_table(waypoint_coordinates, eps)


def waypoint_table_numpy(waypoint_coordinates: np.ndarray, eps: float = 1e-12) -> np.ndarray:
    """The NumPy statement of `waypoint_table` (tests hold the two to 1e-12 and both to the reference's vectors).

    Heading and spacing come from the segment to the next point, the width is the next point's, curvature is
    the wrapped heading change against the previous segment over the spacing.  The previous segment of point 0
    closes the loop to the last point; its curvature is then replaced by point 1's, as the reference does."""
    pts = np.asarray(waypoint_coordinates, dtype=np.float64)
    xy = pts[:, :2]
    n = len(pts) - 1
    seg_ahead = xy[1:] - xy[:-1]
    seg_behind = xy[:-1] - np.roll(xy, 1, axis=0)[:-1]
    table = np.zeros((7, n))
    table[0], table[1] = xy[:-1, 0], xy[:-1, 1]
    table[2] = np.arctan2(seg_ahead[:, 1], seg_ahead[:, 0])
    table[4] = np.linalg.norm(seg_ahead, axis=1)
    table[5] = pts[1:, 2]
    turn = wrap_angle(table[2] - np.arctan2(seg_behind[:, 1], seg_behind[:, 0]))
    table[3] = turn / (table[4] + eps) + eps
    table[3, 0] = table[3, 1]
    return table


class SpatialMPC:
    def __init__(self, config: Dict, model: SpatialBicycleModel):
        self.MPC_horizon = config["horizon"]
        self.model = model
        self.nx, self.nu = 3, 2
        self._eps = 1e-12
        self.speed_profile_constraints = config["speed_profile_constraints"]  # live dict, mutated by the caller
        self.ay_max = self.speed_profile_constraints["ay_max"]
        self.delta_max = model.delta_max
        self.current_prediction = None
        self.infeasibility_counter = 0
        self.cum_time = np.zeros(1)
        self.projected_control = np.zeros((self.nu, self.MPC_horizon))
        solver_config = copy.deepcopy(config)
        solver_config["max_iterations"] = MAX_SOLVER_ITERATIONS
        self._control_solver = ControlSolver(solver_config, model)
        profile_config = {
            "control_horizon": self.MPC_horizon - 1,
            "max_iterations": MAX_SOLVER_ITERATIONS,
            "constraints": self.speed_profile_constraints,
            "check_every": int(config.get("speed_profile_check_every", 5)),
            "method": str(config.get("speed_profile_method", "exact")),
        }
        self._speed_profile_solver = SpeedProfileSolver(profile_config)
        self._localised_speed_profile_solver = LocalisedSpeedProfileSolver(profile_config)
        # optional build keys: `device_prologue` (default on: the per-tick prologue runs on the GPU in front of the
        # sampling rounds; off = the host statements of the same steps), `speed_profile_check_every` (the QP's stopping
        # test runs every this many iterations; warm-started from the previous tick it passes at the first test, so 5
        # halves the prologue against OSQP's customary 10-25 at the same 1e-3 tolerances)
        self._device_prologue = bool(config.get("device_prologue", True))
        self._qp_check_every = int(config.get("speed_profile_check_every", 5))
        # `speed_profile_method`: "exact" (default) = the QP's optimum in two passes, the OSQP-style splitting only for a
        # problem the passes do not solve (infeasible: the reference's status then); "admm" = always the splitting
        self._speed_profile_method = profile_config["method"]

    # -- speed profiles -----------------------------------------------------------------------------------
    def compute_map_speed_profile(self, reference_path: ReferencePath, ay_max: float, a_min: float) -> ReferencePath:
        """Whole-lap profile at race start (spatial_mpc.py:60-87): own solver sized to the lap."""
        constraints = copy.deepcopy(self.speed_profile_constraints)
        constraints.update(a_min=a_min, ay_max=ay_max)
        solver = SpeedProfileSolver({"control_horizon": len(reference_path), "max_iterations": MAX_SOLVER_ITERATIONS_MAP,
                                     "constraints": constraints, "method": self._speed_profile_method})
        return self._compute_speed_profile(solver, reference_path)

    def compute_speed_profile(self, reference_path: ReferencePath, is_localised: bool = False,
                              end_vel=None) -> ReferencePath:
        solver = self._localised_speed_profile_solver if is_localised else self._speed_profile_solver
        return self._compute_speed_profile(solver, reference_path, end_vel)

    def _compute_speed_profile(self, solver, reference_path: ReferencePath, end_vel=None) -> ReferencePath:
        dec = solver.solve(reference_path, end_vel)
        if dec.info.status == "solved":
            reference_path.velocities = dec.x
            self.speed_profile = dec.x
        else:  # keep whatever velocities the path already carries (spatial_mpc.py:119-122)
            logger.warning("Infeasible speed profile (%s); keeping previous velocities", dec.info.status)
        return reference_path

    # -- waypoints / prediction ---------------------------------------------------------------------------
    def construct_waypoints(self, waypoint_coordinates: np.ndarray) -> ReferencePath:
        return ReferencePath.from_table(waypoint_table(waypoint_coordinates, self._eps))

    def update_prediction(self, spatial_state_prediction: np.ndarray, reference_path: ReferencePath) -> np.ndarray:
        """Predicted Frenet states -> n x 2 Cartesian points (spatial_mpc.py:156-168)."""
        return self.model.s2t(reference_path, spatial_state_prediction)[:-1].T

    # -- the entry point ----------------------------------------------------------------------------------
    def get_control(self, reference_path: np.ndarray, is_localised: bool = False, offset: float = 0.0,
                    elapsed: float = None):
        """One MPC solve for an H x 3 reference path given in the vehicle frame (car at the origin, heading +y,
        laterally displaced by `offset`).  Returns None; results are left in attributes.  `elapsed` (optional,
        not in the reference's signature): seconds since the previous solve - the sampler then starts from the
        previous plan advanced by that time instead of the plan as it was."""
        n = self.MPC_horizon - 1
        if elapsed is not None and self.cum_time.shape[0] == n:
            self._control_solver.shift_warm_start(elapsed, self.cum_time)
        if self._device_prologue and self._control_solver.supports_tick():
            return self._get_control_tick(reference_path, is_localised, offset)
        path = self.construct_waypoints(reference_path)
        path = self.compute_speed_profile(path, is_localised,
                                          end_vel=self.speed_profile_constraints["end_velocity"])
        spatial_state = self.model.t2s(path.get_state(0), np.array([offset, 0.0, math.pi / 2]))
        self._control_solver.pose = (float(offset), 0.0, math.pi / 2)   # what `rollout_mode: "T"` starts from
        dec = self._control_solver.solve(spatial_state, path)

        if dec.info.status != SOLVED:
            # keep the previous plan, count the failure (spatial_mpc.py:212-217)
            logger.warning("Infeasible problem! Failed %d time(s).", self.infeasibility_counter)
            self.infeasibility_counter += 1
            return

        if self._control_solver.temporal:
            (self.projected_control, self.current_prediction, self.cum_time, self.times, self.accelerations,
             self.steer_rates) = _capi.unpack_decision_temporal(dec.x, n, self._control_solver._dt, self.model.length)
        else:
            (self.projected_control, self.current_prediction, self.cum_time, self.times, self.accelerations,
             self.steer_rates) = _capi.unpack_decision(dec.x, n, path.table, self.model.length)
        self.reference_path = path
        self.infeasibility_counter = 0


    # -- reference path straight from the map (not in the reference: there perception publishes the centre line) -----
    def bind_map(self, track_map: Dict[str, np.ndarray]) -> None:
        """`track_map`: the dict of utils/load.py:9-35 (`centre` [M, 2], `spacing` in metres).  After this,
        `get_control_at` cuts the reference path out of the map on the device."""
        self._control_solver.bind_map(track_map["centre"], float(track_map.get("spacing", 0.5)))

    def get_control_at(self, map_index: int = -1, pose=(0.0, 0.0), lateral_offset: float = 0.0,
                       is_localised: bool = False, offset: float = 0.0, centreline_points: int = None):
        """One solve for a pose on the bound map: the 150 m window from `map_index` (or from the map point nearest to
        `pose`) is moved into the vehicle frame, resampled and downsampled to the H x 3 path on the device
        (`workloads.local_centreline` + `ControlProcess._reference_path`, controller.py:256-267), and feeds the same
        prologue and rounds as `get_control`.  The path used is left in `self.reference_coordinates`."""
        if not (self._device_prologue and self._control_solver.supports_tick()):
            raise ValueError("get_control_at needs the device prologue (argmin centre update, horizon <= 129)")
        points = centreline_points or self.MPC_horizon * (500 // self.MPC_horizon)
        return self._get_control_tick(None, is_localised, offset, map_index=int(map_index), pose=pose,
                                      lateral_offset=float(lateral_offset), centreline_points=points)

    def _get_control_tick(self, reference_path, is_localised: bool, offset: float, **from_map):
        """The same solve as one round trip (`acmpc_control_tick`): waypoints, speed profile, Frenet start state and
        linearisation run on the device in front of the sampling rounds; this method only keeps the reference's
        bookkeeping (spatial_mpc.py:98-122,193-217)."""
        coords = reference_path
        if coords is not None and np.shape(coords) != (self.MPC_horizon, 3):
            raise ValueError("reference_path must be %d x 3" % self.MPC_horizon)
        out, status, _ = self._control_solver.solve_tick(coords, float(offset), self.speed_profile_constraints,
                                                         is_localised, qp_max_iter=MAX_SOLVER_ITERATIONS,
                                                         qp_check_every=self._qp_check_every,
                                                         qp_method=0 if self._speed_profile_method == "exact" else 1,
                                                         **from_map)
        self.reference_coordinates = out["coords"]
        path = ReferencePath.adopt(out["table"])   # a view of this tick's own snapshot
        if out["info"][4] == 0.0:
            self.speed_profile = path.velocities.copy()
        else:  # the path keeps the velocities it was built with (spatial_mpc.py:119-122)
            logger.warning("Infeasible speed profile (maximum iterations reached); keeping previous velocities")
        if status != SOLVED:
            logger.warning("Infeasible problem! Failed %d time(s).", self.infeasibility_counter)
            self.infeasibility_counter += 1
            return
        self.projected_control, self.current_prediction, self.cum_time = (out["projected_control"], out["prediction"],
                                                                          out["cum_time"])
        self.times, self.accelerations, self.steer_rates = out["times"], out["accelerations"], out["steer_rates"]
        self.reference_path = path
        self.infeasibility_counter = 0


def published_plan(mpc: SpatialMPC):
    """What the control process publishes after a solve (controller.py:102-108,274-280): `control_inputs` [n, 2]
    = projected_control.T, `control_cumtime` [n], `predicted_locations` [n, 2], each float32 as the shared arrays
    store them (perception/shared_memory.py:98-103).  The returned object is what `TemporalCommandSelector` /
    `TemporalCommandInterpolator` take as their controller."""
    from types import SimpleNamespace
    return SimpleNamespace(control_inputs=np.ascontiguousarray(mpc.projected_control.T, dtype=np.float32),
                           control_cumtime=np.ascontiguousarray(mpc.cum_time, dtype=np.float32),
                           predicted_locations=np.ascontiguousarray(mpc.current_prediction, dtype=np.float32))


def build_mpc(control_config: Dict, vehicle_data) -> SpatialMPC:
    """Same factory as controller.py:19-29."""
    limits = control_config["speed_profile_constraints"]
    model = SpatialBicycleModel(vehicle_data, {"max": limits["v_max"], "min": limits["v_min"]})
    return SpatialMPC(control_config, model)
