"""`ControlSolver` with the reference's seam - `solve(spatial_state, reference_path) -> obj` with `obj.x` laid out
`[x_0 .. x_n ; u_0 .. u_{n-1}]` and `obj.info.status == "solved"` on success
(/root/reference/src/acmpc/control/solvers/control.py:15-24, spatial_mpc.py:191-202) - but instead of assembling a
sparse QP for OSQP it scores batches of candidate control sequences on the GPU: every candidate is rolled through
the same linearised spatial bicycle model the QP has as equality rows, costed with the QP's own P, q and checked
against the QP's own box rows (`csrc/acmpc_device.h`), and the cheapest one wins.

A few refinement rounds (sample around the incumbent, shrink the spread) take the sampled optimum close to the QP
optimum.  Candidates are generated on the device too (Philox counters, smooth perturbations - `acmpc_optimize`), so
one solve is a single host round trip.  The rounds, the candidate count and the spread are build parameters read
from optional config keys (`n_candidates`, `sampling_rounds`, `sampling_sigma`, `sampling_cold_rounds`,
`sampling_cold_sigma`, `sampling_seed`, `w_bound`, `sampling_update`, `softmin_lambda`, `lq_candidate`,
`conformant_sync`).

Round 4: the last round of every solve also holds one deterministic candidate, the LQ plan (`lq_candidate`):
the optimum of the reference's QP without its box rows - a backward Riccati pass over the linearised model
(dynamics.py:65-103, control.py:26-79) - rolled forward with its feedback and clipped into the input box
(csrc/acmpc_lq.h, on the host).  Where no bound is active it IS the QP optimum; the argmin keeps it only when it wins.
Round 5 (`lq_candidate: 2`, the default): where a bound IS active - a control on the input box, a corridor or time row
violated (control.py:47-70,130-144) - that plan is refined against the QP with its box rows by the OSQP splitting with
a Riccati z-update (csrc/acmpc_lq_box.h, on the host, warm-started from tick to tick), and the first solve of a handle
plans with a speed profile solved once on the host.

`rollout_mode: "T"` (default "S") scores the candidates with the Cartesian rollout instead - BASELINE.json north_star's
literal shape: kinematic bicycle (localisation/localiser.py:66-95) advanced by `rollout_dt` seconds per step (0.05),
nearest waypoint of the path at every step (localiser.py:282-289: the nearest of ALL waypoints - `nn_window: null`, the
default up to horizon 107; `nn_window: [back, ahead]` searches only that many waypoints round the previous step's, 0.064 ms
per solve instead of 0.066, and is the default beyond), Frenet errors against it (dynamics.py:23-40), the same weights and
bounds.
The plan is then a TIME-indexed one (control i holds from i * dt), which is what `TemporalCommandSelector` consumes.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Dict

import logging

import numpy as np

from . import _capi
from .reference_path import ReferencePath

logger = logging.getLogger(__name__)

SOLVED = "solved"
# Defaults from tools/sweep_solver_settings.py (600 warm-started solves along the synthetic Silverstone circuit, the
# car up to 1 m off the centre line; plan cost relative to the best setting per pose).  One problem of a few
# thousand candidates leaves most of the chip idle, so candidates are nearly free and rounds are not; and the
# spread has to fit the path: a curvature offset k bends the plan by ~k L^2 / 2 over the L = 150 m the horizon
# covers, so 1e-3 1/m already sweeps the whole corridor.  16 384 x 2 rounds at (0.5 m/s, 1e-3 1/m): +0.15 % in 155 us;
# the first settings of this build, 4 096 x 4 at (3 m/s, 1e-2 1/m): +0.5 % in 187 us.
DEFAULT_CANDIDATES = 16384
DEFAULT_ROUNDS = 2
DEFAULT_SIGMA = (0.5, 1.0e-3)
# A solve without a usable previous plan (the first one, or the one after an infeasible result) explores instead: more
# rounds from a wide spread.  Round 3: 6 rounds from (3 m/s, 5e-2 1/m) - nearly half the curvature box, halved every
# round, each round's candidates spread over eight amplitudes from an eighth of it up.  On the 28 scenarios of the
# reference's own MPC script (tests/test_gpu_qp_gap.py; its "curve" family starts the car at 95 degrees to the path, and
# the QP optimum steers at full lock for 18 steps) the cold plan's excess over the QP optimum fell from 14 % of
# |J_qp| + 1 at worst (4 rounds from (3, 1e-2): the curvature the manoeuvre needs was out of reach) to 0.6 %.
COLD_ROUNDS = 6
COLD_SIGMA = (3.0, 5.0e-2)
INFEASIBLE = "primal infeasible"


class ControlSolver:
    def __init__(self, config: Dict, model):
        self._dynamics_model = model
        self._n_horizon = config["horizon"] - 1
        self._max_iterations = config.get("max_iterations", 4000)
        self._n_candidates = int(config.get("n_candidates", DEFAULT_CANDIDATES))
        self._rounds = int(config.get("sampling_rounds", DEFAULT_ROUNDS))
        self._sigma = np.asarray(config.get("sampling_sigma", DEFAULT_SIGMA), dtype=np.float64)
        self._cold_rounds = int(config.get("sampling_cold_rounds", max(COLD_ROUNDS, self._rounds)))
        self._cold_sigma = np.asarray(config.get("sampling_cold_sigma", np.maximum(COLD_SIGMA, self._sigma)),
                                      dtype=np.float64)
        self._shrink = float(config.get("sampling_shrink", 0.5))     # spread of round r = sigma * shrink**r
        self._explore = True
        self._seed = int(config.get("sampling_seed", 0))
        self._solves = 0
        self._Q = np.asarray(config["step_cost"], dtype=np.float64)
        self._R = np.asarray(config["r_term"], dtype=np.float64)
        self._QN = np.asarray(config["final_cost"], dtype=np.float64)
        # Penalty on the summed squared bound violations.  Round 4: 1e4 (was 1e6).  The reference's QP pins t_0 = 0 while it
        # boxes t >= 0.01 (control.py:134 vs :67) and steps t by 1 / (v ds): on fast, finely sampled paths the first rows of
        # the box cannot be met by ANY plan, OSQP accepts them within eps_abs + eps_rel |z| (~0.09 per row here), and at 1e6 the
        # argmin traded a whole unit of tracking cost for a 1e-6 smaller violation sum (tests/test_gpu_qp_gap.py: hairpin(100),
        # straight(147..200)).  At 1e4 a violation AT that tolerance still costs ~70 - several times any tracking cost.
        self._w_bound = float(config.get("w_bound", 1.0e4))
        self._centre_update = config.get("sampling_update", "argmin")   # or "softmin" (MPPI-style weighted mean)
        self._lambda = float(config.get("softmin_lambda", 1.0))
        # the deterministic candidate of the last round (Engine(lq_candidate=...)), argmin update only: 2 (default) = the LQ
        # plan refined against the QP's box rows where one of them is active (csrc/acmpc_lq_box.h), 1 / True = the LQ plan
        # alone (round 4), 0 / False = none
        self._lq_candidate = int(config.get("lq_candidate", 2)) if self._centre_update == "argmin" else 0
        self._conformant_sync = bool(config.get("conformant_sync", False))
        self._incumbent = None
        self._engine = None  # built on first solve: the input box follows the live velocity limits
        mode = str(config.get("rollout_mode", "S")).upper()
        if mode not in ("S", "T"):
            raise ValueError("rollout_mode must be 'S' or 'T'")
        self.temporal = mode == "T"
        self._dt = float(config.get("rollout_dt", 0.05))
        # None: the nearest of all waypoints (the verified window search).  Absent: that, up to the 100 steps whose search
        # frames fit the round's LDS; a longer horizon gets the (2,5) window, which costs it a third of a scan of every waypoint
        window = config["nn_window"] if "nn_window" in config else (None if self._n_horizon <= 106 else (2, 5))
        self._nn_window = None if window is None else (int(window[0]), int(window[1]))
        if self.temporal and "nn_window" not in config and window is not None:
            # not silently: beyond 106 steps the search frames do not fit the round's LDS, and the nearest of ALL waypoints
            # would be a scan of every waypoint at every step; the (2, 5) window is not exhaustive (it can pick another
            # waypoint where a path folds back on itself) - say which semantics this controller runs with
            logger.warning("rollout_mode T at horizon %d: nearest waypoint searched in the window nn_window=(2, 5) round the "
                           "previous step's (pass nn_window=None for the nearest of all waypoints, or a window of your own)",
                           self._n_horizon + 1)
        self.pose = (0.0, 0.0, np.pi / 2)   # mode T start state: set by SpatialMPC before `solve` (spatial_mpc.py:185)

    # QP input box, widened by 0.1 m/s like the reference (control.py:130-139)
    def _input_box(self):
        model = self._dynamics_model
        return (float(model.min_u[0]) - 0.1, float(model.min_u[1]), float(model.max_u[0]) + 0.1, float(model.max_u[1]))

    def _ensure_engine(self):
        box = self._input_box()
        if self._engine is None or box != self._box_key:
            if self._engine is not None:
                self._engine.close()
            self._box_key = box
            self._box = (np.array(box[:2]), np.array(box[2:]))
            self._engine = _capi.Engine(
                mode=_capi.MODE_TEMPORAL if self.temporal else _capi.MODE_SPATIAL, max_problems=1,
                max_candidates=self._n_candidates, max_steps=self._n_horizon, step_cost=self._Q, r_term=self._R,
                final_cost=self._QN, u_min=self._box[0], u_max=self._box[1], margin=self._dynamics_model.margin,
                wheelbase=self._dynamics_model.length, w_bound=self._w_bound, centre_update=self._centre_update,
                softmin_lambda=self._lambda, dt=self._dt, nn_window=self._nn_window if self.temporal else None,
                lq_candidate=self._lq_candidate)
            if self._conformant_sync:   # (control config key `conformant_sync`: include/acmpc.h, acmpc_set_option)
                self._engine.set_option("ACMPC_CONFORMANT_SYNC", "1")
            if getattr(self, "_map", None) is not None:
                self._engine.bind_map(*self._map)
        return self._engine

    def shift_warm_start(self, elapsed_time: float, cum_time: np.ndarray) -> None:
        """Warm start from the previous plan shifted by the time that has passed since it was made:
        the new step i takes the old plan's controls at `cum_time[i] + elapsed_time`, linearly interpolated between
        the old steps the way `TemporalCommandInterpolator` blends neighbouring commands
        (/root/reference/src/acmpc/control/commands.py:41-66); past the old horizon the last command is held."""
        if self._incumbent is None or elapsed_time <= 0.0:
            return
        cum_time = np.asarray(cum_time, dtype=np.float64)
        if cum_time.shape[0] != self._incumbent.shape[0] or not np.all(np.diff(cum_time) > 0.0):
            return
        at = cum_time + float(elapsed_time)
        self._incumbent = np.stack([np.interp(at, cum_time, self._incumbent[:, k]) for k in range(2)], axis=1)

    def supports_tick(self) -> bool:
        """Whether `solve_tick` (the one-round-trip path, prologue on the device) applies: argmin centre update and a
        horizon the single-workgroup prologue holds."""
        return self._centre_update == "argmin" and self._n_horizon <= 128

    def bind_map(self, centre: np.ndarray, spacing: float):
        """Map centre line the tick may cut its reference path from (`solve_tick(None, ..., map_index=...)`)."""
        self._map = (np.ascontiguousarray(centre, dtype=np.float64), float(spacing))
        if self._engine is not None:
            self._engine.bind_map(*self._map)

    def solve_tick(self, coords, offset: float, constraints: Dict, is_localised: bool,
                   qp_max_iter: int = 4000, qp_check_every: int = 10, map_index: int = -1, pose=(0.0, 0.0),
                   lateral_offset: float = 0.0, centreline_points: int = 500, qp_method: int = 0):
        """One whole tick of `SpatialMPC.get_control` as a single call into the library (`acmpc_control_tick`):
        waypoints, speed profile, Frenet start state, linearisation and the sampling rounds all run on the device
        inside one captured hipGraph.  `coords` is the H x 3 reference path (float64, C-contiguous), `constraints` the
        live speed-profile dict; with `coords=None` the path is cut out of the bound map on the device (window of
        150 m from `map_index`, or from the map point nearest to `pose`).  Returns (outputs dict, status string, total
        rounds); the explore / refine schedule and the acceptance test are those of `solve`."""
        engine = self._ensure_engine()
        n = self._n_horizon
        tick = getattr(self, "_tick", None)
        if tick is None:
            tick = self._tick = _capi.Tick()
            tick.struct_size = _capi.C.sizeof(_capi.Tick)
            tick.horizon = n + 1
            tick.n_candidates = self._n_candidates
            tick.shrink = self._shrink
            tick.qp_eps_abs = tick.qp_eps_rel = 1e-3
        end_velocity = constraints["end_velocity"]
        # (a ctypes field store costs ~0.15 us and there are twenty: the struct is only rewritten when an input changed)
        inputs = (is_localised, end_velocity, offset, constraints["v_min"], constraints["v_max"], constraints["a_min"],
                  constraints["a_max"], constraints["ay_max"], constraints["ki_min"], qp_max_iter, qp_check_every,
                  map_index, centreline_points, pose[0], pose[1], lateral_offset, qp_method)
        if inputs != getattr(self, "_tick_inputs", None):
            self._tick_inputs = inputs
            tick.localised = 1 if is_localised else 0
            tick.has_end_velocity = 0 if end_velocity is None else 1
            tick.end_velocity = 0.0 if end_velocity is None else end_velocity
            tick.offset = offset
            tick.v_min, tick.v_max = constraints["v_min"], constraints["v_max"]
            tick.a_min, tick.a_max = constraints["a_min"], constraints["a_max"]
            tick.ay_max, tick.ki_min = constraints["ay_max"], constraints["ki_min"]
            tick.qp_max_iter, tick.qp_check_every = qp_max_iter, qp_check_every
            tick.qp_method = qp_method
            tick.map_index, tick.centreline_points = map_index, centreline_points
            tick.pose_x, tick.pose_y, tick.lateral_offset = pose[0], pose[1], lateral_offset
        warm = self._incumbent is not None and self._incumbent.shape == (n, 2)
        explore = self._explore or not warm
        total_rounds = 0
        while True:
            rounds, sigma = (self._cold_rounds, self._cold_sigma) if explore else (self._rounds, self._sigma)
            self._solves += 1
            total_rounds += rounds
            tick.rounds = rounds
            tick.sigma[0], tick.sigma[1] = sigma[0], sigma[1]
            tick.seed = self._seed + self._solves
            tick.centre_is_reference = 0 if warm else 1
            centre = self._incumbent if warm else None
            out = engine.control_tick(tick, coords, centre)
            info = out["info"]
            finite = info[7] == 0.0                     # cost, violation and every entry of the plan
            # (a plan with a NaN in it is no warm start: the next solve samples round the reference controls again)
            self._incumbent = out["decision"][3 * (n + 1):].reshape(n, 2) if finite else None
            warm = finite
            tolerance = 1e-3 + 1e-3 * info[3]           # eps_abs + eps_rel * |z|_inf, as in `solve`
            # (the violation sums NaN-ignoring maxima: a path with an infinite coordinate can give 0 next to a NaN cost)
            status = SOLVED if finite and info[1] <= tolerance**2 else INFEASIBLE
            if status == SOLVED or explore:
                break
            explore = True
        self._explore = status != SOLVED
        return out, status, total_rounds

    def solve(self, spatial_state: np.ndarray, reference_path: ReferencePath) -> SimpleNamespace:
        engine = self._ensure_engine()
        n = self._n_horizon
        engine.set_paths(reference_path.table)
        lo, hi = self._box
        u_ref = np.empty((n, 2))
        np.clip(reference_path.velocities, lo[0], hi[0], out=u_ref[:, 0])
        np.clip(reference_path.kappas, lo[1], hi[1], out=u_ref[:, 1])
        warm = self._incumbent is not None and self._incumbent.shape == u_ref.shape
        explore = self._explore or not warm
        # mode T rolls the pose itself (the Frenet state the reference's seam hands over is not used then)
        x0 = np.asarray(self.pose if self.temporal else spatial_state, dtype=np.float32)[None]
        total_rounds = 0
        while True:
            centre = self._incumbent if warm else u_ref
            rounds, sigma = (self._cold_rounds, self._cold_sigma) if explore else (self._rounds, self._sigma)
            self._solves += 1
            total_rounds += rounds
            # sample -> rollout + cost -> argmin, `rounds` times, entirely on the device (acmpc_optimize)
            best = engine.optimize(x0, centre[None], u_ref[None], self._n_candidates, rounds, sigma, shrink=self._shrink,
                                   seed=self._seed + self._solves)
            u_star = best["u"][0].astype(np.float64)
            x_star = best["x"][0].astype(np.float64)
            finite = bool(np.isfinite(u_star).all() and np.isfinite(x_star).all() and np.isfinite(best["cost"][0])
                          and np.isfinite(best["violation"][0]))
            self._incumbent, warm = (u_star, True) if finite else (None, False)
            # accept a residual bound violation the way the reference's solver does: every row within
            # eps_abs + eps_rel * |z| (OSQP's test is the infinity norm over the rows).  `violation` is the SUM of the
            # squared row excesses, so `violation <= tolerance^2` bounds every single row by the tolerance
            # (conservative: several rows violated at once are rejected a little earlier than OSQP would).
            tolerance = 1e-3 + 1e-3 * max(np.abs(x_star).max(), np.abs(u_star).max())
            violation = float(best["violation"][0])
            status = SOLVED if finite and violation <= tolerance**2 else INFEASIBLE
            if status == SOLVED or explore:
                break
            explore = True    # the refining schedule failed (the path jumped under the old plan): explore once, now
        self._explore = status != SOLVED      # refine from a good plan, explore again after a bad one
        info = SimpleNamespace(status=status, obj_val=float(best["cost"][0]), violation=violation,
                               n_feasible=int(best["n_feasible"][0]), iter=total_rounds)
        return SimpleNamespace(x=np.concatenate([x_star.ravel(), u_star.ravel()]), info=info)
