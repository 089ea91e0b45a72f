"""Curvature-limited reference speed for every waypoint - the interface of the reference's
`SpeedProfileSolver` / `LocalisedSpeedProfileSolver` (/root/reference/src/acmpc/control/solvers/speed_profile.py).

The problem is the reference's:  min 1/2 |v|^2 - v_hi'v  subject to
    a_min <= (v[i+1] - v[i]) / (2 ds[i]) <= a_max        (speed_profile.py:47-51)
    v_min <= v <= v_hi                                    (speed_profile.py:45)
with v_hi the curvature-limited ceiling sqrt(ay_max / |kappa|) clipped to [v_min, v_max] plus 2 m/s, and the
last entry forced to the end velocity (speed_profile.py:26-43).  It is solved on the host by the library instead of the
`osqp` package: EXACTLY, in two passes, wherever the problem is feasible (`acmpc_speed_profile_exact`: the objective is
1/2 |v - v_hi|^2 and v_hi is the upper bound, so the optimum is the pointwise largest feasible profile - config key
`method: "exact"`, the default), and by the native tridiagonal ADMM that restates OSQP's iteration
(`acmpc_speed_profile_qp`, O(n) per iteration - also for the 10^4-waypoint lap profile) otherwise or with
`method: "admm"`.  The tick's device prologue makes the same choice with the same arithmetic (acmpc_tick.qp_method).  `constraints` is held by reference: the control process rewrites its "v_max"
every tick (controller.py:241-243) and the next solve must see it.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Dict, Optional

import numpy as np

from . import _capi
from .reference_path import ReferencePath


class SpeedProfileSolver:
    def __init__(self, config: Dict):
        self._n_horizon = config["control_horizon"]
        self._max_iterations = config["max_iterations"]
        self._constraints = config["constraints"]
        self._check_every = int(config.get("check_every", 10))   # the stopping test runs every this many iterations
        self._method = str(config.get("method", "exact"))
        if self._method not in ("exact", "admm"):
            raise ValueError("speed profile method must be 'exact' or 'admm', not %r" % (self._method,))
        self._eps = 1e-12
        self._warm = None

    # -- the QP's data ------------------------------------------------------------------------------------
    _localised = False

    def velocity_ceiling(self, reference_path: ReferencePath, end_velocity: Optional[float]) -> np.ndarray:
        c = self._constraints
        return _capi.velocity_ceiling(reference_path.kappas, c["ay_max"], c["ki_min"], c["v_min"], c["v_max"],
                                      self._localised, end_velocity)

    def velocity_ceiling_numpy(self, reference_path: ReferencePath, end_velocity: Optional[float]) -> np.ndarray:
        """The NumPy statement of `velocity_ceiling` (speed_profile.py:26-43), kept for the tests."""
        c = self._constraints
        curvature = np.abs(reference_path.kappas)
        ceiling = np.sqrt(c["ay_max"] / (curvature + self._eps))
        ceiling[curvature < c["ki_min"]] = c["v_max"]
        ceiling = np.maximum(c["v_min"], np.minimum(ceiling, c["v_max"])) + 2.0
        if end_velocity is not None:
            ceiling[-1] = end_velocity
        return ceiling

    def problem(self, reference_path: ReferencePath, end_velocity: Optional[float] = None) -> Dict:
        """P (diagonal), q, A, l, u of the speed-profile QP."""
        n = self._n_horizon
        c = self._constraints
        ceiling = self.velocity_ceiling(reference_path, end_velocity)
        gain = 1.0 / (2.0 * reference_path.distances[:-1])
        A = np.zeros((2 * n - 1, n))
        rows = np.arange(n - 1)
        A[rows, rows] = -gain
        A[rows, rows + 1] = gain
        A[n - 1 + np.arange(n), np.arange(n)] = 1.0
        lower = np.concatenate([np.full(n - 1, float(c["a_min"])), np.full(n, float(c["v_min"]))])
        upper = np.concatenate([np.full(n - 1, float(c["a_max"])), ceiling])
        return dict(P_diag=np.ones(n), q=-ceiling, A=A, l=lower, u=upper, v_hi=ceiling)

    def solve(self, reference_path: ReferencePath, end_velocity: Optional[float] = None) -> SimpleNamespace:
        """Result shaped like osqp's: `.x`, `.y`, `.info.status` ("solved" on success), `.info.iter`."""
        c = self._constraints
        ceiling = self.velocity_ceiling(reference_path, end_velocity)
        if self._method == "exact":
            swept = _capi.speed_profile_exact(ceiling, reference_path.distances, c["a_min"], c["a_max"], c["v_min"])
            if swept is not None:
                self._warm = swept          # (what the device keeps as its iterate too: the optimum, no multipliers)
                return SimpleNamespace(x=swept[0], y=swept[1], info=SimpleNamespace(status="solved", iter=0))
        warm = self._warm if self._warm is not None and self._warm[0].shape == ceiling.shape else None
        x, y, status, iters = _capi.speed_profile_qp(ceiling, reference_path.distances, c["a_min"], c["a_max"],
                                                     c["v_min"], max_iter=self._max_iterations, warm=warm,
                                                     check_every=self._check_every)
        if status == "solved":
            self._warm = (x, y)
        return SimpleNamespace(x=x, y=y, info=SimpleNamespace(status=status, iter=iters))


class LocalisedSpeedProfileSolver(SpeedProfileSolver):
    """Once localised the ceiling is the map-derived reference speed itself (speed_profile.py:131-150)."""

    _localised = True

    def velocity_ceiling_numpy(self, reference_path: ReferencePath, end_velocity: Optional[float]) -> np.ndarray:
        return np.full(self._n_horizon, float(self._constraints["v_max"]))
