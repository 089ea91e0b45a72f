"""Spatial (arc-length indexed) kinematic bicycle model with the reference's interface
(/root/reference/src/acmpc/control/dynamics.py:9-103): Frenet <-> Cartesian transforms and the per-waypoint
linearisation x_{i+1} = A_i x_i + B_i (u_i - u_ref_i) + f_i with x = (e_y, e_psi, t), u = (v, kappa).

The batched rollout of that recurrence over candidate control sequences runs on the GPU
(`csrc/acmpc_device.h: step_spatial`); `linearise` here is the host-side, float64 statement of the same
coefficients, kept for API compatibility and for the tests that pin the C ABI's table preparation.
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import numpy as np

from .reference_path import ReferencePath

TWO_PI = 2.0 * math.pi


def wrap_angle(angle):
    """Wrap to the half-open interval the reference uses (dynamics.py:36)."""
    return np.mod(angle + math.pi, TWO_PI) - math.pi


class SpatialBicycleModel:
    def __init__(self, vehicle_data, velocity_limits: Dict):
        # vehicle_data duck-types ace.steering.SteeringGeometry: .vehicle_data.wheelbase/.width, .max_steering_angle()
        self.length = vehicle_data.vehicle_data.wheelbase
        self.width = vehicle_data.vehicle_data.width
        self.delta_max = vehicle_data.max_steering_angle()
        self.margin = self.width / 2
        self.min_velocity = velocity_limits["min"]
        self.max_velocity = velocity_limits["max"]
        kappa_max = np.tan(self.delta_max) / self.length
        self.min_u = np.array([self.min_velocity, -kappa_max])
        self.max_u = np.array([self.max_velocity, kappa_max])
        self._eps = 1e-12

    def t2s(self, reference_waypoint: np.ndarray, reference_state: np.ndarray) -> np.ndarray:
        """Cartesian pose (x, y, psi) -> Frenet state (e_y, e_psi, t = 0) relative to one waypoint (x, y, psi)."""
        wx, wy, wpsi = reference_waypoint
        px, py, ppsi = reference_state
        lateral = np.cos(wpsi) * (py - wy) - np.sin(wpsi) * (px - wx)
        return np.array([lateral, wrap_angle(ppsi - wpsi), 0.0])

    def s2t(self, reference_waypoints: ReferencePath, reference_states: np.ndarray) -> np.ndarray:
        """Frenet states [n, 3] along a path -> Cartesian [3, n] (x, y, psi)."""
        heading = reference_waypoints.psis
        offset = reference_states[:, 0]
        return np.array([
            reference_waypoints.xs - offset * np.sin(heading),
            reference_waypoints.ys + offset * np.cos(heading),
            heading + reference_states[:, 1],
        ])

    def linearise(self, reference_path: ReferencePath) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """f [n,3], A [n,3,3], B [n,3,2] around (v_ref, kappa_ref) of every waypoint."""
        ds = reference_path.distances
        kappa = reference_path.kappas
        v = reference_path.velocities
        n = len(reference_path)
        inv_vds = 1.0 / (v * ds + self._eps)
        A = np.broadcast_to(np.eye(3), (n, 3, 3)).copy()
        A[:, 0, 1] = ds
        A[:, 1, 0] = -(kappa**2) * ds
        A[:, 2, 0] = -kappa * inv_vds
        B = np.zeros((n, 3, 2))
        B[:, 1, 1] = ds
        B[:, 2, 0] = -1.0 / (v**2 * ds + self._eps)
        f = np.zeros((n, 3))
        f[:, 2] = inv_vds
        return f, A, B
