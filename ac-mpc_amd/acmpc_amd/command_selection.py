"""Picks the control to apply from the last MPC solution by elapsed time - the interface of
/root/reference/src/acmpc/control/commands.py (TemporalCommandSelector is the one the agent uses,
controller.py:110-116)."""
from __future__ import annotations

import numpy as np


class _CommandSource:
    def __init__(self, controller):
        self._controller = controller

    @property
    def _cum_time(self) -> np.ndarray:
        return self._controller.control_cumtime

    def __call__(self, elapsed_time: float) -> np.ndarray:
        return self.get_command(elapsed_time)

    def _offsets(self, elapsed_time: float) -> np.ndarray:
        return self._cum_time - elapsed_time


class TemporalCommandSelector(_CommandSource):
    """Zero-order hold: the latest command whose predicted time is not after `elapsed_time`.

    Quirk kept from the reference (commands.py:30-35): before the first predicted time the index is -1, which
    Python resolves to the LAST command."""

    @property
    def _commands(self) -> np.ndarray:
        return self._controller.control_inputs

    def get_command(self, elapsed_time: float) -> np.ndarray:
        offsets = self._offsets(elapsed_time)
        nearest = int(np.argmin(np.abs(offsets)))
        if offsets[nearest] > 0:
            nearest -= 1
        return self._commands[min(nearest, len(self._commands) - 1)]


class TemporalCommandInterpolator(_CommandSource):
    """Linear interpolation between the two predicted commands that bracket `elapsed_time` (commands.py:41-99)."""

    @property
    def _commands(self) -> np.ndarray:
        return self._controller.control_inputs.T

    def _get_closet_command_index(self, elapsed_time: float):
        offsets = self._offsets(elapsed_time)
        nearest = int(np.argmin(np.abs(offsets)))
        return nearest, offsets[nearest]

    def get_command(self, elapsed_time: float) -> np.ndarray:
        first, offset = self._get_closet_command_index(elapsed_time)
        commands, times = self._commands, self._cum_time
        if first in (0, len(commands) - 1):
            return commands[first]
        second = first + 1 if offset < 0 else first - 1
        span = times[second] - times[first]
        return (commands[first] * ((times[second] - elapsed_time) / span)
                + commands[second] * ((elapsed_time - times[first]) / span))


def steer_target(desired_steering_angle: float, max_steering_angle: float) -> float:
    """What the agent's steering PID is asked to reach for the selected command's steering angle: the angle as a
    fraction of full lock, clipped to [-1, 1], with the simulator's sign convention (positive = left there, right in
    the MPC's frame) - `ElTuarMPC._process_yaw` up to its PID (/root/reference/src/acmpc/agent.py:106-115)."""
    return -1.0 * float(np.clip(desired_steering_angle / max_steering_angle, -1, 1))
