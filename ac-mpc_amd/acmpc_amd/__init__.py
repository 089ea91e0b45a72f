"""acmpc_amd - MI355X-native rollout-and-cost engine behind the ac-mpc controller's Python API.

`build_mpc` / `SpatialMPC` keep the reference's surface (src/acmpc/control/controller.py:19-29,
spatial_mpc.py:20-217); `Engine` is the thin object over the C ABI (include/acmpc.h).
"""
from ._capi import (Engine, EngineError, LAYOUT_CANDIDATE_MAJOR, LAYOUT_STEP_MAJOR, MODE_SPATIAL,  # noqa: F401
                    MODE_TEMPORAL, load_library)
from .bicycle_model import SpatialBicycleModel  # noqa: F401
from .command_selection import TemporalCommandInterpolator, TemporalCommandSelector, steer_target  # noqa: F401
from .mpc import SpatialMPC, build_mpc  # noqa: F401
from .reference_path import ReferencePath  # noqa: F401

__all__ = ["Engine", "EngineError", "load_library", "MODE_SPATIAL", "MODE_TEMPORAL", "LAYOUT_CANDIDATE_MAJOR",
           "LAYOUT_STEP_MAJOR", "build_mpc", "SpatialMPC", "SpatialBicycleModel", "ReferencePath",
           "TemporalCommandSelector", "TemporalCommandInterpolator", "steer_target"]
