"""acmpc_amd - MI355X-native rollout-and-cost engine behind the ac-mpc controller's Python API."""
from ._capi import (Engine, EngineError, LAYOUT_CANDIDATE_MAJOR, LAYOUT_STEP_MAJOR, MODE_SPATIAL,  # noqa: F401
                    MODE_TEMPORAL, load_library)

__all__ = ["Engine", "EngineError", "load_library", "MODE_SPATIAL", "MODE_TEMPORAL", "LAYOUT_CANDIDATE_MAJOR",
           "LAYOUT_STEP_MAJOR"]
