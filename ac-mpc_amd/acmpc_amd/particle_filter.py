"""GPU scoring for the particle-filter localiser, with the seam of the reference's
`LocalisationProcess._update_particles` (/root/reference/src/acmpc/localisation/localiser.py:255-265): particle
states + downsampled track-limit observations in, the reference's `particles` dict out
(`track_indices`, `centreline_idx`, `minimum_offset`, `heading_offset`, `observation_error`, `score`).

The data-parallel work - three nearest-point queries per particle against the map (scipy KD-trees in the
reference), observation placement, error, score, validity - runs in `csrc/acmpc_pf.hip`; the sequential, random
resampling (localiser.py:412-545) is left to the caller's NumPy code, exactly as in the reference.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List

import numpy as np

from . import _capi


class ParticleScorer:
    """`config` is the reference's `localisation` YAML block (configs/monza.yaml:43-66); `track_map` the dict of
    `utils/load.py:9-35` with "centre", "left", "right" polylines."""

    def __init__(self, config: Dict, track_map: Dict[str, np.ndarray], wheelbase: float = 2.65,
                 max_observation_points: int = 2048, device: int = -1):
        lib = _capi.load_library()
        self._lib = lib
        thresholds = config["thresholds"]
        p = _capi.PfParams()
        p.struct_size = C.sizeof(_capi.PfParams)
        p.device = device
        p.max_particles = int(config["n_particles"])
        p.max_observation_points = int(max_observation_points)
        p.score_mean = float(config["score_distribution"]["mean"])
        p.score_sigma = float(config["score_distribution"]["sigma"])
        p.threshold_rotation = float(thresholds["rotation"]) * np.pi / 180
        p.threshold_offset = float(thresholds["offset"])
        p.threshold_error = float(thresholds["track_limit"])
        p.wheelbase = float(wheelbase)
        self._tracks = [np.ascontiguousarray(track_map[k], dtype=np.float64) for k in ("centre", "left", "right")]
        centre = self._tracks[0]
        self._average_distance_between_map_points = float(np.mean(np.linalg.norm(centre[1:] - centre[:-1], axis=1)))
        self._handle = _capi._CTX()
        args = []
        for t in self._tracks:
            args += [t.ctypes.data_as(_capi._F64P), t.shape[0]]
        rc = lib.acmpc_pf_create(C.byref(p), *args, C.byref(self._handle))
        if rc != _capi.OK:
            raise _capi.EngineError(rc, (lib.acmpc_pf_last_error(None) or b"").decode())
        self.max_particles = p.max_particles
        self.scale = float(lib.acmpc_pf_score_scale(self._handle))

    def _check(self, rc: int):
        if rc != _capi.OK:
            raise _capi.EngineError(rc, (self._lib.acmpc_pf_last_error(self._handle) or b"").decode())

    def close(self):
        if getattr(self, "_handle", None) is not None and self._handle.value:
            self._lib.acmpc_pf_destroy(self._handle)
            self._handle = _capi._CTX()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- host-side preprocessing, as the reference does it ---------------------------------------------------
    def downsample_observations(self, observations: Dict[str, np.ndarray]) -> List[np.ndarray]:
        """Thin both observed limits to the map's point spacing (localiser.py:241-253)."""
        return [self._downsample(observations["left"]), self._downsample(observations["right"])]

    def _downsample(self, observation: np.ndarray) -> np.ndarray:
        spacing = np.mean(np.linalg.norm(observation[1:] - observation[:-1], axis=1))
        n_points = len(observation) * (spacing / self._average_distance_between_map_points)
        keep = np.zeros(len(observation), dtype=np.bool_)
        keep[np.linspace(0, len(observation) - 1, int(n_points), dtype=np.uint16)] = True
        return observation[keep]

    # -- the seam ----------------------------------------------------------------------------------------------
    def update_particles(self, states: np.ndarray, observations: List[np.ndarray]) -> Dict:
        """states [P,3] (x, y, yaw); observations = [left, right] downsampled limits in the vehicle frame."""
        states = np.ascontiguousarray(states, dtype=np.float32)
        left, right = (np.ascontiguousarray(o[o[:, 1] < 50], dtype=np.float32) for o in observations)  # :336-337
        P = states.shape[0]
        idx = np.empty((P, 3), dtype=np.int32)
        offset, heading, error, score = (np.empty(P) for _ in range(4))
        valid = np.empty(P, dtype=np.uint8)
        f32, f64 = _capi._F32P, _capi._F64P
        self._check(self._lib.acmpc_pf_score(
            self._handle, states.ctypes.data_as(f32), P, left.ctypes.data_as(f32), left.shape[0],
            right.ctypes.data_as(f32), right.shape[0], idx.ctypes.data_as(_capi._I32P), offset.ctypes.data_as(f64),
            heading.ctypes.data_as(f64), error.ctypes.data_as(f64), score.ctypes.data_as(f64),
            valid.ctypes.data_as(C.POINTER(C.c_uint8))))
        return {"states": states, "track_indices": idx.astype(np.int64), "centreline_idx": idx[:, 0].astype(np.int64),
                "minimum_offset": offset, "heading_offset": heading, "observation_error": error, "score": score,
                "valid_mask": valid.astype(bool)}

    def advance_particles(self, states: np.ndarray, delta: np.ndarray, velocity: np.ndarray, dt: float) -> np.ndarray:
        """Kinematic step of every particle with its own (noisy) steering angle and speed (localiser.py:66-95)."""
        out = np.ascontiguousarray(states, dtype=np.float32).copy()
        delta = np.ascontiguousarray(delta, dtype=np.float32)
        velocity = np.ascontiguousarray(velocity, dtype=np.float32)
        f32 = _capi._F32P
        self._check(self._lib.acmpc_pf_advance(self._handle, out.ctypes.data_as(f32), delta.ctypes.data_as(f32),
                                               velocity.ctypes.data_as(f32), out.shape[0], float(dt)))
        return out

    def estimate_location(self, scores: np.ndarray, states: np.ndarray):
        """(estimate [3], max distance, max |yaw difference|) - localiser.py:561-579."""
        states = np.ascontiguousarray(states, dtype=np.float32)
        scores = np.ascontiguousarray(scores, dtype=np.float32)
        est = np.empty(3)
        md, ma = C.c_double(0), C.c_double(0)
        self._check(self._lib.acmpc_pf_estimate(self._handle, states.ctypes.data_as(_capi._F32P),
                                                scores.ctypes.data_as(_capi._F32P), states.shape[0],
                                                est.ctypes.data_as(_capi._F64P), C.byref(md), C.byref(ma)))
        return est, md.value, ma.value
