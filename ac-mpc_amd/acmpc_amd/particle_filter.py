"""GPU scoring for the particle-filter localiser, with the seam of the reference's
`LocalisationProcess._update_particles` (/root/reference/src/acmpc/localisation/localiser.py:255-265): particle
states + downsampled track-limit observations in, the reference's `particles` dict out
(`track_indices`, `centreline_idx`, `minimum_offset`, `heading_offset`, `observation_error`, `score`).

The data-parallel work - three nearest-point queries per particle against the map (scipy KD-trees in the
reference), observation placement, error, score, validity - runs in `csrc/acmpc_pf.hip`.  `ParticleFilter` adds
the sequential, random part round it (resampling, reset, convergence flag: localiser.py:420-570) in NumPy with
the reference's draw order, so that a run seeded like the reference reproduces it.
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
        self._check(self._lib.acmpc_pf_score(
            self._handle, states.ctypes.data, P, left.ctypes.data, left.shape[0], right.ctypes.data, right.shape[0],
            idx.ctypes.data, offset.ctypes.data, heading.ctypes.data, error.ctypes.data, score.ctypes.data,
            valid.ctypes.data))
        return {"states": states, "track_indices": idx.astype(np.int64), "centreline_idx": idx[:, 0].astype(np.int64),
                "minimum_offset": offset, "heading_offset": heading, "observation_error": error, "score": score,
                "valid_mask": valid.astype(bool)}

    def advance_particles(self, states: np.ndarray, delta: np.ndarray, velocity: np.ndarray, dt: float) -> np.ndarray:
        """Kinematic step of every particle with its own (noisy) steering angle and speed (localiser.py:66-95)."""
        out = np.ascontiguousarray(states, dtype=np.float32).copy()
        delta = np.ascontiguousarray(delta, dtype=np.float32)
        velocity = np.ascontiguousarray(velocity, dtype=np.float32)
        self._check(self._lib.acmpc_pf_advance(self._handle, out.ctypes.data, delta.ctypes.data, velocity.ctypes.data,
                                               out.shape[0], float(dt)))
        return out

    def estimate_location(self, scores: np.ndarray, states: np.ndarray):
        """(estimate [3], max distance, max |yaw difference|) - localiser.py:561-579."""
        states = np.ascontiguousarray(states, dtype=np.float32)
        scores = np.ascontiguousarray(scores, dtype=np.float32)
        est = np.empty(3)
        md, ma = C.c_double(0), C.c_double(0)
        self._check(self._lib.acmpc_pf_estimate(self._handle, states.ctypes.data, scores.ctypes.data, states.shape[0],
                                                est.ctypes.data, C.byref(md), C.byref(ma)))
        return est, md.value, ma.value


class ParticleFilter:
    """The filter round the scorer: state arrays + the reference's update cycle
    (`Localiser.step` -> `_score_particles`: advance, score, resample, convergence flag).

    `rng` is anything with NumPy's `normal` / `choice` (default: the `np.random` module = the global stream the
    reference draws from).  Live particles are the rows of `states` / `scores` (float32, what the reference keeps
    in its shared arrays for scores > 0)."""

    def __init__(self, config: Dict, track_map: Dict[str, np.ndarray], wheelbase: float = 2.65, rng=np.random,
                 **scorer_kwargs):
        self.scorer = ParticleScorer(config, track_map, wheelbase=wheelbase, **scorer_kwargs)
        self._centre = np.asarray(track_map["centre"], dtype=np.float64)
        self._rng = rng
        self._max_n_particles = int(config["n_particles"])
        self._n_converged_particles = int(config["n_converged_particles"])
        self._minimum_particles = int(config["thresholds"]["minimum_particles"])
        noise = config["sampling_noise"]
        self._sampling_sigma = (float(noise["x"]), float(noise["y"]), float(noise["yaw"]) * np.pi / 180)
        control = config["control_noise"]
        self._control_sigma = (float(control["yaw"]) * np.pi / 180, float(control["velocity"]))
        criteria = config["convergence_criteria"]
        self._convergence_distance = float(criteria["maximum_distance"])
        self._convergence_angle = float(criteria["maximum_angle"])     # compared with radians, as the reference does
        self.is_converged = False
        self.was_reset = False
        self.reset()

    def reset(self):
        """Spread the particles evenly along the centre line, heading along it (localiser.py:468-485)."""
        idx = np.linspace(0, len(self._centre) - 3, self._max_n_particles).astype(np.int32)
        step = self._centre[idx + 1] - self._centre[idx]
        self.states = np.column_stack([self._centre[idx], np.arctan2(step[:, 1], step[:, 0])]).astype(np.float32)
        scores = np.ones(self._max_n_particles, dtype=np.float32)
        self.scores = scores / np.sum(scores)
        self.is_converged = False

    def step(self, tyre_angle: float, velocity: float, dt: float):
        """Move every particle with its own noisy control (localiser.py:41-77): yaw noise on the tyre angle,
        |velocity + noise|; the kinematic step itself runs on the GPU."""
        n = self.states.shape[0]
        delta = tyre_angle + self._rng.normal(0, self._control_sigma[0], n)
        speed = np.abs(velocity + self._rng.normal(0, self._control_sigma[1], n))
        self.states = self.scorer.advance_particles(self.states, delta, speed, dt)

    def resample(self, particles: Dict):
        """Drop invalid particles, reset if too few are left, otherwise top up from the valid ones in proportion
        to their score (localiser.py:420-545)."""
        valid = particles["valid_mask"]
        states, scores, score = self.states[valid], self.scores[valid], particles["score"][valid]
        n_valid = states.shape[0]
        self.was_reset = n_valid < self._minimum_particles
        if self.was_reset:
            self.reset()
            return
        desired = self._n_converged_particles if self.is_converged else self._max_n_particles
        n_new = max(0, desired - n_valid)
        noise = np.array([self._rng.normal(0, sigma, n_new) for sigma in self._sampling_sigma]).T
        with np.errstate(all="ignore"):
            weights = score / np.sum(score)
        if np.isnan(weights).any():
            weights = np.ones(n_valid) / n_valid
        picked = self._rng.choice(n_valid, size=n_new, p=weights)
        self.states = np.concatenate((states, states[picked] + noise), axis=0).astype(np.float32)
        self.scores = np.concatenate((scores, scores[picked]), axis=0).astype(np.float32)

    def update(self, observations: Dict[str, np.ndarray]) -> Dict:
        """One `_score_particles` (localiser.py:234-239): downsample, score on the GPU, publish the scores,
        resample, refresh the convergence flag.  Returns the `particles` dict of the scoring."""
        particles = self.scorer.update_particles(self.states, self.scorer.downsample_observations(observations))
        self.scores = particles["score"].astype(np.float32)                        # _update_particle_scores
        self.resample(particles)
        self.update_is_converged_flag()
        return particles

    def update_is_converged_flag(self):
        _, max_distance, max_angle = self.scorer.estimate_location(self.scores, self.states)
        self.is_converged = bool(max_distance < self._convergence_distance and max_angle < self._convergence_angle)

    @property
    def estimated_location(self) -> np.ndarray:
        return self.scorer.estimate_location(self.scores, self.states)[0]



class DeviceParticleFilter:
    """The same update cycle with the particles LIVING on the GPU: `step` and `update` are one launch sequence each and
    `update` is one host round trip (the observation goes up, eight numbers come back) instead of the two of
    `ParticleFilter` (score down, resample on the host, estimate up and down).  The random draws are counter-based
    (Philox4x32-10, key = `seed`, counter = update / step number) and made on the device - not NumPy's global stream,
    whose call order a kernel cannot follow - so this class reproduces `ParticleFilter` statistically, not draw for
    draw; `ParticleFilter` stays the mode that is pinned to the reference's own resampling under a shared seed, this one
    is pinned by the oracle's restatement of its draws (oracle `pf_resample_counter_based`)."""

    def __init__(self, config: Dict, track_map: Dict[str, np.ndarray], wheelbase: float = 2.65, seed: int = 0,
                 **scorer_kwargs):
        self.scorer = ParticleScorer(config, track_map, wheelbase=wheelbase, **scorer_kwargs)
        self._lib, self._handle = self.scorer._lib, self.scorer._handle
        self._max_n_particles = int(config["n_particles"])
        self._n_converged_particles = int(config["n_converged_particles"])
        noise, control, criteria = config["sampling_noise"], config["control_noise"], config["convergence_criteria"]
        self._control_sigma = (float(control["yaw"]) * np.pi / 180, float(control["velocity"]))
        self._convergence_distance = float(criteria["maximum_distance"])
        self._convergence_angle = float(criteria["maximum_angle"])
        self._rs = _capi.PfResample()
        self._rs.struct_size = C.sizeof(_capi.PfResample)
        self._rs.minimum_particles = int(config["thresholds"]["minimum_particles"])
        self._rs.seed = int(seed)
        self._rs.sigma_x, self._rs.sigma_y = float(noise["x"]), float(noise["y"])
        self._rs.sigma_yaw = float(noise["yaw"]) * np.pi / 180
        self._seed, self._steps, self._updates = int(seed), 0, 0
        self._result = np.zeros(8)
        self.is_converged = False
        self.was_reset = False
        self.reset()

    def _check(self, rc: int):
        self.scorer._check(rc)

    def reset(self):
        self._check(self._lib.acmpc_pf_filter_reset(self._handle, self._max_n_particles))
        self.is_converged = False

    def set_particles(self, states: np.ndarray, scores: np.ndarray):
        states = np.ascontiguousarray(states, dtype=np.float32)
        scores = np.ascontiguousarray(scores, dtype=np.float32)
        self._check(self._lib.acmpc_pf_filter_set(self._handle, states.ctypes.data, scores.ctypes.data, states.shape[0]))

    def particles(self):
        """(states [n, 3], scores [n]) downloaded from the device."""
        states = np.empty((self._max_n_particles, 3), dtype=np.float32)
        scores = np.empty(self._max_n_particles, dtype=np.float32)
        n = C.c_int32(0)
        self._check(self._lib.acmpc_pf_filter_get(self._handle, states.ctypes.data, scores.ctypes.data,
                                                  self._max_n_particles, C.byref(n)))
        return states[:n.value].copy(), scores[:n.value].copy()

    @property
    def states(self) -> np.ndarray:
        return self.particles()[0]

    @property
    def scores(self) -> np.ndarray:
        return self.particles()[1]

    def step(self, tyre_angle: float, velocity: float, dt: float):
        self._steps += 1
        self._check(self._lib.acmpc_pf_filter_step(self._handle, float(tyre_angle), float(velocity), float(dt),
                                                   self._control_sigma[0], self._control_sigma[1], self._seed,
                                                   self._steps))

    def update(self, observations: Dict[str, np.ndarray]) -> Dict:
        left, right = (np.ascontiguousarray(o[o[:, 1] < 50], dtype=np.float32)
                       for o in self.scorer.downsample_observations(observations))
        self._updates += 1
        self._rs.counter = self._updates
        self._rs.n_desired = self._n_converged_particles if self.is_converged else self._max_n_particles
        self._check(self._lib.acmpc_pf_filter_update(self._handle, left.ctypes.data, left.shape[0], right.ctypes.data,
                                                     right.shape[0], C.byref(self._rs), self._result.ctypes.data))
        r = self._result
        self.was_reset = bool(r[7])
        self.is_converged = bool(r[3] < self._convergence_distance and r[4] < self._convergence_angle)
        return dict(estimate=r[:3].copy(), max_distance=float(r[3]), max_angle=float(r[4]), n_particles=int(r[5]),
                    n_valid=int(r[6]), was_reset=self.was_reset)

    @property
    def estimated_location(self) -> np.ndarray:
        return self._result[:3].copy()
