"""Particle scoring on the GPU against vectors produced by the reference's own LocalisationProcess methods
(tests/golden/gen_golden.py, G12) and against the oracle on a second, seeded input."""
import numpy as np
import pytest

import acmpc_oracle as orc

pytestmark = pytest.mark.gpu

LOCALISATION = dict(n_particles=500, score_distribution=dict(mean=0, sigma=10),     # configs/monza.yaml:43-66
                    thresholds=dict(offset=10, rotation=90, minimum_particles=20, track_limit=20.0))


def _scorer(golden, **extra):
    from acmpc_amd.particle_filter import ParticleScorer
    track_map = dict(centre=golden["pf/centre"], left=golden["pf/left"], right=golden["pf/right"])
    return ParticleScorer(dict(LOCALISATION, **extra), track_map)


def test_scoring_matches_the_reference(golden):
    g = golden
    scorer = _scorer(g)
    assert abs(scorer.scale - float(g["pf/scale"])) < 1e-17
    obs = scorer.downsample_observations({"left": g["pf/obs_left_raw"], "right": g["pf/obs_right_raw"]})
    np.testing.assert_array_equal(obs[0], g["pf/obs_left_downsampled"])
    np.testing.assert_array_equal(obs[1], g["pf/obs_right_downsampled"])
    out = scorer.update_particles(g["pf/states"], obs)
    np.testing.assert_array_equal(out["track_indices"], g["pf/track_indices"])       # KD-tree answers, exactly
    np.testing.assert_allclose(out["minimum_offset"], g["pf/minimum_offset"], rtol=1e-12)
    np.testing.assert_allclose(out["heading_offset"], g["pf/heading_offset"], rtol=0, atol=1e-12)
    # placement happens in float32 in the reference and here; cosf/sinf may differ in the last bit
    np.testing.assert_allclose(out["observation_error"], g["pf/observation_error"], rtol=1e-5)
    np.testing.assert_allclose(out["score"], g["pf/score"], rtol=1e-5)
    np.testing.assert_array_equal(out["valid_mask"], g["pf/valid_mask"])


def test_scoring_matches_the_oracle_on_other_inputs(golden):
    g = golden
    scorer = _scorer(g)
    rng = np.random.default_rng(5)
    centre = g["pf/centre"]
    P = 500
    seeds = rng.integers(0, len(centre), P)
    states = np.concatenate([centre[seeds] + rng.normal(0, 4.0, (P, 2)), rng.uniform(-np.pi, np.pi, (P, 1))],
                            axis=1).astype(np.float32)
    left = np.stack([-4.5 + rng.normal(0, 0.2, 90), np.linspace(0, 60, 90)], axis=1).astype(np.float32)   # some y >= 50
    right = np.stack([4.5 + rng.normal(0, 0.2, 70), np.linspace(0, 45, 70)], axis=1).astype(np.float32)
    out = scorer.update_particles(states, [left, right])
    want = orc.pf_score_particles(states, g["pf/centre"], g["pf/left"], g["pf/right"], left, right, 0, 10,
                                  dict(rotation=np.pi / 2, offset=10, track_limit=20.0))
    np.testing.assert_array_equal(out["track_indices"], want["track_indices"])
    np.testing.assert_allclose(out["minimum_offset"], want["minimum_offset"], rtol=1e-12)
    np.testing.assert_allclose(out["heading_offset"], want["heading_offset"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(out["observation_error"], want["observation_error"], rtol=1e-5)
    np.testing.assert_allclose(out["score"], want["score"], rtol=1e-5)
    np.testing.assert_array_equal(out["valid_mask"], want["valid"])


def test_advance_and_estimate_match_the_reference(golden):
    g = golden
    scorer = _scorer(g)
    # kinematic step: x_dot from the reference (G7) times dt
    dt = 0.0123
    states, delta, velocity = g["xdot/states"], g["xdot/delta"], g["xdot/velocity"]
    got = scorer.advance_particles(states, delta, velocity, dt)
    want = states + g["xdot/out"] * np.float32(dt)
    np.testing.assert_allclose(got, want, rtol=2e-6, atol=2e-5)
    # weighted mean (G9) incl. the NaN fallback, and the convergence numbers
    scores, st = g["est/scores"], g["est/states"]
    est, max_d, max_a = scorer.estimate_location(scores, st)
    np.testing.assert_allclose(est, g["est/out"], rtol=1e-5)
    want_est, _ = orc.pf_convergence(scores.astype(np.float64), st.astype(np.float64), 50, np.pi / 2)
    np.testing.assert_allclose(max_d, np.linalg.norm(st[:, :2] - want_est[:2], axis=1).max(), rtol=1e-6)
    np.testing.assert_allclose(max_a, np.abs(st[:, 2] - want_est[2]).max(), rtol=1e-6)
    est, _, _ = scorer.estimate_location(np.zeros_like(scores), st)
    np.testing.assert_allclose(est, g["est/out_nan_fallback"], rtol=1e-5)


def test_capacity_errors(golden):
    from acmpc_amd import EngineError
    scorer = _scorer(golden, n_particles=10)
    with pytest.raises(EngineError) as e:
        scorer.update_particles(golden["pf/states"], [golden["pf/obs_left_downsampled"], golden["pf/obs_right_downsampled"]])
    assert e.value.code == -4


def test_filter_cycle_tracks_a_moving_car(golden):
    """`ParticleFilter` end to end (advance on the GPU with noisy controls, score on the GPU, resample, convergence
    flag - localiser.py:41-77,234-239): from a cluster round the true pose the weighted-mean estimate must follow a
    car driving along the centre line.  Functional, not parity: the random part is pinned on the CPU
    (tests/test_host_mirror.py) and the scoring above."""
    from acmpc_amd.particle_filter import ParticleFilter
    g = golden
    centre, left, right = g["pf/centre"], g["pf/left"], g["pf/right"]
    M = len(centre)
    cfg = dict(LOCALISATION, n_particles=300, n_converged_particles=300, sampling_noise=dict(x=1.1, y=1.1, yaw=3.0),
               control_noise=dict(velocity=0.25, yaw=2.0), convergence_criteria=dict(maximum_distance=50, maximum_angle=90))
    rng = np.random.default_rng(11)
    pf = ParticleFilter(cfg, dict(centre=centre, left=left, right=right), wheelbase=2.65, rng=rng)
    tangent = np.roll(centre, -1, axis=0) - np.roll(centre, 1, axis=0)
    heading = np.unwrap(np.arctan2(tangent[:, 1], tangent[:, 0]))
    spacing = float(np.mean(np.linalg.norm(np.diff(centre, axis=0), axis=1)))
    idx, step, dt = 700, 3, 0.05
    speed = step * spacing / dt                                             # the truth moves whole map points
    pose = np.array([centre[idx, 0], centre[idx, 1], heading[idx]])
    pf.states = (pose + rng.normal(0, [1.0, 1.0, 0.05], (300, 3))).astype(np.float32)
    pf.scores = np.full(300, 1.0 / 300, dtype=np.float32)

    def observe(track, count, at, yaw):
        pts = track[(at + np.arange(count)) % M] - centre[at]
        a = np.pi / 2 - yaw
        rot = np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]])
        return (pts @ rot.T + rng.normal(0, 0.15, (count, 2))).astype(np.float32)

    errors = []
    for tick in range(40):
        yaw_rate = (heading[(idx + step) % M] - heading[idx]) / dt
        tyre_angle = float(np.arctan(yaw_rate * 2.65 / speed))               # kinematic bicycle, localiser.py:94
        pf.step(tyre_angle, speed, dt)
        idx = (idx + step) % M
        particles = pf.update({"left": observe(left, 230, idx, heading[idx]),
                               "right": observe(right, 210, idx, heading[idx])})
        assert particles["score"].shape == (300,) and pf.states.shape == (300, 3) and pf.states.dtype == np.float32
        assert not pf.was_reset
        errors.append(np.linalg.norm(pf.estimated_location[:2] - centre[idx]))
    assert pf.is_converged
    assert max(errors) < 3.0 and np.mean(errors[20:]) < 2.0, (max(errors), np.mean(errors[20:]))

