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


@pytest.mark.parametrize("P,points", [(4096, (90, 70)), (9001, (90, 70)), (70001, (90, 70)), (4099, (300, 333)), (66000, (257, 1))])
def test_grid_search_equals_the_exhaustive_scan(golden, P, points, monkeypatch):
    """The nearest map points come from a grid search (pf_nearest_kernel) - the exhaustive scan's answer, bit for bit:
    particles on the track, metres off it, far outside the map's box, exactly on cell borders of the grid, duplicates,
    and a non-finite one.  Behind the grid search the scoring runs on a wavefront per 4 or 16 particles
    (pf_score_given_kernel; 70 001: the sixteen-particle form with a ragged last wave) - the same bits as the workgroup
    kernel that scans for itself, and as the workgroup kernel behind the grid search (ACMPC_PF_WORKGROUP_SCORE); with more
    than 256 observation points a lane sums several per slot, as a thread of the workgroup kernel does."""
    g = golden
    centre = g["pf/centre"]
    rng = np.random.default_rng(17)
    lo = np.minimum(np.minimum(g["pf/centre"].min(0), g["pf/left"].min(0)), g["pf/right"].min(0))
    hi = np.maximum(np.maximum(g["pf/centre"].max(0), g["pf/left"].max(0)), g["pf/right"].max(0))
    seeds = rng.integers(0, len(centre), P)
    xy = centre[seeds] + rng.normal(0, 6.0, (P, 2))
    xy[:50] = centre[seeds[:50]] + rng.normal(0, 60.0, (50, 2))                    # tens of metres off: later rings
    xy[50:80] = rng.uniform(lo - 3000.0, hi + 3000.0, (30, 2))                       # outside the box: the wave's scan
    cells = rng.integers(0, 40, (40, 2))
    xy[80:120] = lo + 8.0 * cells                                                     # on the grid's lines
    xy[120:125] = centre[100]                                                         # on a map point, five times
    states = np.concatenate([xy, rng.uniform(-np.pi, np.pi, (P, 1))], axis=1).astype(np.float32)
    states[125, 0] = np.nan
    n_left, n_right = points
    left = np.stack([-4.5 + rng.normal(0, 0.2, n_left), np.linspace(0, 60, n_left)], axis=1).astype(np.float32)
    right = np.stack([4.5 + rng.normal(0, 0.2, n_right), np.linspace(0, 45, n_right)], axis=1).astype(np.float32)
    grid = _scorer(g, n_particles=P).update_particles(states, [left, right])
    monkeypatch.setenv("ACMPC_PF_WORKGROUP_SCORE", "1")
    workgroup = _scorer(g, n_particles=P).update_particles(states, [left, right])
    monkeypatch.setenv("ACMPC_PF_NO_GRID", "1")
    scan = _scorer(g, n_particles=P).update_particles(states, [left, right])
    for key in ("track_indices", "minimum_offset", "heading_offset", "observation_error", "score", "valid_mask"):
        np.testing.assert_array_equal(grid[key], scan[key], err_msg=key)
        np.testing.assert_array_equal(grid[key], workgroup[key], err_msg=key)
    # and the brute-force statement itself on the finite ones
    some = np.r_[0:125, 126:600]   # (all but the non-finite one of the special cases, and a few hundred ordinary ones)
    d2 = ((states[some, None, :2].astype(np.float64) - centre[None]) ** 2).sum(-1)
    np.testing.assert_array_equal(grid["track_indices"][some, 0], d2.argmin(1))


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



def _scene(golden, n, seed=21):
    """A cluster of particles round a pose on the golden map and an observation of both track limits from there."""
    g = golden
    centre, left, right = g["pf/centre"], g["pf/left"], g["pf/right"]
    M = len(centre)
    rng = np.random.default_rng(seed)
    tangent = np.roll(centre, -1, axis=0) - np.roll(centre, 1, axis=0)
    heading = np.unwrap(np.arctan2(tangent[:, 1], tangent[:, 0]))
    idx = 700
    pose = np.array([centre[idx, 0], centre[idx, 1], heading[idx]])
    states = (pose + rng.normal(0, [1.5, 1.5, 0.08], (n, 3))).astype(np.float32)
    states[::7, :2] += 40.0                                    # some particles far off the track: invalid
    a = np.pi / 2 - heading[idx]
    rot = np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]])
    obs = {k: ((t[(idx + np.arange(c)) % M] - centre[idx]) @ rot.T + rng.normal(0, 0.15, (c, 2))).astype(np.float32)
           for k, t, c in (("left", left, 230), ("right", right, 210))}
    return dict(centre=centre, left=left, right=right), states, obs, idx, heading


@pytest.mark.parametrize("n,n_desired", [(300, 300), (5000, 6000), (300, 200), (20000, 24000)])
def test_device_resampling_matches_its_restatement(golden, n, n_desired):
    """One update of the device-resident filter against the oracle's restatement of its counter-based resampling:
    the same particles kept in the same order, the SAME picked indices for every new particle (integer weights and
    prefix sums: exact), the noise to the accuracy of the device's fast float32 transcendentals, and the estimate of
    the weighted mean.  (24 000: the tiled kernels of capacities from 8 192 up - a launch per step over tiles of 1 024
    particles instead of one workgroup for everything.)"""
    from acmpc_amd.particle_filter import DeviceParticleFilter, ParticleScorer
    track, states, obs, _, _ = _scene(golden, n)
    cfg = dict(LOCALISATION, n_particles=max(n, n_desired), n_converged_particles=n_desired,
               sampling_noise=dict(x=1.1, y=1.1, yaw=3.0), control_noise=dict(velocity=0.25, yaw=2.0),
               convergence_criteria=dict(maximum_distance=50, maximum_angle=90))
    pf = DeviceParticleFilter(cfg, track, seed=1234)
    scores0 = np.full(n, 1.0 / n, dtype=np.float32)
    pf.set_particles(states, scores0)
    out = pf.update(obs)
    got_states, got_scores = pf.particles()
    # the scoring the update used, through the host-pointer seam (pinned to the reference above)
    scorer = ParticleScorer(cfg, track)
    scored = scorer.update_particles(states, scorer.downsample_observations(obs))
    sigma = (1.1, 1.1, 3.0 * np.pi / 180)
    want = orc.pf_resample_counter_based(states, scored["score"].astype(np.float32), scored["score"], scored["valid_mask"],
                                         max(n, n_desired), cfg["thresholds"]["minimum_particles"], sigma, 1234, 1)
    assert want is not None and not out["was_reset"]
    want_states, want_scores, picked = want
    n_valid = int(scored["valid_mask"].sum())
    assert out["n_valid"] == n_valid and out["n_particles"] == want_states.shape[0] == got_states.shape[0]
    np.testing.assert_array_equal(got_states[:n_valid], want_states[:n_valid])          # kept, in order
    np.testing.assert_array_equal(got_scores, want_scores)                                # scores follow the picks exactly
    # (round 4: the sampler's normals are specified bit for bit - oracle box_muller_spec - so the fresh particles are exact too)
    np.testing.assert_array_equal(got_states[n_valid:], want_states[n_valid:])
    assert len(np.unique(picked)) > 1
    est, max_d, max_a = scorer.estimate_location(got_scores, got_states)
    np.testing.assert_allclose(out["estimate"], est, rtol=1e-9)
    np.testing.assert_allclose([out["max_distance"], out["max_angle"]], [max_d, max_a], rtol=1e-9)
    scorer.close()


def test_the_tiled_filter_kernels_equal_the_one_workgroup_ones(golden, monkeypatch):
    """Capacity 20 000: three updates of the device-resident filter through the tiled resampling and estimate
    (pf_resample_tiles / plan / scatter / emit, pf_estimate_partial / spread / final) and through the one-workgroup kernels
    (ACMPC_PF_NARROW_FILTER=1): the same particles and scores bit for bit - ranks, prefix sums and picks are integers - and
    the same estimate to the order of its float64 sums; a reset (nothing near the track) likewise."""
    from acmpc_amd.particle_filter import DeviceParticleFilter
    n = 20000
    track, states, obs, _, _ = _scene(golden, n)
    cfg = dict(LOCALISATION, n_particles=n, n_converged_particles=15000, sampling_noise=dict(x=1.1, y=1.1, yaw=3.0),
               control_noise=dict(velocity=0.25, yaw=2.0), convergence_criteria=dict(maximum_distance=50, maximum_angle=90))
    runs = []
    for narrow in (False, True):
        if narrow:
            monkeypatch.setenv("ACMPC_PF_NARROW_FILTER", "1")
        pf = DeviceParticleFilter(cfg, track, seed=77)
        pf.set_particles(states, np.full(n, 1.0 / n, dtype=np.float32))
        seen = []
        for k in range(3):
            out = pf.update(obs)
            seen.append((out, pf.particles()))
            pf.step(0.01, 30.0, 0.05)
        far = states.copy()
        far[:, :2] += 5000.0
        pf.set_particles(far, np.full(n, 1.0 / n, dtype=np.float32))
        out = pf.update(obs)
        assert out["was_reset"]
        seen.append((out, pf.particles()))
        runs.append(seen)
    for (wide_out, (wide_states, wide_scores)), (narrow_out, (narrow_states, narrow_scores)) in zip(*runs):
        np.testing.assert_array_equal(wide_states, narrow_states)
        np.testing.assert_array_equal(wide_scores, narrow_scores)
        assert wide_out["n_particles"] == narrow_out["n_particles"] and wide_out["n_valid"] == narrow_out["n_valid"]
        assert wide_out["was_reset"] == narrow_out["was_reset"]
        np.testing.assert_allclose(wide_out["estimate"], narrow_out["estimate"], rtol=1e-10)
        np.testing.assert_allclose([wide_out["max_distance"], wide_out["max_angle"]],
                                   [narrow_out["max_distance"], narrow_out["max_angle"]], rtol=1e-10)


def test_device_filter_resets_when_too_few_particles_are_valid(golden):
    from acmpc_amd.particle_filter import DeviceParticleFilter
    track, states, obs, _, _ = _scene(golden, 300)
    cfg = dict(LOCALISATION, n_particles=300, n_converged_particles=300, sampling_noise=dict(x=1.1, y=1.1, yaw=3.0),
               control_noise=dict(velocity=0.25, yaw=2.0), convergence_criteria=dict(maximum_distance=50, maximum_angle=90))
    pf = DeviceParticleFilter(cfg, track, seed=5)
    reset_states, reset_scores = pf.particles()                      # _reset_filter on the device ...
    want_states, want_scores = orc.pf_reset(track["centre"], 300)     # ... against the reference's (G: pf_reset)
    np.testing.assert_array_equal(reset_states, want_states)
    np.testing.assert_array_equal(reset_scores, want_scores)
    far = states.copy()
    far[:, :2] += 500.0                                               # nothing near the track: everything invalid
    pf.set_particles(far, np.full(300, 1 / 300, dtype=np.float32))
    out = pf.update(obs)
    assert out["was_reset"] and out["n_valid"] < cfg["thresholds"]["minimum_particles"] and out["n_particles"] == 300
    np.testing.assert_array_equal(pf.particles()[0], want_states)


def test_device_filter_tracks_a_moving_car(golden):
    """The cycle of test_filter_cycle_tracks_a_moving_car with the particles living on the GPU: one round trip per
    update, counter-based draws on the device."""
    from acmpc_amd.particle_filter import DeviceParticleFilter
    track, states, _, idx, heading = _scene(golden, 300, seed=11)
    states[::7, :2] -= 40.0
    centre, left, right = track["centre"], track["left"], track["right"]
    M = len(centre)
    cfg = dict(LOCALISATION, n_particles=300, n_converged_particles=300, sampling_noise=dict(x=1.1, y=1.1, yaw=3.0),
               control_noise=dict(velocity=0.25, yaw=2.0), convergence_criteria=dict(maximum_distance=50, maximum_angle=90))
    pf = DeviceParticleFilter(cfg, track, seed=99)
    pf.set_particles(states, np.full(300, 1 / 300, dtype=np.float32))
    rng = np.random.default_rng(11)
    spacing = float(np.mean(np.linalg.norm(np.diff(centre, axis=0), axis=1)))
    step, dt = 3, 0.05
    speed = step * spacing / dt

    def observe(t, count, at, yaw):
        a = np.pi / 2 - yaw
        rot = np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]])
        return ((t[(at + np.arange(count)) % M] - centre[at]) @ rot.T + rng.normal(0, 0.15, (count, 2))).astype(np.float32)

    errors = []
    for _ in range(40):
        yaw_rate = (heading[(idx + step) % M] - heading[idx]) / dt
        pf.step(float(np.arctan(yaw_rate * 2.65 / speed)), speed, dt)
        idx = (idx + step) % M
        out = pf.update({"left": observe(left, 230, idx, heading[idx]), "right": observe(right, 210, idx, heading[idx])})
        assert not out["was_reset"] and out["n_particles"] == 300
        errors.append(np.linalg.norm(out["estimate"][:2] - centre[idx]))
    assert pf.is_converged
    assert max(errors) < 3.0 and np.mean(errors[20:]) < 2.0, (max(errors), np.mean(errors[20:]))
