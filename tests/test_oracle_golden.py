"""The oracle's ingredients against vectors computed by the reference itself (not GPU)."""
from types import SimpleNamespace

import numpy as np

import acmpc_oracle as orc


def _limits(v):
    return SimpleNamespace(length=v[0], width=v[1], delta_max=v[2], margin=v[3],
                           min_u=np.array(v[4:6]), max_u=np.array(v[6:8]))


def _weights(w):
    return dict(step_cost=w[0:3], r_term=w[3:5], final_cost=w[5:8])


def test_case_list_is_not_empty(golden_cases):
    assert len(golden_cases) >= 40


def test_construct_waypoints(golden, golden_cases):
    # spatial_mpc.py:125-154
    for key in golden_cases:
        table = orc.construct_waypoints(golden[key + "/coords"])
        np.testing.assert_allclose(table, golden[key + "/table_no_v"], rtol=0, atol=1e-12, err_msg=key)


def test_t2s(golden):
    # dynamics.py:23-40
    got = np.array([orc.t2s(w, s) for w, s in zip(golden["t2s/waypoints"], golden["t2s/states"])])
    np.testing.assert_allclose(got, golden["t2s/out"], rtol=0, atol=1e-12)


def test_s2t_and_prediction(golden, golden_cases):
    # dynamics.py:42-63, spatial_mpc.py:156-168
    for key in golden_cases:
        out = orc.s2t(golden[key + "/table"], golden[key + "/s2t_states"])
        np.testing.assert_allclose(out, golden[key + "/s2t_out"], rtol=0, atol=1e-12)
        np.testing.assert_allclose(out[:-1].T, golden[key + "/prediction"], rtol=0, atol=1e-12)


def test_linearise(golden, golden_cases):
    # dynamics.py:65-103
    for key in golden_cases:
        f, A, B = orc.linearise(golden[key + "/table"])
        np.testing.assert_allclose(f, golden[key + "/lin_f"], rtol=1e-14, atol=0)
        np.testing.assert_allclose(A, golden[key + "/lin_A"], rtol=1e-14, atol=0)
        np.testing.assert_allclose(B, golden[key + "/lin_B"], rtol=1e-14, atol=0)


def test_vehicle_limits(golden, golden_cases):
    # dynamics.py:10-21 with the placeholder vehicle of gen_golden.py
    v = golden[golden_cases[0] + "/limits"]
    lim = orc.vehicle_limits(2.65, 1.94, 0.30, v[4], v[6])
    assert lim.margin == v[3]
    np.testing.assert_array_equal(lim.min_u, v[4:6])
    np.testing.assert_array_equal(lim.max_u, v[6:8])


def test_control_qp_assembly(golden, golden_cases):
    # control.py:15-79,121-158: P, q, A, l, u exactly as the reference assembles them
    for key in golden_cases:
        qp = orc.control_qp(golden[key + "/spatial_state"], golden[key + "/table"],
                            _weights(golden[key + "/weights"]), _limits(golden[key + "/limits"]))
        np.testing.assert_allclose(qp["P_diag"], golden[key + "/qp_Pdiag"], rtol=0, atol=0)
        np.testing.assert_allclose(qp["q"], golden[key + "/qp_q"], rtol=1e-14, atol=1e-15)
        np.testing.assert_allclose(qp["A"], golden[key + "/qp_A"], rtol=1e-14, atol=0)
        np.testing.assert_allclose(qp["l"], golden[key + "/qp_l"], rtol=1e-13, atol=1e-13)
        np.testing.assert_allclose(qp["u"], golden[key + "/qp_u"], rtol=1e-13, atol=1e-13)


def test_speed_profile_qp_inputs(golden, golden_cases):
    # speed_profile.py:26-59 and :131-150 (inputs of the QP; its solution needs osqp -> unpinned)
    from test_support import RACING
    for key in golden_cases:
        track = key.split("_")[0]
        cons = RACING[track]["speed_profile_constraints"]
        for localised in (False, True):
            qp = orc.speed_profile_qp(golden[key + "/table_no_v"], cons, cons["end_velocity"], localised)
            tag = key + "/sp%d_" % int(localised)
            np.testing.assert_allclose(qp["v_hi"], golden[tag + "v_hi"], rtol=1e-14, atol=0)
            np.testing.assert_allclose(qp["q"], golden[tag + "q"], rtol=1e-14, atol=0)
            np.testing.assert_allclose(qp["A"], golden[tag + "A"], rtol=1e-14, atol=0)
            np.testing.assert_allclose(qp["l"], golden[tag + "l"], rtol=0, atol=0)
            np.testing.assert_allclose(qp["u"], golden[tag + "u"], rtol=1e-14, atol=0)


def test_kinematic_x_dot(golden):
    # localiser.py:77-95 (float32 particles)
    got = orc.kinematic_x_dot(golden["xdot/delta"], golden["xdot/states"], golden["xdot/velocity"],
                              float(golden["xdot/wheel_base"]))
    assert got.dtype == golden["xdot/out"].dtype
    np.testing.assert_array_equal(got, golden["xdot/out"])


def test_nearest_waypoint_and_heading_offset(golden):
    # localiser.py:282-318: brute-force first-minimum == KDTree.query
    pts = golden["nn/points"]
    idx_ref = golden["nn/indices"]
    for col, name in enumerate(("centre", "left", "right")):
        dist, idx = orc.nearest_waypoint(pts, golden["nn/" + name])
        np.testing.assert_array_equal(idx, idx_ref[:, col])
        if name == "centre":
            np.testing.assert_allclose(dist, golden["nn/offsets"], rtol=1e-12)
    ho = orc.heading_offset(golden["nn/centre"], idx_ref[:, 0], golden["nn/headings"])
    np.testing.assert_allclose(ho, golden["nn/heading_offset"], rtol=0, atol=1e-12)


def test_estimate_location(golden):
    # localiser.py:572-579
    got = orc.estimate_location(golden["est/scores"], golden["est/states"])
    np.testing.assert_allclose(got, golden["est/out"], rtol=1e-5)
    with np.errstate(all="ignore"):
        got = orc.estimate_location(np.zeros_like(golden["est/scores"]), golden["est/states"])
    np.testing.assert_allclose(got, golden["est/out_nan_fallback"], rtol=1e-5)


def test_command_selector(golden):
    # commands.py:20-38 incl. elapsed < cum_time[0] -> index -1 -> LAST command
    ct, cmds = golden["cmd/cum_time"], golden["cmd/commands"]
    for t, want in zip(golden["cmd/elapsed"], golden["cmd/selected"]):
        np.testing.assert_array_equal(orc.select_command(ct, cmds, t), want)
    np.testing.assert_array_equal(orc.select_command(ct, cmds, -0.1), cmds[-1])


def test_command_interpolator_known_answers(golden):
    # the reference's own known answers, tests/test_commands.py:15-16 and :43-53
    ct = golden["cmd/interp_a_cum_time"]
    want = [(0, 0.0), (2, -0.02), (9, 0.0), (8, -0.05), (7, 0.03)]
    for t, (i_want, d_want), i_ref, d_ref in zip(golden["cmd/interp_a_elapsed"], want,
                                                 golden["cmd/interp_a_index"], golden["cmd/interp_a_distance"]):
        i, d = orc.closest_command_index(ct, t)
        assert i == i_want == i_ref
        assert abs(d - d_want) < 1e-7 and abs(d - d_ref) < 1e-15
    ct, cmds = golden["cmd/interp_b_cum_time"], golden["cmd/interp_b_commands"].T
    expected = np.array([[17, -0.03], [4.2, 0.12], [-2.0, 0.02], [-0.5, 0.21], [6.9, 0.01], [-2.0, 0.02]])
    for t, want, ref in zip(golden["cmd/interp_b_elapsed"], expected, golden["cmd/interp_b_out"]):
        got = orc.interpolate_command(ct, cmds, t)
        np.testing.assert_allclose(got, want, atol=1e-7)
        np.testing.assert_allclose(got, ref, atol=1e-14)


def test_downsample_centreline(golden):
    # controller.py:256-267
    for H in (20, 50, 100):
        got = orc.downsample_centreline(golden["downsample/centreline"], H)
        assert got.shape == (H, 3)
        np.testing.assert_array_equal(got, golden["downsample/H%d" % H])


def test_particle_scoring(golden):
    # localiser.py:234-410,453-462 on the synthetic map of gen_golden.py
    g = golden
    spacing = float(g["pf/average_map_spacing"])
    for side in ("left", "right"):
        np.testing.assert_array_equal(orc.pf_downsample_observation(g["pf/obs_%s_raw" % side], spacing),
                                      g["pf/obs_%s_downsampled" % side])
    assert abs(orc.pf_score_scale(0, 10) - float(g["pf/scale"])) < 1e-18
    out = orc.pf_score_particles(g["pf/states"], g["pf/centre"], g["pf/left"], g["pf/right"],
                                 g["pf/obs_left_downsampled"], g["pf/obs_right_downsampled"], 0, 10,
                                 dict(rotation=np.pi / 2, offset=10, track_limit=20.0))
    np.testing.assert_array_equal(out["track_indices"], g["pf/track_indices"])
    np.testing.assert_allclose(out["minimum_offset"], g["pf/minimum_offset"], rtol=1e-12)
    np.testing.assert_allclose(out["heading_offset"], g["pf/heading_offset"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(out["observation_error"], g["pf/observation_error"], rtol=1e-6)
    np.testing.assert_allclose(out["score"], g["pf/score"], rtol=1e-6)
    np.testing.assert_array_equal(out["valid"], g["pf/valid_mask"])
    np.testing.assert_allclose(out["score"].astype(np.float32), g["pf/published_scores"], rtol=1e-6)


def test_particle_resampling_reset_and_convergence(golden):
    # localiser.py:420-570, continued from the scored particles above; same global-stream seeds as gen_golden.py
    g = golden
    sigma = (1.1, 1.1, 3.0 * np.pi / 180)
    np.random.seed(int(g["pf/resample_seed"]))
    states, scores = orc.pf_resample(g["pf/states"], g["pf/published_scores"], g["pf/score"], g["pf/valid_mask"],
                                     300, 20, sigma)
    np.testing.assert_array_equal(states, g["pf/resampled_states"])
    np.testing.assert_array_equal(scores, g["pf/resampled_scores"])
    assert orc.pf_convergence(scores, states, 50, 90)[1] == bool(g["pf/resampled_is_converged"])
    # a tight cluster converges
    tight = orc.pf_score_particles(g["pf/tight_states"], g["pf/centre"], g["pf/left"], g["pf/right"],
                                   g["pf/obs_left_downsampled"], g["pf/obs_right_downsampled"], 0, 10,
                                   dict(rotation=np.pi / 2, offset=10, track_limit=20.0))
    np.testing.assert_allclose(tight["score"], g["pf/tight_score"], rtol=1e-6)
    np.random.seed(7)
    states, scores = orc.pf_resample(g["pf/tight_states"], g["pf/tight_score"].astype(np.float32), g["pf/tight_score"],
                                     tight["valid"], 300, 20, sigma)
    np.testing.assert_array_equal(states, g["pf/tight_resampled_states"])
    np.testing.assert_array_equal(scores, g["pf/tight_resampled_scores"])
    assert orc.pf_convergence(scores, states, 50, 90)[1] == bool(g["pf/tight_is_converged"]) is True
    # too few valid particles: None = "reset", and the reset itself
    assert orc.pf_resample(states, scores, g["pf/tight_score"], tight["valid"], 300, 301, sigma) is None
    states, scores = orc.pf_reset(g["pf/centre"], 300)
    np.testing.assert_array_equal(states, g["pf/reset_states"])
    np.testing.assert_array_equal(scores, g["pf/reset_scores"])

