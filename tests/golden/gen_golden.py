#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference (never copying it).

Run in the build container only (`/root/reference` does not exist on the GPU box):

    python tests/golden/gen_golden.py

The reference's hot-path modules are pure Python/NumPy/SciPy; its third-party imports that are not
installed here (`osqp`, `loguru`, `ace`, `aci`, `ruamel.yaml`) are replaced by inert stand-ins in
`sys.modules` so that the modules import.  Everything recorded below is computed by the reference's own
functions; the two `osqp.solve()` calls are the only statements that cannot run, so no QP *solution* is
recorded (parity unpinned there).

Vehicle scalars are builder-chosen placeholders (the reference's vehicle file is a network asset):
wheelbase 2.65 m, width 1.94 m, max steering angle 0.30 rad.
"""
from __future__ import annotations

import os
import sys
import types
from types import SimpleNamespace

import numpy as np

REF_SRC = "/root/reference/src"
OUT_DIR = os.path.dirname(os.path.abspath(__file__))

WHEELBASE, WIDTH, DELTA_MAX = 2.65, 1.94, 0.30


def install_stubs():
    def module(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    sink = SimpleNamespace(**{k: (lambda *a, **kw: None) for k in ("info", "warning", "debug", "error")})
    module("loguru", logger=sink)

    class OSQP:  # records the problem, cannot solve it
        def setup(self, **kw):
            self.problem = kw

        def update(self, **kw):
            self.updated = kw

        def solve(self):
            n = self.problem["q"].shape[0]
            return SimpleNamespace(x=np.zeros(n), info=SimpleNamespace(status="not solved (stub)"))

    module("osqp", OSQP=OSQP)

    class SteeringGeometry:
        def __init__(self, path=None):
            self.vehicle_data = SimpleNamespace(wheelbase=WHEELBASE, width=WIDTH)

        def max_steering_angle(self):
            return DELTA_MAX

        def steering_angle(self, u):
            return u * DELTA_MAX

    module("ace")
    module("ace.steering", SteeringGeometry=SteeringGeometry)

    class SystemMonitor:
        def __init__(self, *a, **kw):
            pass

    def track_runtime(monitor):
        return lambda fn: fn

    module("aci")
    module("aci.utils")
    module("aci.utils.system_monitor", SystemMonitor=SystemMonitor, track_runtime=track_runtime)
    module("ruamel")
    module("ruamel.yaml", YAML=object)
    return SteeringGeometry


RACING = {
    # configs/<track>.yaml racing.control blocks (monza.yaml:67-81 etc.)
    "monza": dict(horizon=50, unlocalised_max_speed=28,
                  speed_profile_constraints=dict(v_min=8.0, v_max=84.0, a_min=-1.3, a_max=1.0, ay_max=5.5,
                                                 ki_min=0.005, end_velocity=14.0),
                  step_cost=[4.0e-3, 5.0e-2, 0.0], r_term=[1.0e-2, 10.0], final_cost=[1.0, 0.0, 0.1]),
    "spa": dict(horizon=50, unlocalised_max_speed=8.0,
                speed_profile_constraints=dict(v_min=5.0, v_max=84.0, a_min=-1.0, a_max=1.0, ay_max=4.0,
                                               ki_min=0.003, end_velocity=20.0),
                step_cost=[1.0e-3, 0.0, 0.0], r_term=[1.0e-2, 10.0], final_cost=[1.0, 0.0, 0.1]),
    "nordschleife": dict(horizon=50, unlocalised_max_speed=20,
                         speed_profile_constraints=dict(v_min=12.0, v_max=84.0, a_min=-1.0, a_max=1.0,
                                                        ay_max=3.0, ki_min=0.0, end_velocity=14.0),
                         step_cost=[2.0e-4, 0.0, 0.0], r_term=[1.0e-2, 10.0], final_cost=[1.0, 0.0, 0.1]),
    "silverstone": dict(horizon=50, unlocalised_max_speed=32.0,
                        speed_profile_constraints=dict(v_min=8.0, v_max=84.0, a_min=-1.0, a_max=1.0,
                                                       ay_max=5.0, ki_min=0.003, end_velocity=20.0),
                        step_cost=[2.0e-3, 5.0e-2, 0.0], r_term=[1.0e-2, 10.0], final_cost=[1.0, 0.0, 0.1]),
}


def track_families(H, utils):
    """The four synthetic families of tests/test_spatial_mpc.py:45-75 (7 parameters each)."""
    out = []
    for r in np.linspace(10, 100, 7):
        out.append(("hairpin_%g" % r, utils.get_hairpin_track(r, H, -np.pi / 6)))
    for d in np.linspace(40, 100, 7):
        out.append(("chicane_%g" % d, utils.get_chicane_track(d, 40, H, 0.1)))
    for c in np.linspace(-0.02, 0.02, 7):
        out.append(("curve_%g" % c, utils.get_curved_track(c, H, 0.1)))
    for length in np.linspace(40, 200, 7):
        out.append(("straight_%g" % length, utils.get_straight_track(length, H, 0.1)))
    return out


def main():
    SteeringGeometry = install_stubs()
    sys.path.insert(0, REF_SRC)
    from acmpc.control import utils as ref_utils
    from acmpc.control.commands import TemporalCommandInterpolator, TemporalCommandSelector
    from acmpc.control.controller import ControlProcess, build_mpc
    from acmpc.localisation.localiser import LocalisationProcess, Localiser
    from acmpc.utils.kdtree import KDTree

    rng = np.random.default_rng(20250202)
    vehicle = SteeringGeometry()
    out = {}

    # ---- G1-G6: waypoints, Frenet transforms, linearisation, QP assembly, speed-profile QP inputs -------
    cases = []
    for track_name, H in (("monza", 20), ("monza", 50), ("spa", 50), ("nordschleife", 80), ("silverstone", 50)):
        cfg = dict(RACING[track_name], horizon=H)
        mpc = build_mpc(cfg, vehicle)
        model = mpc.model
        for fam_name, (x, y) in track_families(H, ref_utils)[:: (1 if H == 50 and track_name == "monza" else 5)]:
            width = np.linspace(10.0, 6.0, H) if "hairpin" not in fam_name else np.full(H, 100.0)
            coords = np.stack([x, y, width]).T
            path = mpc.construct_waypoints(coords)
            table_no_v = path._reference_path.copy()
            # speed profile inputs (the stub cannot solve; v_ref below is a deterministic surrogate)
            sp_inputs = {}
            for localised in (False, True):
                solver = mpc._localised_speed_profile_solver if localised else mpc._speed_profile_solver
                solver.solve(path, cfg["speed_profile_constraints"]["end_velocity"])
                sp_inputs[localised] = dict(q=np.array(solver._q), A=solver._A.toarray(),
                                            l=np.array(solver._lower_bounds), u=np.array(solver._upper_bounds),
                                            v_hi=np.array(solver._max_velocities))
            v_ref = np.clip(sp_inputs[False]["v_hi"] - 2.0 + rng.normal(0, 0.5, H - 1), 5.0, 84.0)
            path.velocities = v_ref
            table = path._reference_path.copy()
            offset = float(rng.uniform(-1.0, 1.0))
            state = np.array([offset, 0.0, np.pi / 2])
            spatial_state = model.t2s(path.get_state(0), state)
            f, A, B = model.linearise(path)
            solver = mpc._control_solver
            solver.solve(spatial_state, path)
            states = rng.normal(0, 0.3, (H - 1, 3))
            key = "%s_H%d_%s" % (track_name, H, fam_name)
            cases.append(key)
            out[key + "/coords"] = coords
            out[key + "/table_no_v"] = table_no_v
            out[key + "/table"] = table
            out[key + "/offset"] = np.array(offset)
            out[key + "/spatial_state"] = spatial_state
            out[key + "/lin_f"], out[key + "/lin_A"], out[key + "/lin_B"] = f, A, B
            out[key + "/qp_Pdiag"] = solver._P.diagonal()
            assert np.array_equal(solver._P.toarray(), np.diag(solver._P.diagonal()))  # P is diagonal
            out[key + "/qp_q"] = np.array(solver._q)
            out[key + "/qp_A"] = solver._A.toarray().astype(np.float64)
            out[key + "/qp_l"] = np.array(solver._lower_bounds)
            out[key + "/qp_u"] = np.array(solver._upper_bounds)
            out[key + "/s2t_states"] = states
            out[key + "/s2t_out"] = model.s2t(path, states)
            out[key + "/prediction"] = mpc.update_prediction(states, path)
            for localised in (False, True):
                for k, v in sp_inputs[localised].items():
                    out[key + "/sp%d_%s" % (int(localised), k)] = v
            out[key + "/weights"] = np.array(cfg["step_cost"] + cfg["r_term"] + cfg["final_cost"])
            out[key + "/limits"] = np.array([model.length, model.width, model.delta_max, model.margin,
                                             model.min_u[0], model.min_u[1], model.max_u[0], model.max_u[1]])
    out["cases"] = np.array(cases)

    # t2s on random poses
    mpc = build_mpc(RACING["monza"], vehicle)
    wps = rng.normal(0, 20, (64, 3))
    sts = rng.normal(0, 20, (64, 3))
    wps[:, 2] = rng.uniform(-4, 4, 64)
    sts[:, 2] = rng.uniform(-7, 7, 64)
    out["t2s/waypoints"], out["t2s/states"] = wps, sts
    out["t2s/out"] = np.array([mpc.model.t2s(w, s) for w, s in zip(wps, sts)])

    # ---- G7: batched kinematic bicycle derivative (localiser.py:77-95) -------------------------------------
    fake_loc = SimpleNamespace(_localiser=SimpleNamespace(wheel_base=WHEELBASE))
    pstates = rng.normal(0, 50, (500, 3)).astype(np.float32)
    pstates[:, 2] = rng.uniform(-np.pi, np.pi, 500).astype(np.float32)
    delta = (0.05 + rng.normal(0, 0.02, 500)).astype(np.float32)
    vel = np.abs(30.0 + rng.normal(0, 1.0, 500)).astype(np.float32)
    out["xdot/states"], out["xdot/delta"], out["xdot/velocity"] = pstates, delta, vel
    out["xdot/out"] = Localiser._calculate_x_dot(fake_loc, delta, pstates, vel)
    out["xdot/wheel_base"] = np.array(WHEELBASE)

    # ---- G8: nearest waypoint (KD-tree) + heading offset on a synthetic closed map --------------------------
    s = np.linspace(0, 2 * np.pi, 1201)[:-1]
    centre = np.stack([300 * np.cos(s) + 40 * np.cos(3 * s), 200 * np.sin(s) + 25 * np.sin(2 * s)], axis=1)
    tangent = np.gradient(centre, axis=0)
    normal = np.stack([-tangent[:, 1], tangent[:, 0]], axis=1)
    normal /= np.linalg.norm(normal, axis=1, keepdims=True)
    left, right = centre + 4.75 * normal, centre - 4.75 * normal
    fake_proc = SimpleNamespace(centre_track=KDTree(centre), left_track=KDTree(left), right_track=KDTree(right))
    fake_proc._get_track_points_by_index = lambda idx: LocalisationProcess._get_track_points_by_index(fake_proc, idx)
    pts = centre[rng.integers(0, len(centre), 400)] + rng.normal(0, 6.0, (400, 2))
    offs, idx = LocalisationProcess._find_closest_points_to_particles(fake_proc, pts)
    headings = rng.uniform(-np.pi, np.pi, 400)
    particles = {"states": np.concatenate([pts, headings[:, None]], axis=1)}
    here = fake_proc._get_track_points_by_index(idx[:, 0])
    nxt = fake_proc._get_track_points_by_index(idx[:, 0] + 1)
    out["nn/centre"], out["nn/left"], out["nn/right"], out["nn/points"] = centre, left, right, pts
    out["nn/offsets"], out["nn/indices"] = offs, idx
    out["nn/headings"] = headings
    out["nn/heading_offset"] = LocalisationProcess._calculate_heading_offset(fake_proc, here, particles, nxt)

    # ---- G9: weighted-mean estimate incl. the NaN fallback (localiser.py:572-579) ---------------------------
    scores = rng.uniform(0, 1, 300).astype(np.float32)
    st = rng.normal(0, 30, (300, 3)).astype(np.float32)
    out["est/scores"], out["est/states"] = scores, st
    out["est/out"] = LocalisationProcess._estimate_location(None, scores, st)
    zero_scores = np.zeros(300, dtype=np.float32)
    with np.errstate(all="ignore"):
        out["est/out_nan_fallback"] = LocalisationProcess._estimate_location(None, zero_scores, st)

    # ---- G10: command selection / interpolation (commands.py; known answers of tests/test_commands.py) ------
    cum_time = np.cumsum(rng.uniform(0.02, 0.06, 49)).astype(np.float32)
    commands = np.stack([rng.uniform(8, 60, 49), rng.uniform(-0.3, 0.3, 49)], axis=1).astype(np.float32)
    holder = SimpleNamespace(control_cumtime=cum_time, control_inputs=commands)
    selector = TemporalCommandSelector(holder)
    elapsed = np.concatenate([[-0.1, 0.0, float(cum_time[0]) * 0.5, float(cum_time[-1]) + 1.0],
                              rng.uniform(0, float(cum_time[-1]), 60)])
    out["cmd/cum_time"], out["cmd/commands"], out["cmd/elapsed"] = cum_time, commands, elapsed
    out["cmd/selected"] = np.array([selector(t) for t in elapsed])
    # interpolator: the reference test's own vectors (tests/test_commands.py:14-53) mapped to the current
    # attribute names (control_cumtime / control_inputs, commands.py:45-51)
    ct_a = np.round(np.linspace(0, 1, 10), 1)
    interp = TemporalCommandInterpolator(SimpleNamespace(control_cumtime=ct_a, control_inputs=np.zeros((2, 10))))
    el_a = np.array([0, 0.22, 1.0, 0.95, 0.77])
    res_a = [interp._get_closet_command_index(t) for t in el_a]
    out["cmd/interp_a_cum_time"], out["cmd/interp_a_elapsed"] = ct_a, el_a
    out["cmd/interp_a_index"] = np.array([r[0] for r in res_a])
    out["cmd/interp_a_distance"] = np.array([r[1] for r in res_a])
    ct_b = np.linspace(0, 1, 11)
    pc_b = np.array([[17.0, -0.03], [0.0, 0.0], [5.0, 0.15], [1.0, 0.0], [0.0, 0.0], [0.0, 0.0], [0.0, 0.0],
                     [-5, -0.06], [12.0, 0.04], [1.0, 0.4], [-2.0, 0.02]]).T
    interp = TemporalCommandInterpolator(SimpleNamespace(control_cumtime=ct_b, control_inputs=pc_b))
    el_b = np.array([-0.1, 0.22, 1.0, 0.95, 0.77, 1.1])
    out["cmd/interp_b_cum_time"], out["cmd/interp_b_commands"], out["cmd/interp_b_elapsed"] = ct_b, pc_b, el_b
    out["cmd/interp_b_out"] = np.array([interp.get_command(t) for t in el_b])

    # ---- G11: centreline downsample (controller.py:256-267) -------------------------------------------------
    cl = np.stack([np.linspace(0, 3, 500) ** 2, np.linspace(0, 150, 500)], axis=1).astype(np.float32)
    for H in (20, 50, 100):
        fake_cp = SimpleNamespace(_perceiver=SimpleNamespace(centreline=cl), _control_horizon=H)
        out["downsample/H%d" % H] = ControlProcess._reference_path.fget(fake_cp)
    out["downsample/centreline"] = cl

    # ---- G12: particle scoring of the localiser (localiser.py:234-410,453-462) on a synthetic map ----------------
    import multiprocessing as mp
    from acmpc.perception.shared_memory import SharedPoints

    M = 3000
    s = np.linspace(0, 2 * np.pi, M + 1)[:-1]
    centre = np.stack([220 * np.cos(s) + 30 * np.cos(3 * s + 0.4), 150 * np.sin(s) + 18 * np.sin(2 * s)], axis=1)
    arc = np.concatenate([[0], np.cumsum(np.linalg.norm(np.diff(np.vstack([centre, centre[:1]]), axis=0), axis=1))])
    t = np.linspace(0, arc[-1], M + 1)[:-1]
    closed = np.vstack([centre, centre[:1]])
    centre = np.stack([np.interp(t, arc, closed[:, 0]), np.interp(t, arc, closed[:, 1])], axis=1)
    tang = np.roll(centre, -1, axis=0) - np.roll(centre, 1, axis=0)
    tang /= np.linalg.norm(tang, axis=1, keepdims=True)
    normal = np.stack([-tang[:, 1], tang[:, 0]], axis=1)
    left, right = centre + 4.75 * normal, centre - 4.75 * normal
    n_particles = 300
    proc = object.__new__(LocalisationProcess)       # the reference's own TestLocalisationProcess trick
    proc.centre_track, proc.left_track, proc.right_track = KDTree(centre), KDTree(left), KDTree(right)
    proc._set_average_distance_between_map_points(centre)
    proc._score_distribution_mean, proc._score_distribution_sigma = 0, 10          # configs/monza.yaml:61-63
    proc._initialise_score_distribution()
    proc._threshold_error, proc._threshold_offset = 20.0, 10                        # configs/monza.yaml:56-60
    proc._threshold_rotation = 90 * np.pi / 180
    proc._max_n_particles = n_particles
    proc._is_collecting_localisation_data = False
    proc.particle_lock = mp.Lock()
    proc._shared_particle_scores = SharedPoints(n_particles, 0)
    proc._shared_particle_states = SharedPoints(n_particles, 3)
    true_idx = 700
    heading = np.arctan2(tang[true_idx, 1], tang[true_idx, 0])
    pose = np.array([centre[true_idx, 0], centre[true_idx, 1], heading])
    seeds = rng.integers(0, M, n_particles)
    states = np.stack([centre[seeds, 0], centre[seeds, 1], np.arctan2(tang[seeds, 1], tang[seeds, 0])], axis=1)
    states[:150] = pose + rng.normal(0, [2.0, 2.0, 0.1], (150, 3))                  # a cluster near the truth
    states[150:] += rng.normal(0, [3.0, 3.0, 0.4], (150, 3))                        # the rest spread round the lap
    states[-5:, 2] += np.pi                                                         # some facing backwards
    states[-10:-5, :2] += 14.0                                                      # some far off the track
    proc.particle_scores = np.ones(n_particles, dtype=np.float32)
    proc.particle_states = states.astype(np.float32)

    def observe(track, count):
        pts = track[(true_idx + np.arange(count)) % M] - pose[:2]
        a = np.pi / 2 - heading
        rot = np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]])
        return (pts @ rot.T + rng.normal(0, 0.15, (count, 2))).astype(np.float32)

    observation = {"left": observe(left, 230), "right": observe(right, 210)}
    downsampled = proc._downsample_observations({k: v.copy() for k, v in observation.items()})
    obs_in = [o.copy() for o in downsampled]
    particles = proc._update_particles(downsampled)        # mutates the list: y < 50 filter
    out["pf/centre"], out["pf/left"], out["pf/right"] = centre, left, right
    out["pf/states"] = proc.particle_states
    out["pf/obs_left_raw"], out["pf/obs_right_raw"] = observation["left"], observation["right"]
    out["pf/obs_left_downsampled"], out["pf/obs_right_downsampled"] = obs_in
    out["pf/obs_left_used"], out["pf/obs_right_used"] = downsampled
    out["pf/track_indices"] = particles["track_indices"]
    out["pf/minimum_offset"] = particles["minimum_offset"]
    out["pf/heading_offset"] = particles["heading_offset"]
    out["pf/observation_error"] = particles["observation_error"]
    out["pf/score"] = particles["score"]
    out["pf/valid_mask"] = proc._get_valid_particle_mask(particles)
    out["pf/scale"] = np.array(proc._scale)
    out["pf/average_map_spacing"] = np.array(proc._average_distance_between_map_points)
    out["pf/published_scores"] = proc.particle_scores      # float32 shared array after _update_particle_scores

    # ---- G13: map ingestion helper (utils/load.py:30-35) ------------------------------------------------------------
    from acmpc.utils import load as ref_load
    pts = np.cumsum(rng.uniform(0.2, 0.6, (400, 2)), axis=0)
    pts[50] = pts[49]                       # exact duplicate
    pts[120] = pts[119] + 5e-5              # near duplicate (below 0.1 mm)
    pts[200] = pts[199] + np.array([2e-4, 0.0])   # just above the threshold: kept
    out["map/points"] = pts
    out["map/deduplicated"] = ref_load.remove_near_duplicate_points(pts)

    # ---- G14: resampling, reset and convergence flag of the localiser (localiser.py:420-570) -------------------------
    # continues from the scored particles of G12; the reference draws from NumPy's global stream, seeded here
    proc._threshold_n_particles = 20                                              # configs/monza.yaml:56-60
    proc._n_converged_particles = n_particles
    proc._sampling_noise_x, proc._sampling_noise_y = 1.1, 1.1                       # configs/monza.yaml:49-52
    proc._sampling_noise_yaw = 3.0 * np.pi / 180
    proc._convergence_distance, proc._convergence_angle = 50, 90                    # configs/monza.yaml:64-66
    proc._is_converged = mp.Value("i", False)
    proc._is_previously_converged = False
    np.random.seed(20240917)
    proc._resample_particles(particles)
    out["pf/resample_seed"] = np.array(20240917)
    out["pf/resampled_states"] = proc.particle_states
    out["pf/resampled_scores"] = proc.particle_scores
    proc._update_is_converged_flag()
    out["pf/resampled_is_converged"] = np.array(bool(proc.is_converged))
    # a second round from a tight cluster: converges
    tight = (pose + rng.normal(0, [1.0, 1.0, 0.05], (n_particles, 3))).astype(np.float32)
    proc.particle_scores = np.ones(n_particles, dtype=np.float32)
    proc.particle_states = tight
    out["pf/tight_states"] = tight
    particles = proc._update_particles([o.copy() for o in downsampled])
    out["pf/tight_score"] = particles["score"]
    np.random.seed(7)
    proc._resample_particles(particles)
    proc._update_is_converged_flag()
    out["pf/tight_resampled_states"], out["pf/tight_resampled_scores"] = proc.particle_states, proc.particle_scores
    out["pf/tight_is_converged"] = np.array(bool(proc.is_converged))
    # too few valid particles -> the filter is reset onto the centre line (localiser.py:468-485)
    proc._threshold_n_particles = n_particles + 1
    proc._resample_particles({k: v.copy() for k, v in particles.items()})
    out["pf/reset_states"], out["pf/reset_scores"] = proc.particle_states, proc.particle_scores
    out["pf/reset_is_converged"] = np.array(bool(proc.is_converged))

    path = os.path.join(OUT_DIR, "reference_ingredients.npz")
    np.savez_compressed(path, **out)
    print("wrote %s: %d arrays, %.1f kB" % (path, len(out), os.path.getsize(path) / 1e3))


if __name__ == "__main__":
    main()
