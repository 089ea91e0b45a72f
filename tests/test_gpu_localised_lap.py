"""The pieces together, wired the way the reference's agent wires them (agent.py:137-143,286-296; controller.py:49-57,
233-243; localiser.py:41-77,234-239): lap speed profile at race start, particle filter on synthetic track-limit
observations, the localised reference speed window written into `speed_profile_constraints["v_max"]` before every
solve, `get_control(path, is_localised)`, the command one step in applied to a kinematic car.

Functional, not parity (the reference has no closed-loop vectors): the car must stay on the road with no
infeasible solve, the filter must keep tracking it, and once localised the plan must use the map's speed profile -
faster than the unlocalised cap where the map allows it."""
import copy

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LOCALISATION = dict(n_particles=300, n_converged_particles=300,                       # configs/monza.yaml:43-66
                    sampling_noise=dict(x=1.1, y=1.1, yaw=3.0), control_noise=dict(velocity=0.25, yaw=2.0),
                    thresholds=dict(offset=10, rotation=90, minimum_particles=20, track_limit=20.0),
                    score_distribution=dict(mean=0, sigma=10),
                    convergence_criteria=dict(maximum_distance=50, maximum_angle=90))


def test_localised_racing_on_the_synthetic_circuit():
    from acmpc_amd import track_map, workloads
    from acmpc_amd.command_selection import TemporalCommandSelector
    from acmpc_amd.mpc import build_mpc, published_plan
    from acmpc_amd.particle_filter import ParticleFilter

    cfg = copy.deepcopy(workloads.RACING_CONTROL["monza"])
    mpc = build_mpc(cfg, workloads.PlaceholderVehicle())
    track = workloads.synthetic_track("monza")
    centre, left, right = track["centre"], track["left"], track["right"]
    M = len(centre)

    # race start: the whole-lap speed profile (agent.py:286-296 -> controller.py:49-57; configs/monza.yaml:82-84)
    lap = mpc.compute_map_speed_profile(mpc.construct_waypoints(track_map.lap_reference_path(centre)),
                                        ay_max=7.0, a_min=-0.15)
    reference_speeds = lap.velocities
    assert reference_speeds.shape == (M - 1,) and reference_speeds.max() > 40.0

    rng = np.random.default_rng(5)
    pf = ParticleFilter(LOCALISATION, dict(centre=centre, left=left, right=right), wheelbase=workloads.VEHICLE.wheelbase,
                        rng=rng)
    tangent = np.roll(centre, -1, axis=0) - np.roll(centre, 1, axis=0)
    heading = np.arctan2(tangent[:, 1], tangent[:, 0])
    start = int(np.argmax(reference_speeds[:4000]))            # somewhere fast
    pose = np.array([centre[start, 0], centre[start, 1], heading[start]])
    pf.states = (pose + rng.normal(0, [1.0, 1.0, 0.05], (300, 3))).astype(np.float32)
    pf.scores = np.full(300, 1.0 / 300, dtype=np.float32)

    def observe(limit, count, at, yaw, origin):
        pts = limit[(at + np.arange(count)) % M] - origin
        a = np.pi / 2 - yaw
        rot = np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]])
        return (pts @ rot.T + rng.normal(0, 0.15, (count, 2))).astype(np.float32)

    speed, dt, wheelbase = 20.0, 0.05, workloads.VEHICLE.wheelbase
    delta = 0.0
    localisation_error, lateral, speeds, v_max_used = [], [], [], []
    for tick in range(300):                                                              # 15 s
        at = int(np.argmin(((centre - pose[:2]) ** 2).sum(axis=1)))
        # localiser: move the particles with the applied control, score them against what the car sees
        pf.step(delta, speed, dt)
        pf.update({"left": observe(left, 230, at, pose[2], pose[:2]), "right": observe(right, 210, at, pose[2], pose[:2])})
        estimate = pf.estimated_location
        localisation_error.append(np.linalg.norm(estimate[:2] - pose[:2]))
        # agent: the localised reference speed (agent.py:127-143), written where the control loop reads it
        is_localised = pf.is_converged
        if is_localised:
            index = int(np.argmin(((centre - estimate[:2]) ** 2).sum(axis=1)))            # estimated_map_index
            v_max = track_map.reference_speed_window(reference_speeds, index)
        else:
            v_max = float(cfg["unlocalised_max_speed"])
        mpc.speed_profile_constraints["v_max"] = v_max                                    # controller.py:241-243
        v_max_used.append(v_max)
        # controller: perception's centreline from the true pose, one solve
        window = centre[(at + np.arange(301)) % M] - pose[:2]
        a = np.pi / 2 - pose[2]
        local = window @ np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]]).T
        t = np.linspace(0, 300, 500)
        line = np.stack([np.interp(t, np.arange(301), local[:, 0]), np.interp(t, np.arange(301), local[:, 1])], axis=1)
        mpc.get_control(workloads.reference_path_from_centreline(line, 50), is_localised)
        assert mpc.infeasibility_counter == 0, "infeasible solve at tick %d" % tick
        plan = published_plan(mpc)                                                         # controller.py:274-280
        assert plan.control_inputs.shape == (49, 2) and plan.predicted_locations.shape == (49, 2)
        v_cmd, delta = TemporalCommandSelector(plan)(float(mpc.cum_time[1]))
        delta = float(delta)
        speed += float(np.clip(v_cmd - speed, -10.0 * dt, 6.0 * dt))
        pose = pose + np.array([speed * np.cos(pose[2]), speed * np.sin(pose[2]),
                                speed * np.tan(delta) / wheelbase]) * dt
        lateral.append(np.linalg.norm(pose[:2] - centre[int(np.argmin(((centre - pose[:2]) ** 2).sum(axis=1)))]))
        speeds.append(speed)
    assert pf.is_converged and not pf.was_reset
    assert max(localisation_error) < 5.0, "filter lost the car: %.1f m" % max(localisation_error)
    assert max(lateral) < 3.0, "left the 9.5 m road: %.2f m" % max(lateral)
    assert max(v_max_used) > 28.5, "the localised speed window never lifted the cap"
    assert max(speeds) > 28.5, "the car never used the localised profile (max %.1f m/s)" % max(speeds)
