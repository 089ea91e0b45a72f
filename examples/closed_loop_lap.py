#!/usr/bin/env python3
"""A lap segment on a synthetic circuit with everything in the loop, wired the way the reference's agent wires it:

    lap speed profile at race start        agent.py:286-296 -> controller.py:49-57
    particle filter on track-limit points  localiser.py:41-77, 234-239          (GPU: scoring, kinematic step, estimate)
    localised reference-speed window       agent.py:137-143 -> controller.py:241-243
    MPC solve                              spatial_mpc.py:170-217               (GPU: sample, roll out, cost, argmin)
    command selection                      commands.py:20-38

Perception is replaced by the true track limits seen from the car plus noise; the car is a kinematic bicycle.
Needs an MI355X.      python examples/closed_loop_lap.py [--track monza] [--seconds 30]
"""
import argparse
import copy
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ac-mpc_amd"))
from acmpc_amd import track_map, workloads  # noqa: E402
from acmpc_amd.command_selection import TemporalCommandSelector  # noqa: E402
from acmpc_amd.mpc import build_mpc, published_plan  # noqa: E402
from acmpc_amd.particle_filter import ParticleFilter  # noqa: E402

LOCALISATION = dict(n_particles=500, n_converged_particles=500,                      # configs/monza.yaml:43-66
                    sampling_noise=dict(x=1.1, y=1.1, yaw=3.0), control_noise=dict(velocity=0.25, yaw=2.0),
                    thresholds=dict(offset=10, rotation=90, minimum_particles=20, track_limit=20.0),
                    score_distribution=dict(mean=0, sigma=10),
                    convergence_criteria=dict(maximum_distance=50, maximum_angle=90))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--track", default="monza", choices=sorted(workloads.TRACKS))
    ap.add_argument("--seconds", type=float, default=30.0)
    args = ap.parse_args()

    cfg = copy.deepcopy(workloads.RACING_CONTROL[args.track])
    mpc = build_mpc(cfg, workloads.PlaceholderVehicle())
    track = workloads.synthetic_track(args.track)
    centre, left, right = track["centre"], track["left"], track["right"]
    M = len(centre)
    t0 = time.perf_counter()
    lap = mpc.compute_map_speed_profile(mpc.construct_waypoints(track_map.lap_reference_path(centre)),
                                        ay_max=7.0, a_min=-0.15)
    reference_speeds = lap.velocities
    print("lap speed profile: %d waypoints in %.2f s, %.1f .. %.1f m/s" % (M, time.perf_counter() - t0,
                                                                           reference_speeds.min(), reference_speeds.max()))
    rng = np.random.default_rng(1)
    pf = ParticleFilter(LOCALISATION, dict(centre=centre, left=left, right=right), wheelbase=workloads.VEHICLE.wheelbase,
                        rng=rng)
    tangent = np.roll(centre, -1, axis=0) - np.roll(centre, 1, axis=0)
    heading = np.arctan2(tangent[:, 1], tangent[:, 0])
    pose = np.array([centre[0, 0], centre[0, 1], heading[0]])
    pf.states = (pose + rng.normal(0, [1.0, 1.0, 0.05], (pf.states.shape[0], 3))).astype(np.float32)

    def observe(limit, count, at):
        pts = limit[(at + np.arange(count)) % M] - pose[:2]
        a = np.pi / 2 - pose[2]
        rot = np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]])
        return (pts @ rot.T + rng.normal(0, 0.15, (count, 2))).astype(np.float32)

    speed, dt, wheelbase, delta = 15.0, 0.05, workloads.VEHICLE.wheelbase, 0.0
    solve_ms, lateral, error = [], [], []
    for tick in range(int(args.seconds / dt)):
        at = int(np.argmin(((centre - pose[:2]) ** 2).sum(axis=1)))
        pf.step(delta, speed, dt)
        pf.update({"left": observe(left, 230, at), "right": observe(right, 210, at)})
        estimate = pf.estimated_location
        error.append(np.linalg.norm(estimate[:2] - pose[:2]))
        if pf.is_converged:
            index = int(np.argmin(((centre - estimate[:2]) ** 2).sum(axis=1)))
            mpc.speed_profile_constraints["v_max"] = track_map.reference_speed_window(reference_speeds, index)
        else:
            mpc.speed_profile_constraints["v_max"] = float(cfg["unlocalised_max_speed"])
        window = centre[(at + np.arange(301)) % M] - pose[:2]
        a = np.pi / 2 - pose[2]
        local = window @ np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]]).T
        t = np.linspace(0, 300, 500)
        line = np.stack([np.interp(t, np.arange(301), local[:, 0]), np.interp(t, np.arange(301), local[:, 1])], axis=1)
        t0 = time.perf_counter()
        mpc.get_control(workloads.reference_path_from_centreline(line, cfg["horizon"]), pf.is_converged, elapsed=dt)
        solve_ms.append((time.perf_counter() - t0) * 1e3)
        v_cmd, delta = TemporalCommandSelector(published_plan(mpc))(float(mpc.cum_time[1]))
        delta = float(delta)
        speed += float(np.clip(v_cmd - speed, -10.0 * dt, 6.0 * dt))
        pose = pose + np.array([speed * np.cos(pose[2]), speed * np.sin(pose[2]), speed * np.tan(delta) / wheelbase]) * dt
        lateral.append(np.linalg.norm(pose[:2] - centre[int(np.argmin(((centre - pose[:2]) ** 2).sum(axis=1)))]))
        if tick % 100 == 99:
            print("t=%5.1f s  v=%5.1f m/s  v_max=%5.1f  lateral %.2f m  localisation error %.2f m  solve %.3f ms%s"
                  % ((tick + 1) * dt, speed, mpc.speed_profile_constraints["v_max"], lateral[-1], error[-1],
                     np.median(solve_ms[-100:]), "" if pf.is_converged else "  (not localised)"))
    print("done: %d solves, %d infeasible, solve p50 %.3f ms p99 %.3f ms, max lateral %.2f m, max localisation error %.2f m"
          % (len(solve_ms), mpc.infeasibility_counter, np.percentile(solve_ms, 50), np.percentile(solve_ms, 99),
             max(lateral), max(error)))


if __name__ == "__main__":
    main()
