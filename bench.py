"""Benchmark of the rollout-and-cost hot path (contract: see the task statement / DESIGN.md "Measurement").

    python bench.py --gpus N --steps K --warmup W      (N > 1 without torchrun: starts its own N ranks as a child)

One "step" = one pass of the hot path over one batch of synthetic input already resident in HBM: P poses round
the synthetic Monza circuit x 4 096 candidate control sequences x horizon 50 (BASELINE.json configs[1], batched
over poses so that one launch fills the GPU), i.e. rollout through the bicycle model + per-step cost + argmin +
winner record for every pose.  With N > 1 every rank (one process per GPU) evaluates its own 4 096 candidates of
every pose (weak scaling) and the ranks meet in one RCCL all-reduce(MIN) of the packed (cost, index) keys plus the
all-reduce(SUM) that distributes the winners' records.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import copy
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "ac-mpc_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--track", default="monza", choices=["monza", "spa", "nordschleife", "silverstone"])
    ap.add_argument("--horizon", type=int, default=50)
    ap.add_argument("--candidates", type=int, default=4096, help="candidates per pose per GPU")
    ap.add_argument("--poses", type=int, default=4096,
                    help="independent solves batched into one launch (16.8 M candidates, 6.6 GB of controls per step: "
                         "throughput and roofline fraction keep rising with the batch - 0.74 / 0.77 / 0.81 of roofline at "
                         "1024 / 2048 / 4096 poses - and 288 GB of HBM is there to be used)")
    ap.add_argument("--mode", default="S", choices=["S", "T"], help="S: spatial bicycle model, T: Cartesian + NN search")
    ap.add_argument("--nn-window", default=None, help="mode T: 'back,ahead' search window (default: exhaustive scan)")
    ap.add_argument("--layout", type=int, default=1, help="0: U[P][N][n][2], 1: U[P][n][2][N]")
    ap.add_argument("--buffers", type=int, default=2, help="distinct control matrices cycled through (HBM-cold reads)")
    ap.add_argument("--pipeline", action="store_true",
                    help="overlap argmin/record of batch i (side stream) with the rollout of batch i+1; measured "
                         "slower than plain stream order on MI355X (cross-queue dependencies cost ~10 us each)")
    ap.add_argument("--no-stream", action="store_true",
                    help="(one rank) argmin and records of every batch as a launch of their own behind its rollout "
                         "(acmpc_solve_sampled_device) instead of inside the NEXT batch's rollout launch, the last batch's "
                         "behind it (acmpc_solve_stream_device, the default: the finalize's 14 us and a launch boundary per "
                         "1.0 ms step leave the critical path, DESIGN 4.2)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (the real thing); gloo = rehearsal of the multi-rank path with the "
                         "collective payloads staged through the CPU (ranks may then share one GPU)")
    ap.add_argument("--collective", default="torch", choices=["torch", "capi"],
                    help="who carries the step's one all-reduce(MIN) of the packed keys: torch = torch.distributed (RCCL "
                         "through PyTorch; nothing to exchange on one rank); capi = the library's own entry point, "
                         "acmpc_reduce_across_ranks, on a communicator it makes itself (acmpc_rccl_unique_id on rank 0, shared "
                         "through the torch.distributed group, acmpc_rccl_comm_create on every rank) - also with ONE rank, "
                         "where it forces the multi-rank step (rollout, keys, reduce, finalize) through a one-rank communicator")
    ap.add_argument("--no-kernel-timing", action="store_true",
                    help="(diagnostic) attach no events to the rollout launches; roofline fields become null")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default): every rank evaluates its own --candidates of every pose.  strong: BASELINE.json "
                         "configs[3] as stated - Nordschleife, 262 144 candidates x horizon 80 in TOTAL, split over the ranks "
                         "(262 144 / N each), one all-reduce(MIN) of the packed keys per step; latency-bound by design")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--no-single-solve", action="store_true",
                    help="skip the extra 'single_solve' block (one 4096-candidate problem per launch: latency)")
    args = ap.parse_args()
    if args.scaling == "strong":   # configs[3]: the track, horizon and TOTAL candidate count are the configuration
        args.track, args.horizon, args.layout, args.mode = "nordschleife", 80, 1, "S"
        world = int(os.environ.get("WORLD_SIZE", "1"))
        args.total_candidates = 262144
        args.candidates = args.total_candidates // max(world, 1)
        if args.poses == 4096:     # (not given) one problem per step, as the controller would pose it
            args.poses = 1
    return args


SAMPLE_SIGMA = (2.0, 0.01)   # sigma = (2.0 m/s, 0.01 1/m)


def make_controls(batch, P, N, n, layout, device, seed):
    """u_ref + sigma * N(0,1), clipped to the input box, candidate 0 = u_ref, generated on
    the device with a seeded generator so no 400 MB host buffer is needed."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    u_ref = torch.tensor(np.stack([batch.tables[:, 6, :], batch.tables[:, 3, :]], axis=2), dtype=torch.float32,
                         device=device)  # [P,n,2]
    lo = torch.tensor(batch.u_lo, dtype=torch.float32, device=device)
    hi = torch.tensor(batch.u_hi, dtype=torch.float32, device=device)
    sigma = torch.tensor([2.0, 0.01], dtype=torch.float32, device=device)
    if layout == 1:
        U = torch.randn(P, n, 2, N, generator=g, device=device, dtype=torch.float32)
        U = U * sigma.view(1, 1, 2, 1) + u_ref.view(P, n, 2, 1)
        U = torch.maximum(torch.minimum(U, hi.view(1, 1, 2, 1)), lo.view(1, 1, 2, 1))
        U[:, :, :, 0] = torch.maximum(torch.minimum(u_ref, hi), lo)
    else:
        U = torch.randn(P, N, n, 2, generator=g, device=device, dtype=torch.float32)
        U = U * sigma + u_ref.view(P, 1, n, 2)
        U = torch.maximum(torch.minimum(U, hi), lo)
        U[:, 0] = torch.maximum(torch.minimum(u_ref, hi), lo)
    return U.contiguous()


def cpu_baseline(batch, mode, U_dev, layout, N, n, gpu_costs, seconds):
    """The oracle's C restatement (oracle/acmpc_oracle.c, OpenMP over candidates) timed on this box's host cores
    on a bounded sample of the same workload: the first poses of the batch, repeated until ~`seconds` of CPU
    work.  Also re-checks parity of the sample (costs bit-identical to the GPU's)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import c_oracle

    # the one-GPU box gives this job 16 host cores; never more threads than cores we may run on
    cores = min(16, len(os.sched_getaffinity(0)))
    os.environ["OMP_NUM_THREADS"] = str(cores)
    sample = min(32, U_dev.shape[0])   # 131 072 candidates per pass: enough blocks to keep 16 threads busy
    U = U_dev[:sample].cpu().numpy()
    cfg = batch.cfg
    w = c_oracle.make_weights(cfg["step_cost"], cfg["r_term"], cfg["final_cost"], batch.u_lo, batch.u_hi, 1.0e6)
    x0 = batch.x0 if mode == 0 else batch.pose0

    coef_host = np.stack(batch.coef_host[:sample])

    def one_pass(check):
        if mode == 0 and layout == 1:   # all sampled poses in one call: threads over (pose, 64-candidate block)
            costs, _ = c_oracle.rollout_spatial_batch(x0[:sample], coef_host, U, w)
        else:
            costs = np.stack([c_oracle.rollout(mode, x0[p], batch.coef_host[p], U[p], layout, w)[0] for p in range(sample)])
        for p in range(sample):
            c_oracle.argmin(costs[p])
            if check and not np.array_equal(costs[p], gpu_costs[p]):
                raise SystemExit("bench: GPU costs differ from the oracle on pose %d" % p)

    one_pass(check=True)
    repeats = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        one_pass(check=False)
        repeats += 1
    elapsed = time.perf_counter() - t0
    return {
        "value": sample * N * repeats / elapsed,
        "unit": "candidate-trajectories/s",
        "cores": cores,
        "kind": "port",
        "sample": "%d poses x %d candidates x horizon %d of the same batch, %d passes, %.1f s; oracle/acmpc_oracle.c "
                  "(gcc -O3 -mavx2, 8-wide SIMD across candidates, OpenMP threads), costs checked bit-identical to the GPU's" %
                  (sample, N, n + 1, repeats, elapsed),
    }


def numpy_baseline(batch, U_dev, layout, N, n, gpu_costs, seconds=3.0):
    """SURVEY.md 8(d)'s other CPU figure: the oracle's single-process NumPy restatement (vectorised over candidates,
    sequential over steps) on ONE pose of the same batch, repeated for ~`seconds`; parity of the pose re-checked."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import acmpc_oracle as orc
    U = U_dev[0].cpu().numpy()
    if layout == 1:
        U = np.ascontiguousarray(U.transpose(2, 0, 1))    # [N, n, 2]
    cfg = batch.cfg
    args = (batch.x0[0], batch.coef_host[0], U, cfg["step_cost"], cfg["r_term"], cfg["final_cost"], batch.u_lo, batch.u_hi,
            1.0e6)
    cost, _ = orc.rollout_spatial(*args, dtype=np.float32)
    if not np.array_equal(cost, gpu_costs[0]):
        raise SystemExit("bench: GPU costs differ from the NumPy oracle")
    repeats, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        orc.pick_best(orc.rollout_spatial(*args, dtype=np.float32)[0])
        repeats += 1
    elapsed = time.perf_counter() - t0
    return {"value": N * repeats / elapsed, "unit": "candidate-trajectories/s", "cores": 1, "kind": "port",
            "sample": "1 pose x %d candidates x horizon %d, %d passes, %.1f s; oracle/acmpc_oracle.py rollout_spatial "
                      "(NumPy float32, one process), costs bit-identical to the GPU's" % (N, n + 1, repeats, elapsed)}


_NUMPY_WORKER = r"""
import sys, time
import numpy as np
sys.path.insert(0, sys.argv[1])
import acmpc_oracle as orc
d = np.load(sys.argv[2])
lo, hi, seconds = int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5])
args = (d["x0"], d["coef"], np.ascontiguousarray(d["U"][lo:hi]), d["Q"], d["R"], d["QN"], d["u_lo"], d["u_hi"], 1.0e6)
best = orc.pick_best(orc.rollout_spatial(*args, dtype=np.float32)[0])
repeats, t0 = 0, time.perf_counter()
while time.perf_counter() - t0 < seconds:
    best = orc.pick_best(orc.rollout_spatial(*args, dtype=np.float32)[0])
    repeats += 1
print(repeats, time.perf_counter() - t0, best[0] + lo, repr(float(best[1])))
"""


def numpy_all_cores_baseline(batch, U_dev, layout, N, n, seconds=3.0):
    """SURVEY.md 8(d)'s second NumPy figure: as many processes as this job has cores, each rolling out its share of one
    pose's candidates with the oracle's NumPy restatement and taking its own argmin; the parent takes the argmin of
    those.  The workers are plain Python processes (no torch, no GPU)."""
    import subprocess
    import tempfile
    cores = min(16, len(os.sched_getaffinity(0)))
    U = U_dev[0].cpu().numpy()
    if layout == 1:
        U = np.ascontiguousarray(U.transpose(2, 0, 1))    # [N, n, 2]
    cfg = batch.cfg
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "pose.npz")
        np.savez(path, x0=batch.x0[0], coef=batch.coef_host[0], U=U, Q=np.asarray(cfg["step_cost"]),
                 R=np.asarray(cfg["r_term"]), QN=np.asarray(cfg["final_cost"]), u_lo=batch.u_lo, u_hi=batch.u_hi)
        env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
        bounds = [N * k // cores for k in range(cores + 1)]
        procs = [subprocess.Popen([sys.executable, "-c", _NUMPY_WORKER, os.path.join(ROOT, "oracle"), path,
                                   str(bounds[k]), str(bounds[k + 1]), str(seconds)], stdout=subprocess.PIPE, text=True,
                                  env=env) for k in range(cores)]
        outs = [p.communicate(timeout=120)[0].split() for p in procs]
    if any(p.returncode != 0 for p in procs):
        raise SystemExit("bench: a NumPy baseline worker failed")
    rate = sum((bounds[k + 1] - bounds[k]) * int(o[0]) / float(o[1]) for k, o in enumerate(outs))
    winner = min((float(o[3]), int(o[2])) for o in outs)
    return {"value": rate, "unit": "candidate-trajectories/s", "cores": cores, "kind": "port",
            "sample": "1 pose x %d candidates x horizon %d split over %d processes for %.1f s each; "
                      "oracle/acmpc_oracle.py rollout_spatial (NumPy float32) + argmin per process, argmin of those in "
                      "the parent (winner %d)" % (N, n + 1, cores, seconds, winner[1])}


def config5_host_pointer(workloads, Engine, device, poses=10000, N=4096, H=50, pinned=False):
    """BASELINE.json configs[4] exactly as SURVEY.md 8(d) defines it: 10 000 consecutive poses along the synthetic
    Silverstone circuit, one solve each of N = 4 096 candidates through the host-pointer `acmpc_solve` (H2D of the
    1.6 MB control matrix + rollout + argmin + record + D2H), wall-clock p50 / p99 per solve.  Candidates are
    u_ref + sigma * N(0, 1) clipped to the input box, candidate 0 = u_ref (one seeded noise matrix, re-centred per pose)."""
    import copy
    from acmpc_amd import _capi
    from acmpc_amd.bicycle_model import SpatialBicycleModel
    from acmpc_amd.mpc import waypoint_table
    name, n = "silverstone", H - 1
    cfg = copy.deepcopy(workloads.RACING_CONTROL[name])
    cons = dict(cfg["speed_profile_constraints"], v_max=float(cfg["unlocalised_max_speed"]))
    model = SpatialBicycleModel(workloads.PlaceholderVehicle(), {"min": cons["v_min"], "max": cons["v_max"]})
    track = workloads.synthetic_track(name)
    lo = np.array([model.min_u[0] - 0.1, model.min_u[1]], dtype=np.float32)
    hi = np.array([model.max_u[0] + 0.1, model.max_u[1]], dtype=np.float32)
    eng = Engine(mode=0, max_problems=1, max_candidates=N, max_steps=n, step_cost=cfg["step_cost"], r_term=cfg["r_term"],
                 final_cost=cfg["final_cost"], u_min=lo, u_max=hi, margin=model.margin, wheelbase=model.length,
                 device=device.index)
    noise = (np.random.default_rng(5).standard_normal((N, n, 2)) * np.array(SAMPLE_SIGMA)).astype(np.float32)
    noise[0] = 0.0
    U = np.empty((1, N, n, 2), dtype=np.float32)
    if pinned:   # the matrix built in page-locked memory (acmpc_host_alloc): the rollout reads it in place over the host link
        U = _capi.pinned_empty((1, N, n, 2), np.float32)
    warm, wall, infeasible = None, np.empty(poses), 0
    for i in range(poses + 20):
        coords = workloads.reference_path_from_centreline(workloads.local_centreline(track, (i * 2) % len(track["centre"])), H)
        table = waypoint_table(coords)
        v_hi = _capi.velocity_ceiling(table[3], cons["ay_max"], cons["ki_min"], cons["v_min"], cons["v_max"], False,
                                      cons["end_velocity"])
        v, y, status, _ = _capi.speed_profile_qp(v_hi, table[4], cons["a_min"], cons["a_max"], cons["v_min"], warm=warm)
        if status == "solved":
            warm, table[6] = (v, y), v
        eng.set_paths(table)
        x0 = model.t2s(table[:3, 0], np.array([0.0, 0.0, np.pi / 2])).astype(np.float32)[None]
        u_ref = np.stack([table[6], table[3]], axis=1).astype(np.float32)
        np.add(noise, u_ref, out=U[0])
        np.clip(U[0], lo, hi, out=U[0])
        t0 = time.perf_counter()
        out = eng.solve(x0, U, layout=0, want_costs=False)
        if i >= 20:
            wall[i - 20] = time.perf_counter() - t0
            infeasible += int(out["violation"][0] > 0.0)
    eng.close()
    return {"workload": "%s (synthetic circuit): %d consecutive poses 1 m apart, one acmpc_solve each (host pointers), %d "
                        "candidates x horizon %d" % (name, poses, N, H),
            "solve_us_p50": float(np.percentile(wall, 50) * 1e6), "solve_us_p99": float(np.percentile(wall, 99) * 1e6),
            "traj_per_s_at_p50": N / float(np.percentile(wall, 50)), "winners_with_a_bound_violation": infeasible,
            "controls_in": "page-locked host memory (acmpc_host_alloc)" if pinned else "pageable host memory",
            "note": "PCIe-inclusive (1.6 MB of controls per solve: staged into device memory from pageable memory, read in place "
                    "from page-locked memory): never `value`"}


def device_resident_filter(track, ticks=300):
    """The localiser's whole update cycle with the particles LIVING on the GPU (`DeviceParticleFilter`: kinematic step with
    control noise, scoring, exact integer inverse-CDF resampling, estimate - localiser.py:41-77,234-239,420-579; one host round
    trip per update): a car driving along the centre line at 20 m/s, the true track limits seen from it plus noise as the
    observation.  `update_us` includes the host's NumPy thinning of the observation (localiser.py:241-253, ~30 us)."""
    from acmpc_amd.particle_filter import DeviceParticleFilter
    centre, left, right = track["centre"], track["left"], track["right"]
    M = len(centre)
    tangent = np.roll(centre, -1, axis=0) - np.roll(centre, 1, axis=0)
    heading = np.arctan2(tangent[:, 1], tangent[:, 0])
    rng = np.random.default_rng(4)

    def observe(limit, count, at, pose):
        a = np.pi / 2 - pose[2]
        rot = np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]])
        return ((limit[(at + np.arange(count)) % M] - pose[:2]) @ rot.T + rng.normal(0, 0.15, (count, 2))).astype(np.float32)

    out = {}
    for n in (500, 100000):
        cfg = dict(n_particles=n, n_converged_particles=n, score_distribution=dict(mean=0, sigma=10),
                   thresholds=dict(offset=10, rotation=90, minimum_particles=20, track_limit=20.0),
                   sampling_noise=dict(x=1.1, y=1.1, yaw=3.0), control_noise=dict(velocity=0.25, yaw=2.0),
                   convergence_criteria=dict(maximum_distance=50, maximum_angle=90))
        pf = DeviceParticleFilter(cfg, dict(centre=centre, left=left, right=right), seed=3)
        start = np.array([centre[0, 0], centre[0, 1], heading[0]])
        pf.set_particles((start + rng.normal(0, [1.0, 1.0, 0.05], (n, 3))).astype(np.float32), np.full(n, 1.0 / n, dtype=np.float32))
        t_step, t_update, resets = [], [], 0
        for tick in range(ticks + 50):
            pose = np.array([centre[tick % M, 0], centre[tick % M, 1], heading[tick % M]])
            obs = {"left": observe(left, 230, tick % M, pose), "right": observe(right, 210, tick % M, pose)}
            t0 = time.perf_counter()
            pf.step(0.0, 20.0, 0.05)
            t1 = time.perf_counter()
            res = pf.update(obs)
            t2 = time.perf_counter()
            if tick >= 50:
                t_step.append(t1 - t0)
                t_update.append(t2 - t1)
                resets += int(res["was_reset"])
        out["particles_%d" % n] = {"step_us_p50": float(np.median(t_step) * 1e6), "update_us_p50": float(np.median(t_update) * 1e6),
                                   "update_us_p99": float(np.percentile(t_update, 99) * 1e6), "resets": resets,
                                   "resampling": "one workgroup" if n < 8192 else "tiled launches (capacity >= 8 192)"}
        pf.scorer.close()
    return out


def particle_filter_block(workloads, iters=20):
    """the localiser's particle scoring (three nearest-point queries against the ~11.6 k-point map
    polylines, observation placement, score) through the host-pointer seam, at the reference's size (500 particles,
    configs/monza.yaml:47) and at 100 000."""
    from acmpc_amd.particle_filter import ParticleScorer
    track = workloads.synthetic_track("monza")
    cfg = dict(n_particles=100000, score_distribution=dict(mean=0, sigma=10),
               thresholds=dict(offset=10, rotation=90, minimum_particles=20, track_limit=20.0))
    scorer = ParticleScorer(cfg, track)
    rng = np.random.default_rng(0)
    centre = track["centre"]
    left = np.stack([-4.7 + rng.normal(0, 0.1, 100), np.linspace(0, 49, 100)], axis=1).astype(np.float32)
    right = np.stack([4.7 + rng.normal(0, 0.1, 100), np.linspace(0, 49, 100)], axis=1).astype(np.float32)
    out = {"map_points": int(3 * len(centre)), "observation_points": 200}
    for P in (500, 100000):
        seeds = rng.integers(0, len(centre), P)
        states = np.concatenate([centre[seeds] + rng.normal(0, 2.0, (P, 2)), rng.uniform(-3, 3, (P, 1))],
                                axis=1).astype(np.float32)
        scorer.update_particles(states, [left, right])
        t = []
        for _ in range(iters):
            t0 = time.perf_counter()
            scorer.update_particles(states, [left, right])
            t.append(time.perf_counter() - t0)
        us = float(np.median(t) * 1e6)
        entry = {"update_us": us, "particles_per_s": P / (us * 1e-6)}
        # the three nearest points per particle through the uniform grid (exact: certified blocks of cells, a wave-wide scan
        # for what they do not settle): inside the scoring workgroup below 4 096 particles (a wavefront per polyline), as a
        # launch of its own in front of the scoring above (sixteen lanes per query)
        entry.update(nearest_point_search="uniform grid, exact; " + ("inside the scoring launch" if P < 4096 else
                                                                      "pf_nearest_kernel in front of the scoring launch"))
        out["particles_%d" % P] = entry
    scorer.close()
    out["device_resident_filter"] = device_resident_filter(track)
    # What bounds the kernels: kernel durations from the committed rocprofv3 summary of tools/bench_pf.py
    # (profiles/*_pf_kernel_stats.json, tools/profile_pf.sh) and - round 5 - the counters of the same command
    # (profiles/*_pf_sq_counters.json, tools/pmc_pf.sh): the cache LINES the kernel's vector loads ask the per-CU vector
    # caches for (TCP_TOTAL_CACHE_ACCESSES) against what 256 of them serve - a 64-byte line per clock each at the ~2.1 GHz
    # the chip holds, 34.4 TB/s, the guide's L2 -> L1 figure.  Round 4's searches (one lane per query walking its own cells)
    # asked for 16 lines per load instruction and sat at 0.59 of that rate; round 5's (consecutive lanes on consecutive
    # points) ask for a third of the lines and are bound by the length of a query's chain of dependent loads.
    stats, stats_path = newest_profile("pf_kernel_stats.json")
    counters, counters_path = newest_profile("pf_sq_counters.json")
    line_rate = 256 * 64 * 2.1e9
    if stats is not None:
        bounds = {"source": stats_path, "counters_source": counters_path, "measured_in_this_run": False,
                  "vector_cache_line_rate_TBps": line_rate / 1e12}
        runs = (counters or {}).get("runs", {})
        newest = [run for label, run in sorted(runs.items()) if label.endswith("new")]
        for label, size, kernel in (("pf_score_kernel<1>_at_500", "500", "pf_score_kernel<1>"),
                                    ("pf_nearest_kernel_at_100000", "100000", "pf_nearest_kernel"),
                                    ("pf_score_kernel<8>_at_100000", "100000", "pf_score_kernel<8>"),
                                    ("pf_score_given_kernel_at_100000", "100000", "pf_score_given_kernel")):
            timed = stats["sizes"].get(size, {}).get(kernel)
            if timed is None:
                continue
            entry = {"kernel_us": timed["average_us"]}
            counted = newest[-1]["sizes"].get(size, {}).get(kernel) if newest else None
            if counted is not None and "derived" in counted and "vector_cache_line_accesses" in counted["derived"]:
                d = counted["derived"]
                lines = d["vector_cache_line_accesses"]
                entry.update(vector_cache_line_accesses=lines,
                             line_accesses_per_read_instruction=d["vector_cache_line_accesses_per_read_instruction"],
                             l2_hit_rate=d["l2_hit_rate"],
                             share_of_wave_cycles_waiting=d["share_of_wave_cycles"]["waiting_on_memory_or_barrier (SQ_WAIT_ANY)"],
                             share_of_wave_cycles_issue_stalled=d["share_of_wave_cycles"][
                                 "issue_stalled (SQ_WAIT_INST_ANY: the vector-memory queue is full)"],
                             frac_of_vector_cache_line_rate=lines * 64 / (timed["average_us"] * 1e-6) / line_rate)
            if kernel == "pf_nearest_kernel":
                entry["queries_per_s"] = 300000 / (timed["average_us"] * 1e-6)
            if counted is not None and "derived" in counted and "valu_instructions_per_wave" in counted["derived"] \
                    and "waves" in counted["derived"]:
                # instruction issue: a wave64 vector instruction holds its SIMD's issue port for four clocks
                issue = counted["derived"]["valu_instructions_per_wave"] * counted["derived"]["waves"] * 4 / (1024 * 2.1e9)
                entry["vector_issue_floor_us"] = issue * 1e6
                entry["frac_of_vector_issue_rate"] = issue / (timed["average_us"] * 1e-6)
            bounds[label] = entry
        out["kernel_bounds"] = bounds
    return out


SECONDARY_SCALARS = ("mode_T_window_2_5_frac_1M", "mode_T_window_2_5_frac_16M", "mode_T_exhaustive_frac_1M",
                     "mode_T_exhaustive_frac_16M", "mode_T_window_1_2_frac_1M", "mode_T_valu_frac_window_2_5_1M",
                     "mode_T_valu_frac_window_2_5_16M", "mode_T_valu_frac_exhaustive_1M", "mode_T_valu_frac_exhaustive_16M",
                     "mode_T_window_2_5_us_1M", "mode_T_exhaustive_us_1M", "mode_T_window_2_5_late_workgroups_1M",
                     "mode_T_exhaustive_late_workgroups_1M", "mode_S_candidate_major_frac_1M",
                     "sampled_fused_16M_traj_per_s", "single_solve_us_p50", "config3_single_us_p50", "config4_share_us_p50",
                     "config5_us_p50", "config5_us_p99", "config5_pinned_us_p50", "config5_pinned_us_p99", "tick_ms_p50",
                     "tick_ms_p99", "tick_infeasible_solves", "tick_mode_T_ms_p50", "tick_mode_T_ms_p99",
                     "tick_mode_T_window_2_5_ms_p50", "pf_update_us_500", "pf_update_us_100000", "pf_filter_update_us_500",
                     "pf_filter_update_us_100000")


def secondary_scalars(out):
    """Scalar copies of the secondary results, for `roofline` (see main)."""
    second = out["secondary_kernels"]

    def frac(name, key="frac_of_hbm_roofline"):
        return second.get(name, {}).get(key)

    def valu(name):
        return second.get(name, {}).get("roofline_valu", {}).get("frac")

    pf = out.get("particle_filter", {})
    scalars = {
        "mode_T_window_2_5_frac_1M": frac("mode_T_window_2_5"), "mode_T_window_2_5_frac_16M": frac("mode_T_window_2_5_16M"),
        "mode_T_exhaustive_frac_1M": frac("mode_T_exhaustive_search"),
        "mode_T_exhaustive_frac_16M": frac("mode_T_exhaustive_search_16M"),
        "mode_T_window_1_2_frac_1M": frac("mode_T_window_1_2"),
        "mode_T_valu_frac_window_2_5_1M": valu("mode_T_window_2_5"), "mode_T_valu_frac_window_2_5_16M": valu("mode_T_window_2_5_16M"),
        "mode_T_valu_frac_exhaustive_1M": valu("mode_T_exhaustive_search"),
        "mode_T_valu_frac_exhaustive_16M": valu("mode_T_exhaustive_search_16M"),
        "mode_T_window_2_5_us_1M": frac("mode_T_window_2_5", "kernel_us"),
        "mode_T_exhaustive_us_1M": frac("mode_T_exhaustive_search", "kernel_us"),
        "mode_T_window_2_5_late_workgroups_1M": frac("mode_T_window_2_5", "workgroups_started_more_than_10us_late"),
        "mode_T_exhaustive_late_workgroups_1M": frac("mode_T_exhaustive_search", "workgroups_started_more_than_10us_late"),
        "mode_S_candidate_major_frac_1M": frac("mode_S_candidate_major"),
        "sampled_fused_16M_traj_per_s": frac("sampled_fused_16M", "candidate_trajectories_per_s"),
        "single_solve_us_p50": out.get("single_solve", {}).get("device_resident_us_p50"),
        "config3_single_us_p50": out.get("config3_single", {}).get("device_resident_us_p50"),
        "config4_share_us_p50": out.get("config4_share", {}).get("device_resident_us_p50"),
        "config5_us_p50": out.get("config5_host_pointer", {}).get("solve_us_p50"),
        "config5_us_p99": out.get("config5_host_pointer", {}).get("solve_us_p99"),
        "config5_pinned_us_p50": out.get("config5_host_pointer", {}).get("pinned", {}).get("solve_us_p50"),
        "config5_pinned_us_p99": out.get("config5_host_pointer", {}).get("pinned", {}).get("solve_us_p99"),
        "tick_ms_p50": out.get("closed_loop_replay", {}).get("solve_ms_p50"),
        "tick_ms_p99": out.get("closed_loop_replay", {}).get("solve_ms_p99"),
        "tick_infeasible_solves": out.get("closed_loop_replay", {}).get("infeasible_solves"),
        "tick_mode_T_ms_p50": out.get("closed_loop_replay_mode_T", {}).get("solve_ms_p50"),
        "tick_mode_T_ms_p99": out.get("closed_loop_replay_mode_T", {}).get("solve_ms_p99"),
        "tick_mode_T_window_2_5_ms_p50": out.get("closed_loop_replay_mode_T_window_2_5", {}).get("solve_ms_p50"),
        "pf_update_us_500": pf.get("particles_500", {}).get("update_us"),
        "pf_update_us_100000": pf.get("particles_100000", {}).get("update_us"),
        "pf_filter_update_us_500": pf.get("device_resident_filter", {}).get("particles_500", {}).get("update_us_p50"),
        "pf_filter_update_us_100000": pf.get("device_resident_filter", {}).get("particles_100000", {}).get("update_us_p50"),
    }
    assert set(scalars) == set(SECONDARY_SCALARS)
    return scalars


def latest_traffic(algorithmic_bytes, kernel_name):
    """HBM bytes per launch of the dominant kernel from the newest committed PMC summary for this exact workload
    (profiles/*_summary.json, produced by tools/profile.sh + tools/summarize_profile.py: separate --pmc passes,
    FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md).  None when no matching profile exists."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_summary.json"))):
        try:
            summary = json.load(open(path))
        except (OSError, ValueError):
            continue
        if summary.get("algorithmic_bytes_per_launch") == algorithmic_bytes and \
                kernel_name in summary.get("dominant_kernel", "") and "hbm_traffic_bytes_per_launch" in summary:
            best = (summary["hbm_traffic_bytes_per_launch"], os.path.relpath(path, ROOT))
    return best


SIMDS = 1024                 # 256 CUs x 4 SIMDs
LDS_ARRAY_CYCLES = {"ds_read_b128": 4, "ds_read_b96": 8, "ds_read_b64": 2, "ds_read_b32": 2, "ds_read2_b32": 4,
                    "ds_read2_b64": 8}   # LDS-array cycles per conflict-free wave64 instruction (MI355X_MICROARCH.md, LDS table)
LDS_CLOCK_HZ = 2.4e9
# opcodes that issue like v_add_f32 (two cycles per wave64 at full occupancy) when tools/valu_probe.hip has no row of their own
CHEAP_LIKE = ("v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fmac_f32", "v_mov_b32", "v_xor_b32", "v_and_b32",
              "v_or_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_fmamk_f32", "v_fmaak_f32")


def newest_profile(suffix):
    """The newest committed profiles/*_<suffix> as (dict, relative path), or (None, None)."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_" + suffix)), reverse=True):
        try:
            return json.load(open(path)), os.path.relpath(path, ROOT)
        except (OSError, ValueError):
            continue
    return None, None


def counted_valu(profile, entry):
    """VALU instructions per candidate-step and LDS instructions per wave-step of a kernel from the newest committed
    SQ-counter summary (profiles/*_<profile>_sq_counters.json: SQ_INSTS_VALU / waves / steps / candidates per lane,
    collected with rocprofv3 --pmc in its own run).  None when no such profile is committed."""
    import glob
    found = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_%s_sq_counters.json" % profile))):
        try:
            derived = json.load(open(path))[entry]["derived"]
        except (OSError, ValueError, KeyError):
            continue
        if derived.get("valu_instructions_per_candidate_step"):
            derived = dict(derived, source_sha256=json.load(open(path)).get("source_sha256"))
            found = (float(derived["valu_instructions_per_candidate_step"]), os.path.relpath(path, ROOT), derived)
    return found


# opcode prefix -> the row of tools/valu_probe.hip that stands for its class (checked in this order, after an exact match)
ISSUE_CLASSES = (("v_cmp", "v_cmp_eq_f32_e64"), ("v_cndmask", "v_cndmask_b32_e64"), ("v_min3", "v_min3_f32"),
                 ("v_max3", "v_min3_f32"), ("v_med3", "v_med3_f32"), ("v_min_", "v_min_i32"), ("v_max_", "v_max_f32"),
                 ("v_lshl_add", "v_lshl_add_u32"), ("v_lshlrev", "v_lshlrev_b32"), ("v_lshrrev", "v_lshlrev_b32"),
                 ("v_ashrrev", "v_lshlrev_b32"), ("v_add3", "v_add3_u32"), ("v_mad_u64", "v_mad_u64_u32"),
                 ("v_pk_fma", "v_pk_fma_f32"), ("v_pk_", "v_pk_mul_f32"), ("v_fma_f32", "v_fma_f32"),
                 ("v_log", "v_log_f32"), ("v_exp", "v_log_f32"), ("v_sin", "v_sin_f32"), ("v_cos", "v_sin_f32"),
                 ("v_sqrt", "v_sqrt_f32"), ("v_rsq", "v_sqrt_f32"), ("v_rcp", "v_rcp_f32"))


def issue_cost_ns(opcode, probe):
    """Issue time of one wave64 instruction on one SIMD at eight waves per SIMD, independent instructions
    (tools/valu_probe.hip -> profiles/*_valu_probe.json, wall-clock ns: the clock the chip really ran at is in it).
    An opcode is priced by its own row, else by its class's row (ISSUE_CLASSES), else like the CHEAPEST class
    (v_add_f32) - the roof must not overstate what the kernel has to issue."""
    table = probe["instructions"]
    name = opcode
    for suffix in ("_e32", "_e64", "_dpp", "_sdwa"):
        if name.endswith(suffix):
            name = name[:-len(suffix)]
    for key in (opcode, name):
        if key in table:
            return table[key]["W8"]["ns_per_simd"]
    for prefix, row in ISSUE_CLASSES:
        if name.startswith(prefix) and row in table:
            return table[row]["W8"]["ns_per_simd"]
    return table["v_add_f32"]["W8"]["ns_per_simd"]


def loaded_source_hash():
    """sha256 of the library's sources and flags as tools/isa_mix.py records it: does a committed mix describe THIS build?"""
    import hashlib
    from acmpc_amd import _build
    h = hashlib.sha256()
    try:
        for name in sorted(_build.SOURCES + _build.HEADERS):
            with open(os.path.join(_build.CSRC_DIR, name), "rb") as handle:
                h.update(handle.read())
    except OSError:
        return None
    h.update(_build.flag_record().encode())
    return h.hexdigest()


def valu_roofline(counted, candidates, steps, kernel_s, mix_entry=None, cpt=2):
    """The issue-time roof of an instruction-bound kernel.  Its vector instructions - COUNTED by SQ_INSTS_VALU per
    candidate-step - cost what their opcode classes cost to issue at the kernel's own occupancy: the opcode mix of the
    step loop (tools/isa_mix.py, from the compiler's assembly) priced per opcode with tools/valu_probe.hip's measurement at
    eight waves per SIMD (v_add / v_mul / v_fmac ~2 cycles, v_fma_f32 ~3, compares / selects / min / max / med3 / integer
    shift-adds ~4, transcendentals ~8).  peak = 1 024 SIMDs / (mean issue time of the mix); frac <= 1 by construction as
    long as the kernel issues what was counted.  `roofline_lds` beside it: the LDS array's cycles for the loop's reads."""
    if counted is None:
        return None
    per_step, source, derived = counted
    probe, probe_path = newest_profile("valu_probe.json")
    mix, mix_path = newest_profile("isa_mix.json")
    if probe is None or mix is None or mix_entry not in mix.get("entries", {}):
        return None
    entry = mix["entries"][mix_entry]
    total = float(sum(entry["valu"].values()))
    mean_ns = sum(count * issue_cost_ns(op, probe) for op, count in entry["valu"].items()) / total
    needed = per_step * candidates * steps / 64.0          # wave64 vector instructions of the launch
    peak = SIMDS / (mean_ns * 1e-9)
    out = {"bound": "valu_issue", "valu_instructions_per_candidate_step": per_step, "achieved": needed / kernel_s,
           "peak": peak, "unit": "wave-instructions/s", "frac": needed / kernel_s / peak,
           "mean_issue_ns_per_instruction_per_simd": mean_ns,
           "peak_is": "1 024 SIMDs / the mean issue time of the step loop's opcode mix at eight waves per SIMD "
                      "(per-opcode wall-clock issue times of tools/valu_probe.hip x the static mix of tools/isa_mix.py)",
           "instruction_count_source": source, "instruction_count_measured_in_this_run": False,
           # (the counters' summary names the sources they were collected from; None: an older summary without the record)
           "instruction_count_matches_loaded_sources": (derived.get("source_sha256") == loaded_source_hash()
                                                        if derived.get("source_sha256") else None),
           "opcode_mix_source": mix_path, "issue_cost_source": probe_path,
           "opcode_mix_matches_loaded_sources": mix.get("source_sha256") == loaded_source_hash()}
    lds_cycles = sum(count * LDS_ARRAY_CYCLES.get(op, 4) for op, count in entry["lds"].items())
    wave_steps = candidates * steps / (64.0 * cpt)
    lds_needed = lds_cycles * wave_steps
    lds_peak = 256 * LDS_CLOCK_HZ
    out_lds = {"bound": "lds_array", "lds_instructions_per_wave_step": float(sum(entry["lds"].values())),
               "lds_array_cycles_per_wave_step": lds_cycles, "achieved": lds_needed / kernel_s, "peak": lds_peak,
               "unit": "LDS-array cycles/s", "frac": lds_needed / kernel_s / lds_peak,
               "counted_lds_instructions_per_wave_step": derived.get("lds_instructions_per_wave_step"),
               "peak_is": "256 CUs x 2.4 GHz, one LDS array per CU; cycles per instruction from the guide's LDS table "
                          "(ds_read_b128 4, ds_read_b96 8, ds_read2_b32 4: conflict-free)"}
    return out, out_lds


def single_solve(workloads, Engine, track, H, N, mode, layout, device, iters=300, host_pointer=True):
    """BASELINE.json configs[1] as ONE problem per call (the closed-loop shape): device-resident solve latency
    (rollout + argmin + winner record, HIP events) and the host-pointer acmpc_solve (H2D + kernels + D2H)."""
    n = H - 1
    batch = workloads.problem_batch(track, 1, H, seed=1)
    eng = Engine(**workloads.engine_kwargs(batch, mode, N, device=device.index))
    eng.set_paths(batch.tables)
    stream = torch.cuda.current_stream().cuda_stream
    eng.sync_tables(stream)
    U = make_controls(batch, 1, N, n, layout, device, seed=7)
    x0 = torch.tensor(batch.x0 if mode == 0 else batch.pose0, device=device)
    from acmpc_amd._capi import record_floats
    keys = torch.empty(1, dtype=torch.int64, device=device)
    rec = torch.empty(1, record_floats(n), device=device)
    costs = torch.empty(1, N, device=device)

    def call():
        eng.solve_device(x0.data_ptr(), U.data_ptr(), 1, N, n, layout, costs.data_ptr(), keys.data_ptr(),
                         rec.data_ptr(), stream)

    for _ in range(20):
        call()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record()
        call()
        b.record()
    torch.cuda.synchronize()
    dev_us = np.array([a.elapsed_time(b) for a, b in ev]) * 1e3
    p50 = float(np.percentile(dev_us, 50))
    if not host_pointer:
        eng.close()
        byts = N * (8 * n + 4)
        return {"workload": "%s: 1 solve x %d candidates x horizon %d per call, device-resident" % (track, N, H),
                "device_resident_us_p50": p50, "device_resident_us_p99": float(np.percentile(dev_us, 99)),
                "device_resident_traj_per_s": N / (p50 * 1e-6),
                "frac_of_hbm_roofline": byts / (p50 * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                "note": "one problem per call: rollout + argmin + winner record in ONE launch (rollout_solo_kernel) up to "
                        "1 024 workgroups of 64 candidates, two launches beyond; latency-bound"}
    U_host, x0_host = U.cpu().numpy(), x0.cpu().numpy()
    wall = []
    for _ in range(20):
        eng.solve(x0_host, U_host, layout=layout, want_costs=False)
    for _ in range(iters):
        t0 = time.perf_counter()
        eng.solve(x0_host, U_host, layout=layout, want_costs=False)
        wall.append((time.perf_counter() - t0) * 1e6)
    wall = np.array(wall)
    out = {
        "workload": "%s: 1 solve x %d candidates x horizon %d per call" % (track, N, H),
        "device_resident_us_p50": float(np.percentile(dev_us, 50)), "device_resident_us_p99": float(np.percentile(dev_us, 99)),
        "device_resident_traj_per_s": N / (np.percentile(dev_us, 50) * 1e-6),
        "host_pointer_us_p50": float(np.percentile(wall, 50)), "host_pointer_us_p99": float(np.percentile(wall, 99)),
        "host_pointer_traj_per_s": N / (np.percentile(wall, 50) * 1e-6),
        "note": "host-pointer figures include PCIe H2D of the %.1f MB control matrix and D2H of the record" % (U_host.nbytes / 1e6),
    }
    eng.close()
    return out


def closed_loop_replay(workloads, track_name="silverstone", poses=2000, rollout_mode="S", nn_window="default"):
    """BASELINE.json configs[4]: the drop-in controller against consecutive poses along the synthetic Silverstone
    circuit - one full MPC solve per pose, each ONE call into the library: the reference path is cut out of the map on
    the device (150 m window in the vehicle frame, 500 -> H downsample), then waypoints, speed-profile QP, Frenet start
    state and linearisation (the prologue kernel), then the solver's default rounds x candidates sampled, rolled out
    and reduced - wall-clock p50/p99 per solve.  `host_path_*`: the same loop with the path built in NumPy and passed
    to `get_control` (the reference's call), for comparison."""
    import copy
    from acmpc_amd.mpc import build_mpc

    cfg = copy.deepcopy(workloads.RACING_CONTROL[track_name])
    cfg["speed_profile_constraints"]["v_max"] = float(cfg["unlocalised_max_speed"])  # controller.py:241-243
    cfg["rollout_mode"] = rollout_mode
    if nn_window != "default":   # mode T: the default is the nearest of ALL waypoints at every step; (back, ahead) = a window
        cfg["nn_window"] = nn_window
    track = workloads.synthetic_track(track_name)
    H = cfg["horizon"]
    stride = 2  # 1 m between consecutive poses at 0.5 m map spacing (~30 m/s at 30 Hz perception)
    M = len(track["centre"])

    mpc = build_mpc(copy.deepcopy(cfg), workloads.PlaceholderVehicle())
    mpc.bind_map(track)
    for i in range(20):
        mpc.get_control_at(map_index=(i * stride) % M)
    wall = np.empty(poses)
    for i in range(poses):
        t0 = time.perf_counter()
        mpc.get_control_at(map_index=((i + 20) * stride) % M)
        wall[i] = time.perf_counter() - t0

    host = build_mpc(copy.deepcopy(cfg), workloads.PlaceholderVehicle())
    count = min(poses, 500)
    paths = [workloads.reference_path_from_centreline(workloads.local_centreline(track, (i * stride) % M), H)
             for i in range(count + 20)]
    for p in paths[:20]:
        host.get_control(p)
    host_wall = np.empty(count)
    for i, p in enumerate(paths[20:]):
        t0 = time.perf_counter()
        host.get_control(p)
        host_wall[i] = time.perf_counter() - t0
    return {
        "workload": "%s (synthetic circuit): %d consecutive poses, one SpatialMPC solve per pose (path from the map, "
                    "prologue and rounds on the device), horizon %d, %d candidates x %d rounds, rollout mode %s"
                    % (track_name, poses, H, mpc._control_solver._n_candidates, mpc._control_solver._rounds,
                       rollout_mode + (", nearest waypoint searched in nn_window %s" % (nn_window,) if rollout_mode == "T" and
                                       nn_window not in ("default", None) else
                                       ", nearest waypoint of all at every step" if rollout_mode == "T" else "")),
        "solve_ms_p50": float(np.percentile(wall, 50) * 1e3), "solve_ms_p99": float(np.percentile(wall, 99) * 1e3),
        "solve_ms_max": float(wall.max() * 1e3), "sustainable_hz": float(1.0 / np.percentile(wall, 99)),
        "infeasible_solves": int(mpc.infeasibility_counter),
        "host_path_solve_ms_p50": float(np.percentile(host_wall, 50) * 1e3),
        "host_path_solve_ms_p99": float(np.percentile(host_wall, 99) * 1e3),
        "host_path_infeasible_solves": int(host.infeasibility_counter),
    }


def secondary_kernels(workloads, Engine, track, H, N, device, iters=20):
    """The other kernels of the path on the same kind of batch, timed alone with dispatch-attached HIP events:
    mode T (Cartesian rollout + nearest-waypoint projection; exhaustive and windowed search) and the
    candidate-major layout.  256 poses x N candidates each (1 M candidates per launch: one generation of waves, launch
    ramp and tail included), and - `*_16M` - the two that are furthest from the roofline again at the headline's own
    batch of 4 096 poses (their steady state), mode T with both searches."""
    n = H - 1
    base = workloads.problem_batch(track, 256, H, seed=0)
    stream = torch.cuda.current_stream().cuda_stream
    out = {}
    cases = [("mode_S_candidate_major", 0, 0, None, 256), ("mode_T_exhaustive_search", 1, 1, None, 256),
             ("mode_T_window_2_5", 1, 1, (2, 5), 256), ("mode_T_window_1_2", 1, 1, (1, 2), 256),
             ("mode_S_step_major_1M", 0, 1, None, 256),
             ("mode_T_window_2_5_16M", 1, 1, (2, 5), 4096), ("mode_T_exhaustive_search_16M", 1, 1, None, 4096),
             ("mode_S_candidate_major_16M", 0, 0, None, 4096)]
    for name, mode, layout, window, P in cases:
        batch = copy.copy(base)
        reps = P // 256
        batch.tables, batch.x0, batch.pose0 = (np.tile(base.tables, (reps, 1, 1)), np.tile(base.x0, (reps, 1)),
                                               np.tile(base.pose0, (reps, 1)))
        u_ref = torch.tensor(np.stack([batch.tables[:, 6, :], batch.tables[:, 3, :]], axis=2), dtype=torch.float32,
                             device=device).contiguous()
        costs = torch.empty(P, N, device=device)
        eng = Engine(**workloads.engine_kwargs(batch, mode, N, device=device.index, nn_window=window))
        eng.set_paths(batch.tables)
        eng.sync_tables(stream)
        x0 = torch.tensor(batch.x0 if mode == 0 else batch.pose0, device=device)
        U = torch.empty((P, n, 2, N) if layout == 1 else (P, N, n, 2), device=device)
        eng.sample_device(u_ref.data_ptr(), 2 * n, u_ref.data_ptr(), P, N, n, layout, 0, SAMPLE_SIGMA, 77, 0,
                          U.data_ptr(), stream)
        for _ in range(3):
            eng.rollout_device(x0.data_ptr(), U.data_ptr(), P, N, n, layout, 0, costs.data_ptr(), 0, stream)
        eng.profile_enable(iters)
        for _ in range(iters):
            eng.rollout_device(x0.data_ptr(), U.data_ptr(), P, N, n, layout, 0, costs.data_ptr(), 0, stream)
        torch.cuda.synchronize()
        ms = float(np.median(eng.profile_collect()))
        byts = P * N * (8 * n + 4)
        out[name] = {"kernel_us": ms * 1e3, "candidate_trajectories_per_s": P * N / (ms * 1e-3),
                     "algorithmic_GBps": byts / (ms * 1e-3) / 1e9, "frac_of_hbm_roofline": byts / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS}
        if mode == 1:   # instruction-bound: the roof it is under is the vector pipes', not HBM's
            entry = "exhaustive" if window is None else "window_%d_%d" % window
            roofs = valu_roofline(counted_valu("mode_T", entry), P * N, n, ms * 1e-3, mix_entry=entry)
            if roofs is not None:
                out[name]["roofline_valu"], out[name]["roofline_lds"] = roofs
            if P * N <= 128 * 8 * 1024:
                # one generation of waves (two candidates per lane, eight waves on each of 1 024 SIMDs): every workgroup fits
                # from the start, so all of them should START within a couple of microseconds - the library's own check that
                # the dispatcher dealt them evenly (acmpc_rollout_start_clocks; round 4's launch of the 8-waypoint window had
                # 12-60 workgroups wait 65 us for a compute unit while others ran seven: DESIGN 4.1).  Untimed extra launches.
                eng.profile_enable(0)
                eng.set_option("ACMPC_START_CLOCKS", "1")
                late = []
                for _ in range(6):
                    eng.rollout_device(x0.data_ptr(), U.data_ptr(), P, N, n, layout, 0, costs.data_ptr(), 0, stream)
                    starts = eng.rollout_start_clocks()
                    late.append(int((starts > 10.0).sum()))
                # (the first launch after the option is set allocates and clears the stamp buffer in front of the kernel:
                # not counted, as in tests/test_gpu_parity.py)
                out[name]["workgroups"] = int(len(starts))
                out[name]["workgroups_started_more_than_10us_late"] = max(late[1:])
                out[name]["workgroups_started_more_than_10us_late_per_launch"] = late
        eng.close()
        del U, costs
    return out


def sampled_fused_at_scale(workloads, Engine, track, H, device, P=1024, N=16384, iters=8):
    """SURVEY.md 8f #3 at the scale a many-agent deployment would see: P independent problems x N candidates drawn ON THE
    DEVICE (Philox, smooth noise) and rolled out in the same launch - one fused round of `acmpc_optimize` - so the
    control-sample matrix never exists and the 8n bytes per candidate of the headline's byte model are not moved at
    all: reported separately, against the vector pipes' roof.  Kernel time from the event pair attached to the
    round's dispatch; the winners' records (argmin + re-roll inside the launch) are part of it."""
    n = H - 1
    base = workloads.problem_batch(track, 256, H, seed=0)
    batch = copy.copy(base)
    reps = P // 256
    batch.tables, batch.x0, batch.pose0 = (np.tile(base.tables, (reps, 1, 1)), np.tile(base.x0, (reps, 1)),
                                           np.tile(base.pose0, (reps, 1)))
    eng = Engine(**workloads.engine_kwargs(batch, 0, N, device=device.index))
    eng.set_paths(batch.tables)
    u_ref = np.ascontiguousarray(np.stack([batch.tables[:, 6, :], batch.tables[:, 3, :]], axis=2), dtype=np.float32)
    x0 = np.ascontiguousarray(batch.x0, dtype=np.float32)
    eng.optimize(x0, u_ref, u_ref, N, 1, SAMPLE_SIGMA, shrink=0.5, seed=1)        # warm-up (buffers, sampler tables)
    eng.profile_enable(iters)
    for i in range(iters):
        out = eng.optimize(x0, u_ref, u_ref, N, 1, SAMPLE_SIGMA, shrink=0.5, seed=2 + i)
    ms = float(np.median(eng.profile_collect()))
    eng.profile_enable(0)
    eng.close()
    block = {"workload": "%s (synthetic circuit): %d problems x %d candidates x horizon %d per launch, candidates sampled "
                         "in the rollout kernel (one fused round: sample + rollout + cost + argmin + winner record)"
                         % (track, P, N, H),
             "kernel_us": ms * 1e3, "candidate_trajectories_per_s": P * N / (ms * 1e-3),
             "winners_finite": bool(np.isfinite(out["cost"]).all()),
             "note": "no control matrix is read: the HBM byte model of `value` does not apply (a matrix of this batch "
                     "would be %.1f GB)" % (P * N * 8 * n / 1e9)}
    roofs = valu_roofline(counted_valu("sampled", "fused_round"), P * N, n, ms * 1e-3, mix_entry="fused_round", cpt=1)
    if roofs is not None:
        block["roofline_valu"] = roofs[0]
    return block


def sharded_solve_latency(workloads, Engine, ShardedRollout, world, rank, local_rank, device, backend, solves=200,
                          library_comm=None):
    """The second half of BASELINE.json's metric at N > 1 GPUs: BASELINE configs[3] as ONE sharded solve - 262 144
    candidates x horizon 80 in total, 262 144 / N per rank - one call at a time: this rank's rollout, the all-reduce(MIN)
    of the packed key over the ranks, the winner's record on every rank; barrier before, drained after, max over the
    ranks per solve.  Run by every rank (it contains collectives), reported by rank 0."""
    track, H, total = "nordschleife", 80, 262144
    n, N = H - 1, total // world
    batch = workloads.problem_batch(track, 1, H, seed=0)
    engine = Engine(**workloads.engine_kwargs(batch, 0, N, device=local_rank))
    engine.set_paths(batch.tables)
    stream = torch.cuda.current_stream().cuda_stream
    engine.sync_tables(stream)
    x0 = torch.tensor(batch.x0, device=device)
    u_ref = torch.tensor(np.stack([batch.tables[:, 6, :], batch.tables[:, 3, :]], axis=2), dtype=torch.float32,
                         device=device).contiguous()
    U = torch.empty((1, n, 2, N), dtype=torch.float32, device=device)
    engine.sample_device(u_ref.data_ptr(), 2 * n, u_ref.data_ptr(), 1, N, n, 1, rank * N, SAMPLE_SIGMA, 1000, 0,
                         U.data_ptr(), stream)
    slot = ShardedRollout(engine, 1, N, n, 1, index_offset=rank * N, device=device, host_collectives=backend == "gloo")
    slot.use_sampler(u_ref, u_ref, SAMPLE_SIGMA, 1000, 0)
    if library_comm is not None:
        slot.use_library_collective(library_comm.handle)
    warm = 10
    lat = np.empty(warm + solves)
    for i in range(lat.shape[0]):
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        slot.rollout(x0, U, stream)
        slot.select(x0, U, stream, seed=1000)
        torch.cuda.synchronize()
        lat[i] = time.perf_counter() - t0
    t = torch.tensor(lat[warm:], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    lat = t.cpu().numpy()
    engine.close()
    return {"workload": "%s (synthetic circuit), BASELINE configs[3] as one sharded solve: %d candidates x horizon %d in "
                        "total, %d per GPU over %d GPUs" % (track, total, H, N, world),
            "solve_us_p50": float(np.percentile(lat, 50) * 1e6), "solve_us_p99": float(np.percentile(lat, 99) * 1e6),
            "includes": "rollout of this rank's slice + all-reduce(MIN) of the packed key + winner record on every rank; "
                        "max over ranks per solve"}


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment: this process becomes the launcher - it
    starts `python -m torch.distributed.run --nproc-per-node N bench.py <the same arguments>` as a CHILD (never an exec
    of itself: see the pool's rules), lets rank 0's JSON line through on the inherited stdout and returns the child's exit
    code.  It only COUNTS devices (`torch.cuda.device_count()`: NVML / amdsmi where torch has them, else the runtime's
    device count - which may initialise the runtime in THIS process; harmless, it launches nothing on the GPU and never
    replaces itself).  With `--backend gloo` (the rehearsal on fewer cards than ranks) the ranks check for a GPU themselves."""
    import subprocess
    visible = torch.cuda.device_count()
    if args.backend == "nccl" and visible < args.gpus:
        print("bench: --gpus %d but only %d GPU(s) visible (one rank per device over RCCL); not running a smaller job "
              "under that label" % (args.gpus, visible), file=sys.stderr)
        return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this host driver
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench: no GPU visible (rank %d of %d) - the rollout path has no CPU fallback" % (rank, world))
    if args.backend == "gloo":
        local_rank = min(local_rank, torch.cuda.device_count() - 1)   # rehearsal: ranks may share a GPU
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    collective = None
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend="gloo")
        # the process group must really span --gpus ranks, one per device (backend "nccl" is RCCL on ROCm)
        if dist.get_world_size() != args.gpus:
            raise SystemExit("bench: process group has %d ranks, --gpus says %d" % (dist.get_world_size(), args.gpus))
        mine = torch.tensor([local_rank], dtype=torch.int64, device=device if args.backend == "nccl" else "cpu")
        seen = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(seen, mine)
        seen = [int(t.item()) for t in seen]
        if args.backend == "nccl" and len(set(seen)) != world:
            raise SystemExit("bench: ranks share devices: LOCAL_RANKs %s" % seen)
        collective = {"backend": "rccl (torch.distributed nccl)" if args.backend == "nccl" else "gloo (rehearsal)",
                      "world_size": dist.get_world_size(), "local_ranks": seen}

    from acmpc_amd import Engine, workloads
    from acmpc_amd.sharding import LibraryCommunicator, PipelinedRollout, ShardedRollout

    library_comm = None
    if args.collective == "capi":
        if args.pipeline:
            raise SystemExit("bench: --collective capi runs the collective on the launch stream (no --pipeline)")
        if args.backend != "nccl":
            raise SystemExit("bench: --collective capi is RCCL on device memory: one rank per GPU, --backend nccl")
        library_comm = LibraryCommunicator(local_rank)   # collective: every rank joins here
        collective = dict(collective or {"world_size": 1, "local_ranks": [local_rank]},
                          backend="rccl (the library's own communicator: acmpc_rccl_comm_create)",
                          entry="acmpc_reduce_across_ranks")
    elif collective is not None:
        collective["entry"] = "torch.distributed.all_reduce"

    mode = 0 if args.mode == "S" else 1
    P, N, H = args.poses, args.candidates, args.horizon
    n = H - 1
    # at most 256 distinct poses round the circuit (each needs a host-side speed-profile QP); beyond that the
    # poses repeat with independent candidate sets
    batch = workloads.problem_batch(args.track, min(P, 256), H, seed=0)
    if P > 256:
        reps = (P + 255) // 256
        batch.tables = np.tile(batch.tables, (reps, 1, 1))[:P]
        batch.x0 = np.tile(batch.x0, (reps, 1))[:P]
        batch.pose0 = np.tile(batch.pose0, (reps, 1))[:P]
    stream = torch.cuda.current_stream().cuda_stream
    engines = []
    for _ in range(2 if args.pipeline else 1):   # one handle per pipeline slot (own partial-key buffers)
        window = tuple(int(v) for v in args.nn_window.split(",")) if args.nn_window else None
        engine = Engine(**workloads.engine_kwargs(batch, mode, N, device=local_rank, nn_window=window))
        engine.set_paths(batch.tables)
        engine.sync_tables(stream)
        engines.append(engine)
    batch.coef_host = [engines[0].coefficients(p) for p in range(min(32, P))]
    x0 = torch.tensor(batch.x0 if mode == 0 else batch.pose0, device=device)
    # Synthetic control matrices, produced on the device by the library's counter-based sampler: candidate
    # (rank * N + c) of pose p in buffer b = clip(u_ref + amplitude * sigma * smooth noise), candidate 0 = u_ref.
    # Counter-based means any rank can re-draw any candidate from its global index, which is what lets the
    # multi-GPU step get by with ONE all-reduce(MIN) of the packed (cost, index) keys.
    u_ref = torch.tensor(np.stack([batch.tables[:, 6, :], batch.tables[:, 3, :]], axis=2), dtype=torch.float32,
                         device=device).contiguous()
    seeds = [1000 + b for b in range(args.buffers)]
    shape = (P, n, 2, N) if args.layout == 1 else (P, N, n, 2)
    controls = []
    for b in range(args.buffers):
        U = torch.empty(shape, dtype=torch.float32, device=device)
        engines[0].sample_device(u_ref.data_ptr(), 2 * n, u_ref.data_ptr(), P, N, n, args.layout, rank * N,
                                 SAMPLE_SIGMA, seeds[b], 0, U.data_ptr(), stream)
        controls.append(U)
    torch.cuda.synchronize()
    if not args.pipeline:
        class _Serial:  # rollout and argmin/record back to back on the launch stream
            def __init__(self):
                self.slot = ShardedRollout(engines[0], P, N, n, args.layout, index_offset=rank * N, device=device,
                                           host_collectives=args.backend == "gloo")
                if args.layout == 1:
                    self.slot.use_sampler(u_ref, u_ref, SAMPLE_SIGMA, seeds[0], 0)
                if library_comm is not None:
                    self.slot.use_library_collective(library_comm.handle)

                # one rank, nothing to exchange between rollout and argmin: the batches are a STREAM - batch i's argmin and
                # records run in the last rows of batch i + 1's rollout launch, the last batch's behind drain()
                # (acmpc_solve_stream_device) -, all of it inside the timed region; --no-stream: a launch of their own
                self.streamed = not (args.no_stream or self.slot.distributed or self.slot.offset != 0 or
                                     args.scaling == "strong")

            def step(self, x0_, U_, seed=None):
                # N ranks: rollout, the all-reduce(MIN) of the keys, the records
                if self.streamed:
                    self.slot.step_stream(x0_, U_, stream, seed=seed)
                else:
                    self.slot.step(x0_, U_, stream, seed=seed)
                return self.slot

            def drain(self):
                if self.streamed:
                    self.slot.flush(stream)

        pipe = _Serial()
    else:
        pipe = PipelinedRollout(engines, P, N, n, args.layout, index_offset=rank * N, device=device)
        pipe.bind_stream(torch.cuda.current_stream())

    streamed = bool(getattr(pipe, "streamed", False))
    # (the handle read these at create: with either, acmpc_solve_stream_device runs the pending finalize as its own launch)
    separate_finalize = any(os.environ.get(name, "0") not in ("", "0") for name in ("ACMPC_CONFORMANT_SYNC", "ACMPC_NO_CHAINED_STREAM"))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run_step(i):
        b = i % args.buffers
        if args.pipeline:
            return pipe.step(x0, controls[b])
        return pipe.step(x0, controls[b], seed=seeds[b])

    for i in range(args.warmup):
        run_step(i)
    pipe.drain()
    barrier()
    # HIP event pairs attached to every rollout dispatch of the timed region (hipExtLaunchKernel through
    # acmpc_profile_enable): the kernel's own begin/end on its launch stream, no marker packets between launches
    per_engine = (args.steps + len(engines) - 1) // len(engines)
    for engine in engines:
        engine.profile_enable(0 if args.no_kernel_timing else per_engine)
    t0 = time.perf_counter()
    last = None
    for i in range(args.steps):
        last = run_step(i)
    host_enqueue = time.perf_counter() - t0   # (diagnostic) time the host needed to enqueue all steps
    pipe.drain()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_times = np.concatenate([engine.profile_collect() for engine in engines])
    step_latency = None
    if args.scaling == "strong":
        # the number a controller feels: one step at a time (rollout of this rank's slice, the all-reduce(MIN) of the
        # keys, every rank re-drawing the winner), drained before the next one starts; max over ranks per step
        lat = np.empty(min(args.steps, 200))
        for i in range(lat.shape[0]):
            barrier()
            t1 = time.perf_counter()
            run_step(i)
            torch.cuda.synchronize()
            lat[i] = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor(lat, dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            lat = t.cpu().numpy()
        step_latency = {"p50_us": float(np.percentile(lat, 50) * 1e6), "p99_us": float(np.percentile(lat, 99) * 1e6),
                        "includes": "rollout of %d candidates per rank + all-reduce(MIN) of %d key(s) over %d rank(s) + "
                                    "winner record on every rank" % (N, P, world)}
    if world > 1:
        # self-check outside the timed region: after the single all-reduce(MIN) every rank must hold the same plan
        # (feasible counts are per rank and excluded)
        rec = last.records.clone()
        rec[:, 2] = 0
        digest = torch.stack([rec.double().sum(), rec.double().abs().max()])
        lo, hi = digest.clone(), digest.clone()
        if args.backend == "gloo":
            lo, hi = lo.cpu(), hi.cpu()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        if not torch.equal(lo, hi):
            raise SystemExit("bench: ranks disagree on the winners' records after the all-reduce")
    if args.no_kernel_timing:
        kernel_ms = float("nan")
    else:
        assert len(kernel_times) == args.steps, "every rollout launch of the timed region carries an event pair"
        kernel_ms = float(kernel_times.mean())

    sharded_solve = None
    if world > 1 and args.scaling == "weak" and not args.no_single_solve:
        # (every rank: it contains collectives; after the timed region, so it cannot disturb the headline)
        try:
            sharded_solve = sharded_solve_latency(workloads, Engine, ShardedRollout, world, rank, local_rank, device,
                                                  args.backend, library_comm=library_comm)
        except Exception as error:   # (the same on every rank, or the collectives above would not have returned)
            sharded_solve = {"error": "%s: %s" % (type(error).__name__, error)}
    if rank == 0:
        algorithmic_bytes = P * N * (8 * n + 4)  # 8n B of controls read + 4 B of cost written
        achieved = algorithmic_bytes / (kernel_ms * 1e-3) / 1e9
        total_candidates = P * N * world
        if args.scaling == "strong":
            workload = ("%s (synthetic circuit), BASELINE configs[3]: %d candidates x horizon %d in total, %d per GPU over %d "
                        "GPU(s), %d problem(s) per step; mode S rollout + cost + one all-reduce(MIN) of the packed keys + "
                        "winner record on every rank" % (args.track, N * world, H, N, world, P))
        else:
            workload = ("%s (synthetic circuit): %d candidates x horizon %d per solve per GPU, %d solves (poses) "
                        "batched per launch (%d distinct poses); mode %s rollout + cost + argmin + winner record"
                        % (args.track, N, H, P, min(P, 256), args.mode))
        out = {
            "metric": "candidate-trajectories/sec (horizon=%d)" % H,
            "value": total_candidates * args.steps / elapsed,
            "unit": "candidate-trajectories/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": workload,
                "candidates_per_solve_per_gpu": N, "solves_per_step": P, "horizon": H, "mode": args.mode,
                "layout": "U[P][n][2][N]" if args.layout == 1 else "U[P][N][n][2]",
                "parallelism": "candidate-sharded x%d, one all-reduce(MIN) of %d packed keys per step" % (world, P),
                "pipeline": ("argmin/record of batch i overlaps rollout of batch i+1" if args.pipeline else
                             "argmin/record of batch i inside the rollout launch of batch i+1, the last batch's behind it "
                             "(acmpc_solve_stream_device)" if streamed else "none (stream order)"),
            },
            "roofline": {
                "bound": "hbm", "kernel": "rollout_chained_kernel" if (streamed and not separate_finalize) else "rollout_kernel",
                "achieved": achieved, "peak": HBM_PEAK_GBPS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                "algorithmic_bytes_per_launch": algorithmic_bytes, "kernel_ms": kernel_ms,
            },
            "host_enqueue_ms_per_step": host_enqueue / args.steps * 1e3,
        }
        traffic = latest_traffic(algorithmic_bytes, "rollout_chained_kernel<%d" % args.layout if (streamed and not separate_finalize) else
                                 "rollout_kernel<%d, %d" % (mode, args.layout))
        if traffic is not None:
            out["roofline"]["traffic"], out["roofline"]["traffic_source"] = traffic
        # (PMC counters need rocprofv3 round the process: the bytes come from the committed profile of this same command)
        out["roofline"]["traffic_measured_in_this_run"] = False
        if step_latency is not None:
            out["step_latency"] = step_latency
        if collective is not None:
            out["collective"] = collective
        if sharded_solve is not None:
            # "+ MPC solve p50 latency" of the metric at this GPU count: one problem's candidates split over the ranks
            out["sharded_solve"] = sharded_solve
        if sharded_solve is not None and "error" not in sharded_solve:
            out["mpc_solve_latency_ms"] = {"p50": sharded_solve["solve_us_p50"] * 1e-3,
                                           "p99": sharded_solve["solve_us_p99"] * 1e-3,
                                           "of": "one sharded solve (sharded_solve workload)"}
        if world == 1 and not args.no_single_solve and args.scaling == "weak":
            out["single_solve"] = single_solve(workloads, Engine, args.track, H, N, mode, args.layout, device)
            out["closed_loop_replay"] = closed_loop_replay(workloads)
            # the same loop with north_star's literal rollout (Cartesian bicycle + nearest waypoint) behind get_control
            out["closed_loop_replay_mode_T"] = closed_loop_replay(workloads, poses=1000, rollout_mode="T")
            out["closed_loop_replay_mode_T_window_2_5"] = closed_loop_replay(workloads, poses=1000, rollout_mode="T",
                                                                             nn_window=(2, 5))
            # the second half of BASELINE.json's metric ("+ MPC solve p50 latency"): the drop-in get_control
            out["mpc_solve_latency_ms"] = {"p50": out["closed_loop_replay"]["solve_ms_p50"],
                                           "p99": out["closed_loop_replay"]["solve_ms_p99"],
                                           "of": "one SpatialMPC solve per pose (get_control_at), closed_loop_replay workload"}
            out["secondary_kernels"] = secondary_kernels(workloads, Engine, args.track, H, N, device)
            out["secondary_kernels"]["sampled_fused_16M"] = sampled_fused_at_scale(workloads, Engine, args.track, H, device)
            # the other BASELINE configurations as ONE problem per call (what DESIGN.md quotes for them)
            out["config3_single"] = single_solve(workloads, Engine, "spa", 50, 65536, 0, 1, device, host_pointer=False)
            out["config4_share"] = single_solve(workloads, Engine, "nordschleife", 80, 32768, 0, 1, device,
                                                host_pointer=False)
            out["config5_host_pointer"] = config5_host_pointer(workloads, Engine, device)
            out["config5_host_pointer"]["pinned"] = config5_host_pointer(workloads, Engine, device, poses=3000, pinned=True)
            out["particle_filter"] = particle_filter_block(workloads)
        if "secondary_kernels" in out:
            # The driver's record keeps `roofline`'s scalar keys verbatim and only the NAMES of the nested blocks: the figures
            # the review reads - north_star's literal kernel, the one-solve configurations, the tick, the particle filter -
            # are copied here as scalars (the blocks they come from stay in the line).
            out["roofline"].update(secondary_scalars(out))
        if world == 1 and not args.no_cpu_baseline:
            gpu_costs = last.costs[:32].cpu().numpy()  # costs of the last step = controls[(steps-1) % buffers]
            out["cpu_baseline"] = cpu_baseline(batch, mode, controls[(args.steps - 1) % args.buffers], args.layout, N,
                                               n, gpu_costs, args.cpu_seconds)
            if mode == 0:
                out["cpu_baseline"]["numpy"] = numpy_baseline(batch, controls[(args.steps - 1) % args.buffers],
                                                              args.layout, N, n, gpu_costs)
                out["cpu_baseline"]["numpy_all_cores"] = numpy_all_cores_baseline(
                    batch, controls[(args.steps - 1) % args.buffers], args.layout, N, n)
        print(json.dumps(out), flush=True)
    if library_comm is not None:
        library_comm.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
