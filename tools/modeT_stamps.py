"""Development probe (GPU box): the timeline of ONE launch of the mode T rollout kernel, wave by wave.

A scratch build of the library with -DACMPC_T_STAMPS (tools/ab_build.sh stamps "-DACMPC_T_STAMPS") makes lane 0 of every
wave stamp the 100 MHz wall clock at entry, after its workgroup's tables are staged, after the step loop and at its end,
and leave its place on the chip (HW_ID, XCC_ID).  This script rolls the bench's secondary workload (P poses x 4 096
candidates x H = 50) once per search and prints where the launch's time goes: the dispatch ramp (first to last wave
started), staging, the step loop (fastest / median / slowest wave), the drain (first to last wave finished), and how
the waves were dealt over XCDs, CUs and SIMDs.

usage:  LIB=$(tools/ab_build.sh stamps "-DACMPC_T_STAMPS") && ACMPC_HIP_LIBRARY=$LIB python3 tools/modeT_stamps.py [P] [out.json]"""
import copy
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ac-mpc_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from acmpc_amd import Engine, _capi, workloads  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 256
out_path = sys.argv[2] if len(sys.argv) > 2 else None
device = torch.device("cuda", 0)
torch.cuda.set_device(0)
H, N, n = 50, 4096, 49
base = workloads.problem_batch("monza", 256, H, seed=0)
stream = torch.cuda.current_stream().cuda_stream
lib = _capi.load_library()
reader = lib.acmpc_debug_t_stamps          # AttributeError: not a -DACMPC_T_STAMPS build
reader.restype = ctypes.c_int
reader.argtypes = [ctypes.c_void_p, ctypes.c_int]
report = {"tool": "tools/modeT_stamps.py %d" % P, "unit": "microseconds (100 MHz wall clock, 10 ns resolution)", "searches": {}}
REPEATS = int(os.environ.get("STAMP_REPEATS", "1"))
for name, window in [(("exhaustive", None), ("window_2_5", (2, 5)))[k % 2] for k in range(2 * REPEATS)]:
    batch = copy.copy(base)
    reps = P // 256
    batch.tables, batch.x0, batch.pose0 = (np.tile(base.tables, (reps, 1, 1)), np.tile(base.x0, (reps, 1)),
                                           np.tile(base.pose0, (reps, 1)))
    u_ref = torch.tensor(np.stack([batch.tables[:, 6, :], batch.tables[:, 3, :]], axis=2), dtype=torch.float32,
                         device=device).contiguous()
    costs = torch.empty(P, N, device=device)
    eng = Engine(**workloads.engine_kwargs(batch, 1, N, device=0, nn_window=window))
    eng.set_paths(batch.tables)
    eng.sync_tables(stream)
    x0 = torch.tensor(batch.pose0, device=device)
    U = torch.empty((P, n, 2, N), device=device)
    eng.sample_device(u_ref.data_ptr(), 2 * n, u_ref.data_ptr(), P, N, n, 1, 0, bench.SAMPLE_SIGMA, 77, 0, U.data_ptr(), stream)
    eng.profile_enable(8)
    for _ in range(8):
        eng.rollout_device(x0.data_ptr(), U.data_ptr(), P, N, n, 1, 0, costs.data_ptr(), 0, stream)
    torch.cuda.synchronize()
    kernel_us = float(eng.profile_collect()[-1]) * 1e3
    waves = P * N // 128      # two candidates per lane
    raw = np.zeros(waves * 6, dtype=np.uint64)
    rc = reader(raw.ctypes.data, waves)
    assert rc == 0, rc
    s = raw.reshape(waves, 6)
    t = s[:, :4].astype(np.int64)
    t0 = t[:, 0].min()
    us = (t - t0) / 100.0
    hw, xcc = s[:, 4].astype(np.int64), s[:, 5].astype(np.int64) & 0xF
    simd, cu, sh, se = (hw >> 4) & 0x3, (hw >> 8) & 0xF, (hw >> 12) & 0x1, (hw >> 13) & 0x7
    place_cu = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    place_simd = place_cu * 4 + simd
    per_cu = np.bincount(np.unique(place_cu, return_inverse=True)[1])
    per_simd = np.bincount(np.unique(place_simd, return_inverse=True)[1])
    per_xcc = np.bincount(xcc, minlength=8)
    loop = us[:, 2] - us[:, 1]
    q = lambda a: {"min": float(a.min()), "p50": float(np.median(a)), "p99": float(np.percentile(a, 99)), "max": float(a.max())}
    # how many waves are still running at time x (the drain): fraction of the launch with fewer than half the waves resident
    ends = np.sort(us[:, 3])
    span = float(us[:, 3].max())
    entry = {"kernel_us_by_hip_events": kernel_us, "waves": int(waves), "span_first_start_to_last_end": span,
             "start": q(us[:, 0]), "staged_minus_start": q(us[:, 1] - us[:, 0]), "step_loop": q(loop),
             "end_minus_loop": q(us[:, 3] - us[:, 2]), "end": q(us[:, 3]),
             "drain_us_from_first_wave_finished_to_last": float(ends[-1] - ends[0]),
             "time_with_fewer_than_half_the_waves_left": float(ends[-1] - ends[len(ends) // 2]),
             "mean_resident_fraction_over_the_span": float((us[:, 3] - us[:, 0]).sum() / (span * waves)),
             "waves_per_xcd": per_xcc.tolist(), "compute_units_used": int(len(per_cu)),
             "waves_per_cu": {"min": int(per_cu.min()), "max": int(per_cu.max())},
             "simds_used": int(len(per_simd)), "waves_per_simd": {"min": int(per_simd.min()), "max": int(per_simd.max()),
                                                                   "histogram": np.bincount(per_simd).tolist()}}
    # which compute units ran other than 8 workgroups (place: xcc, se, sh, cu -> waves), and when their late waves started
    places, counts = np.unique(place_cu, return_counts=True)
    odd = {}
    for place, count in zip(places, counts):
        if count != 32:
            waves_here = us[place_cu == place]
            odd["xcc%d se%d sh%d cu%d" % (place // 256, (place // 32) % 8, (place // 16) % 2, place % 16)] = {
                "waves": int(count), "latest_start": float(waves_here[:, 0].max()), "earliest_end": float(waves_here[:, 3].min())}
    entry["compute_units_with_other_than_32_waves"] = odd
    # does a SIMD with more waves finish later?  mean end time by the SIMD's wave count
    inverse = np.unique(place_simd, return_inverse=True)[1]
    count_of_wave = per_simd[inverse]
    entry["mean_end_by_waves_on_the_simd"] = {int(c): float(us[count_of_wave == c, 3].mean()) for c in np.unique(count_of_wave)}
    entry["mean_loop_by_waves_on_the_simd"] = {int(c): float(loop[count_of_wave == c].mean()) for c in np.unique(count_of_wave)}
    report["searches"].setdefault(name, []).append(entry)
    print(name, json.dumps(entry), flush=True)
    eng.close()
if out_path:
    with open(out_path, "w") as handle:
        json.dump(report, handle, indent=1)
