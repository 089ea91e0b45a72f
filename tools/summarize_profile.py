#!/usr/bin/env python3
"""Turns gpurun_out/prof_<tag>/ (written by tools/profile.sh) into the committed evidence under profiles/:
   profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (verbatim, our kernels first)
   profiles/<tag>_summary.json       dominant-kernel duration, PMC counters, HBM traffic with the gfx950 correction
   profiles/<tag>_bench.json         the bench line of the un-profiled run
usage: tools/summarize_profile.py <tag> [kernel-substring]
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    needle = sys.argv[2] if len(sys.argv) > 2 else "rollout_"   # (the dominant one: rows are sorted by total duration)
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    # gpurun MERGES a call's files into gpurun_out/ and never deletes: after several profile runs the directory holds
    # one set of CSVs per run (the process id is in the file name).  Only the newest of each kind is this run's.
    def newest(pattern):
        found = glob.glob(os.path.join(src, pattern), recursive=True)
        if not found:
            raise SystemExit("no %s under %s - run tools/profile.sh %s on the GPU box first" % (pattern, src, tag))
        return max(found, key=os.path.getmtime)

    stats = newest(os.path.join("trace", "**", "*_kernel_stats.csv"))
    rows = list(csv.DictReader(open(stats)))
    rows.sort(key=lambda r: (0 if "acmpc" in r["Name"] else 1, -float(r["TotalDurationNs"])))
    with open(os.path.join(dst, tag + "_kernel_stats.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows)
    dom = [r for r in rows if needle in r["Name"]][0]
    counters = collections.defaultdict(list)
    for pass_dir in sorted(d for d in glob.glob(os.path.join(src, "pmc_*")) if os.path.isdir(d)):
        path = newest(os.path.join(os.path.basename(pass_dir), "**", "*_counter_collection.csv"))
        for r in csv.DictReader(open(path)):
            if r["Kernel_Name"] == dom["Name"]:
                counters[r["Counter_Name"]].append(float(r["Counter_Value"]))
    mean = {k: sum(v) / len(v) for k, v in counters.items()}
    bench = json.loads(open(os.path.join(src, "bench_plain.json")).read().strip().splitlines()[-1])
    summary = {
        "tag": tag,
        "command": "python bench.py --steps 200 --warmup 20: plain run, and the same under rocprofv3 --kernel-trace --stats; --pmc passes with --steps 8",
        "dominant_kernel": dom["Name"],
        "calls": int(dom["Calls"]),
        "average_ns": float(dom["AverageNs"]),
        "min_ns": float(dom["MinNs"]),
        "max_ns": float(dom["MaxNs"]),
        "pmc_mean_per_launch": mean,
        "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"],
    }
    if "FETCH_SIZE" in mean and "WRITE_SIZE" in mean:
        # MI355X_MICROARCH.md "HBM": FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly half of
        # the bytes of a wide coalesced streaming read -> double it; WRITE_SIZE is exact for streaming stores.
        fetch = mean["FETCH_SIZE"] * 1024 * 2
        write = mean["WRITE_SIZE"] * 1024
        summary["hbm_traffic_bytes_per_launch"] = fetch + write
        summary["hbm_fetch_bytes_corrected"] = fetch
        summary["hbm_write_bytes"] = write
        summary["traffic_over_algorithmic"] = (fetch + write) / summary["algorithmic_bytes_per_launch"]
    summary["achieved_GBps_rocprof"] = summary["algorithmic_bytes_per_launch"] / summary["average_ns"]
    summary["frac_of_8TBps_rocprof"] = summary["achieved_GBps_rocprof"] / 8000.0
    summary["bench_kernel_ms_hip_events"] = bench["roofline"]["kernel_ms"]
    traced = os.path.join(src, "bench_trace.json")
    if os.path.exists(traced):
        t = json.loads(open(traced).read().strip().splitlines()[-1])
        # the same process, the same launches: HIP events vs rocprofv3's own dispatch timestamps
        summary["hip_events_kernel_ms_under_rocprof"] = t["roofline"]["kernel_ms"]
        summary["note"] = ("compare average_ns with hip_events_kernel_ms_under_rocprof (the same process, the same "
                           "launches); bench_kernel_ms_hip_events is a separate, un-profiled run on the same box - "
                           "the two runs differ by a few per cent either way (clocks, MI355X_MICROARCH.md 'DVFS "
                           "give-back')")
    json.dump(summary, open(os.path.join(dst, tag + "_summary.json"), "w"), indent=1)
    json.dump(bench, open(os.path.join(dst, tag + "_bench.json"), "w"), indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
