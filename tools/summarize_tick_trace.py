#!/usr/bin/env python3
"""Development tool: per-kernel timeline of the closed-loop tick out of the raw rocprofv3 --kernel-trace output that
tools/tick_trace.sh leaves under gpurun_out/trace_<name>/, as profiles/<tag>_tick_timeline.json.

usage: tools/summarize_tick_trace.py <tag> <label>=<name> [<label>=<name> ...]"""
import csv
import glob
import json
import os
import re
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def timeline(name):
    rows = []
    for path in glob.glob(os.path.join(ROOT, "gpurun_out", "trace_" + name, "**", "*kernel_trace.csv"), recursive=True):
        with open(path, newline="") as handle:
            for row in csv.DictReader(handle):
                rows.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), row["Kernel_Name"]))
    rows.sort()
    out = {"prologue_kernel": [], "first_round": [], "last_round": []}
    kernels = set()
    i = 0
    while i + 2 < len(rows):
        if "prologue_kernel" in rows[i][2] and "rollout_sampled" in rows[i + 1][2] and "rollout_sampled" in rows[i + 2][2]:
            for key, row in zip(out, rows[i:i + 3]):
                out[key].append(row[1] - row[0])
            kernels.add(re.search(r"rollout_sampled\w*(<\d+>)?", rows[i + 1][2]).group(0))
            i += 3
        else:
            i += 1
    summary = {key: {"median_us": statistics.median(v) / 1e3, "p10_us": sorted(v)[len(v) // 10] / 1e3,
                     "p90_us": sorted(v)[9 * len(v) // 10] / 1e3} for key, v in out.items() if v}
    summary["ticks"] = len(out["prologue_kernel"])
    summary["round_kernel"] = sorted(kernels)
    return summary


def main():
    tag = sys.argv[1]
    result = {"command": "tools/tick_trace.sh <name>: rocprofv3 --kernel-trace of tools/tick_probe.py (1 520 consecutive "
                         "get_control calls, 16 384 candidates x 2 rounds; TICK_H = horizon), medians per kernel of a tick",
              "runs": {}}
    for item in sys.argv[2:]:
        label, name = item.split("=", 1)
        result["runs"][label] = timeline(name)
    path = os.path.join(ROOT, "profiles", tag + "_tick_timeline.json")
    with open(path, "w") as handle:
        json.dump(result, handle, indent=1)
    print(json.dumps(result, indent=1))


if __name__ == "__main__":
    main()
