"""Development tool: tools/profile_pf.sh's raw rocprofv3 --stats output -> profiles/<tag>_pf_kernel_stats.json (the average
duration of every particle-filter kernel per particle count) and profiles/<tag>_pf_kernel_stats_<P>.csv (the summaries
themselves).   usage: python3 tools/summarize_pf_profile.py r04"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
out = {"tool": "tools/profile_pf.sh " + tag + " (rocprofv3 --kernel-trace --stats -- python3 tools/bench_pf.py <P>), one particle "
               "count per run", "sizes": {}}
for P in (500, 100000):
    found = glob.glob(os.path.join(ROOT, "gpurun_out", "pf_%s_%d" % (tag, P), "**", "*kernel_stats.csv"), recursive=True)
    if not found:
        continue
    shutil.copy(found[0], os.path.join(ROOT, "profiles", "%s_pf_kernel_stats_%d.csv" % (tag, P)))
    kernels = {}
    with open(found[0], newline="") as handle:
        for row in csv.DictReader(handle):
            name = row["Name"]
            for short in ("pf_score_kernel<1>", "pf_score_kernel<8>", "pf_score_given_kernel", "pf_nearest_kernel", "pf_advance_kernel", "pf_estimate_kernel"):
                if short in name:
                    kernels[short] = {"calls": int(row["Calls"]), "average_us": float(row["AverageNs"]) / 1e3,
                                      "min_us": float(row["MinNs"]) / 1e3}
    out["sizes"][str(P)] = kernels
path = os.path.join(ROOT, "profiles", tag + "_pf_kernel_stats.json")
with open(path, "w") as handle:
    json.dump(out, handle, indent=1)
print(json.dumps(out, indent=1))
