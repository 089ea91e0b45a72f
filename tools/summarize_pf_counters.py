"""Development tool: tools/pmc_pf.sh's raw rocprofv3 passes -> profiles/<tag>_pf_sq_counters.json: per particle-filter
kernel and particle count the mean SQ / L2 / vector-cache counters per launch, for the tree's kernels (`<tag>new`) and -
when the before / after pair was taken on one box - round 4's (`<tag>old`, tools/ab_old_pf.sh), with the kernel-trace
durations of the same runs beside them.   usage: python3 tools/summarize_pf_counters.py r05"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
SHORT = {"pf_nearest_kernel": "pf_nearest_kernel", "pf_score_kernel<1>": "pf_score_kernel<1>", "pf_score_kernel<8>": "pf_score_kernel<8>",
         "pf_score_given_kernel": "pf_score_given_kernel"}


def short(name):
    for key, value in SHORT.items():
        if key in name:
            return value
    return None


def one(label, P):
    sums = collections.defaultdict(lambda: collections.defaultdict(list))
    for suffix in ("a", "b"):
        for path in glob.glob(os.path.join(ROOT, "gpurun_out", "pmc_%s_pf_%d_%s" % (label, P, suffix), "**", "*counter_collection.csv"), recursive=True):
            with open(path, newline="") as handle:
                for row in csv.DictReader(handle):
                    kernel = short(row["Kernel_Name"])
                    if kernel:
                        sums[kernel][row["Counter_Name"]].append(float(row["Counter_Value"]))
    durations = {}
    for path in glob.glob(os.path.join(ROOT, "gpurun_out", "pf_%s_%d" % (label, P), "**", "*kernel_stats.csv"), recursive=True):
        with open(path, newline="") as handle:
            for row in csv.DictReader(handle):
                kernel = short(row["Name"])
                if kernel:
                    durations[kernel] = {"average_us": float(row["AverageNs"]) / 1e3, "min_us": float(row["MinNs"]) / 1e3, "calls": int(row["Calls"])}
    out = {}
    for kernel, counters in sums.items():
        m = {name: sum(v) / len(v) for name, v in sorted(counters.items())}
        d = {"counters_mean_per_launch": m, "kernel_trace": durations.get(kernel)}
        wc = m.get("SQ_WAVE_CYCLES")
        if wc:
            d["derived"] = {
                "share_of_wave_cycles": {"waiting_on_memory_or_barrier (SQ_WAIT_ANY)": m["SQ_WAIT_ANY"] / wc,
                                         "issue_stalled (SQ_WAIT_INST_ANY: the vector-memory queue is full)": m["SQ_WAIT_INST_ANY"] / wc,
                                         "issuing (SQ_ACTIVE_INST_ANY)": m["SQ_ACTIVE_INST_ANY"] / wc},
                "vector_memory_reads_per_wave": m["SQ_INSTS_VMEM_RD"] / m["SQ_WAVES"],
                "valu_instructions_per_wave": m["SQ_INSTS_VALU"] / m["SQ_WAVES"],
                "waves": m["SQ_WAVES"],
            }
            if "TCC_HIT_sum" in m:
                lines = m.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0.0)
                d["derived"].update({
                    "l2_hit_rate": m["TCC_HIT_sum"] / max(m["TCC_HIT_sum"] + m["TCC_MISS_sum"], 1.0),
                    "l2_requests": m["TCC_REQ_sum"],
                    "vector_cache_line_accesses": lines,
                    "vector_cache_line_accesses_per_read_instruction": lines / max(m["SQ_INSTS_VMEM_RD"], 1.0),
                    "l2_read_requests_from_the_vector_caches": m.get("TCP_TCC_READ_REQ_sum"),
                })
        out[kernel] = d
    return out


result = {"tool": "tools/pmc_pf.sh (separate --pmc passes of python3 tools/bench_pf.py <P>; SQ pass, then TCC / TCP / GRBM pass, then "
                  "--kernel-trace --stats), summarised by tools/summarize_pf_counters.py " + tag,
          "workload": "tools/bench_pf.py: P particles scattered 2 m (1 sigma) round the synthetic Monza centre line, 11 586 points "
                      "per polyline, 200 observation points; every kernel's counters are means over the run's 11 scoring calls",
          "runs": {}}
for label, what in ((tag + "old", "round 4's kernels (tools/_ab_old/acmpc_pf.hip built by tools/ab_old_pf.sh), same box, same call"),
                    (tag + "new", "this tree's kernels")):
    sizes = {str(P): one(label, P) for P in (500, 100000)}
    if any(sizes.values()):
        result["runs"][label] = {"what": what, "sizes": sizes}
path = os.path.join(ROOT, "profiles", tag + "_pf_sq_counters.json")
with open(path, "w") as handle:
    json.dump(result, handle, indent=1)
print(path)
for label, run in result["runs"].items():
    for P, kernels in run["sizes"].items():
        for kernel, d in kernels.items():
            print(label, P, kernel, d["kernel_trace"], {k: round(v, 3) if isinstance(v, float) else v for k, v in d.get("derived", {}).items() if not isinstance(v, dict)})
