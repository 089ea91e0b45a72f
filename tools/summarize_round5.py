"""Development tool (here, after tools/final_profiles.sh <tag>): round 5's extra evidence -> profiles/.
  gpurun_out/modeT_stamps_<tag>.json   -> profiles/<tag>_mode_T_timeline.json, variant "shipped" (beside the A/B variants
                                          taken while the launcher was changed)
  gpurun_out/time_modeT_<tag>.log      -> profiles/<tag>_mode_T_launches.json (median / min / max of 20 launches per search,
                                          late workgroups, three processes at 1 M and one at 16.8 M)
  gpurun_out/conformant_cost_<tag>.json -> profiles/<tag>_conformant_cost.json
usage: python3 tools/summarize_round5.py r05"""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
out = os.path.join(ROOT, "gpurun_out")
timeline_path = os.path.join(ROOT, "profiles", tag + "_mode_T_timeline.json")
stamps = os.path.join(out, "modeT_stamps_%s.json" % tag)
if os.path.exists(stamps) and os.path.exists(timeline_path):
    timeline = json.load(open(timeline_path))
    shipped = {}
    for name, runs in json.load(open(stamps))["searches"].items():
        shipped[name] = [{"kernel_us_by_hip_events": e["kernel_us_by_hip_events"], "start": e["start"], "staged_minus_start": e["staged_minus_start"],
                          "step_loop": e["step_loop"], "end": e["end"],
                          "mean_resident_fraction_over_the_span": e["mean_resident_fraction_over_the_span"],
                          "time_with_fewer_than_half_the_waves_left": e["time_with_fewer_than_half_the_waves_left"],
                          "waves_per_simd_histogram": e["waves_per_simd"]["histogram"],
                          "compute_units_with_other_than_32_waves": len(e["compute_units_with_other_than_32_waves"])} for e in runs]
    timeline["variants"]["shipped"] = shipped
    timeline["shipped_is"] = "the tree's library built with -DACMPC_T_STAMPS on the box of tools/final_profiles.sh %s (another box than the A/B variants')" % tag
    json.dump(timeline, open(timeline_path, "w"), indent=1)
    print("timeline: shipped variant added")
log = os.path.join(out, "time_modeT_%s.log" % tag)
if os.path.exists(log):
    rows = []
    pattern = re.compile(r"(\S+)\s+([\d.]+) us\s+([\d.]+) of the HBM roofline\s+\(min ([\d.]+) max ([\d.]+) us; workgroups started > 10 us late per "
                         r"launch: median (\d+), max (\d+) of (\d+);")
    for line in open(log):
        m = pattern.search(line)
        if m:
            rows.append({"search": m.group(1), "median_us": float(m.group(2)), "frac_of_hbm_roofline": float(m.group(3)),
                         "min_us": float(m.group(4)), "max_us": float(m.group(5)), "late_workgroups_median": int(m.group(6)),
                         "late_workgroups_max": int(m.group(7)), "workgroups": int(m.group(8))})
    json.dump({"tool": "tools/time_modeT.py 256 (three processes) and 4096 (one), tools/final_profiles.sh " + tag,
               "note": "workgroups = 2 048: 1 M candidates (one generation of waves); 32 768: 16.8 M (sixteen - `late` means nothing there)",
               "launch_sets": rows}, open(os.path.join(ROOT, "profiles", tag + "_mode_T_launches.json"), "w"), indent=1)
    print("launches:", len(rows), "rows")
cost = os.path.join(out, "conformant_cost_%s.json" % tag)
if os.path.exists(cost):
    shutil.copy(cost, os.path.join(ROOT, "profiles", tag + "_conformant_cost.json"))
    print("conformant cost copied")
