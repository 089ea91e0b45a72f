#!/bin/bash
# Here, after tools/final_profiles.sh <tag> ran on the GPU box: drop what older runs left in the merged directories, then
# every summariser -> profiles/<tag>_*.   usage: tools/summarize_all.sh <tag> [minutes: files older than this are stale, 10]
set -eu
TAG=${1:-r04}; AGE=${2:-10}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT/gpurun_out"
for d in prof_$TAG allk_$TAG trace_f_* pmc_${TAG}_T_*_a pmc_${TAG}_T_*_b pmc_${TAG}_sampled_a pmc_${TAG}_sampled_b pf_${TAG}_500 pf_${TAG}_100000 pmc_${TAG}new_pf_* pmc_${TAG}old_pf_* pf_${TAG}new_* pf_${TAG}old_*; do
  [ -d "$d" ] && find "$d" -type f -mmin +$AGE -delete
done
cd "$ROOT"
python3 tools/summarize_profile.py $TAG > /dev/null
python3 tools/summarize_sq_counters.py $TAG > /dev/null
python3 tools/summarize_sq_counters.py $TAG sampled > /dev/null
python3 tools/summarize_pf_profile.py $TAG > /dev/null
python3 tools/summarize_tick_trace.py $TAG mode_S_h50=f_S50 mode_S_h25=f_S25 mode_S_h100=f_S100 mode_S_h50_pair=f_S50_pair \
  mode_T_h50=f_T50 mode_T_h50_one=f_T50_one mode_T_h50_window_2_5=f_T50_w25 > /dev/null
python3 tools/summarize_pf_counters.py $TAG > /dev/null
python3 tools/summarize_round5.py $TAG
cp "$(ls -t gpurun_out/allk_$TAG/runc/*_kernel_stats.csv | head -1)" profiles/${TAG}_all_kernels_stats.csv
python3 - "$TAG" <<'PY'
import json, sys
sys.path[:0] = [".", "ac-mpc_amd"]
import bench
tag, h = sys.argv[1], bench.loaded_source_hash()
for name in ("mode_T_sq_counters", "sampled_sq_counters", "isa_mix"):
    print("profiles/%s_%s.json describes the sources in the tree: %s" % (tag, name, json.load(open("profiles/%s_%s.json" % (tag, name))).get("source_sha256") == h))
b = json.load(open("profiles/%s_bench.json" % tag)); r = b["roofline"]
print("bench: value %.4g  ms_per_step %.4f  kernel_ms %.4f  ratio %.4f  frac %.3f  traffic %s" % (b["value"], b["ms_per_step"], r["kernel_ms"], b["ms_per_step"] / r["kernel_ms"], r["frac"], r["traffic"]))
s = json.load(open("profiles/%s_summary.json" % tag))
print("trace: %s x%d  %.1f us (events in the same process %.1f us), traffic %.4f x algorithmic" % (s["dominant_kernel"][40:80], s["calls"], s["average_ns"] / 1e3, s["hip_events_kernel_ms_under_rocprof"] * 1e3, s["traffic_over_algorithmic"]))
PY
