#!/bin/bash
# The round's committed evidence in one call on the GPU box (about nine minutes): tools/profile.sh (headline kernel: trace,
# counters, un-profiled bench line), the all-kernels trace of a full bench run, the tick timelines, the mode T counters, the
# fused sampled round's counters, the particle filter's kernel traces.
# usage: tools/final_profiles.sh <tag>      then, here:  python3 tools/summarize_profile.py <tag>; python3
# tools/summarize_sq_counters.py <tag>; python3 tools/summarize_sq_counters.py <tag> sampled; python3 tools/summarize_pf_profile.py
# <tag>; python3 tools/summarize_tick_trace.py <tag> mode_S_h50=f_S50 ... (profiles/README.md)
set -u
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
$ROOT/tools/profile.sh $TAG > $ROOT/gpurun_out/final_profile_$TAG.log 2>&1
echo "profile.sh done"
(cd /tmp && export TMPDIR=/tmp && rm -rf $ROOT/gpurun_out/allk_$TAG && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/allk_$TAG -- python3 $ROOT/bench.py --steps 50 --warmup 10 > $ROOT/gpurun_out/allk_$TAG.json 2> $ROOT/gpurun_out/allk_$TAG.err)
echo "all-kernels trace done"
for spec in "f_S50 S 50 - -" "f_S25 S 25 - -" "f_S100 S 100 - -" "f_S50_pair S 50 ACMPC_NO_QUAD_ROUNDS -" "f_T50 T 50 - -" \
            "f_T50_one T 50 ACMPC_NO_TRIO_ROUNDS -" "f_T50_w25 T 50 - 2,5"; do
  set -- $spec
  rm -rf $ROOT/gpurun_out/trace_$1
  ( export TICK_MODE=$2 TICK_H=$3; [ "$4" != "-" ] && export $4=1; [ "$5" != "-" ] && export TICK_WINDOW=$5; $ROOT/tools/tick_trace.sh $1 | sed -n 1p )
done
echo "tick traces done"
rm -rf $ROOT/gpurun_out/pmc_${TAG}_T_*
$ROOT/tools/pmc_modeT.sh $TAG > $ROOT/gpurun_out/pmc_modeT_run.log 2>&1
tail -3 $ROOT/gpurun_out/pmc_modeT_run.log
rm -rf $ROOT/gpurun_out/pmc_${TAG}_sampled_*
$ROOT/tools/pmc_sampled.sh $TAG > $ROOT/gpurun_out/pmc_sampled_run.log 2>&1
tail -1 $ROOT/gpurun_out/pmc_sampled_run.log
$ROOT/tools/profile_pf.sh $TAG > $ROOT/gpurun_out/profile_pf_run.log 2>&1
echo "particle filter traces done"
