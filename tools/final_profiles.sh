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
# round 5: the particle filter's counters (the tree's kernels, and round 4's beside them when tools/_ab_old/ is there), the
# mode T launch wave by wave, what the conformant forms cost
rm -rf $ROOT/gpurun_out/pmc_${TAG}new_pf_* $ROOT/gpurun_out/pmc_${TAG}old_pf_* $ROOT/gpurun_out/pf_${TAG}new_* $ROOT/gpurun_out/pf_${TAG}old_*
$ROOT/tools/pmc_pf.sh ${TAG}new > $ROOT/gpurun_out/pmc_pf_run.log 2>&1
if [ -f $ROOT/tools/_ab_old/acmpc_pf.hip ]; then $ROOT/tools/pmc_pf.sh ${TAG}old $($ROOT/tools/ab_old_pf.sh | tail -1) >> $ROOT/gpurun_out/pmc_pf_run.log 2>&1; fi
echo "particle filter counters done"
STAMP_REPEATS=2 $ROOT/tools/modeT_stamps_ab.sh 256 ${TAG}= > /dev/null 2>&1
for k in 1 2 3; do python3 $ROOT/tools/time_modeT.py 256; done > $ROOT/gpurun_out/time_modeT_${TAG}.log 2>&1
python3 $ROOT/tools/time_modeT.py 4096 >> $ROOT/gpurun_out/time_modeT_${TAG}.log 2>&1
echo "mode T timeline done"
$ROOT/tools/conformant_cost.sh $ROOT/gpurun_out/conformant_cost_${TAG}.json > /dev/null 2>&1
echo "conformant cost done"
