#!/bin/bash
# Collects the rocprofv3 evidence for bench.py's dominant kernel on the GPU box (run through gpurun):
#   pass 1  --kernel-trace --stats         per-kernel durations of the default bench command
#   pass 2+ --pmc ... (own runs, no trace domains besides kernel-trace)   SQ activity, HBM fetch/write bytes
# Output: gpurun_out/prof_<tag>/...  then summarised by tools/summarize_profile.py into profiles/.
# usage: tools/profile.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf "$OUT"   # never mix runs: the summariser averages every CSV it finds under this directory
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-single-solve $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/bench_trace.json" 2> "$OUT/trace.err"
SHORT="python3 $ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-single-solve $*"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY \
  --kernel-trace --output-format csv -d "$OUT/pmc_sq" -- $SHORT > /dev/null 2> "$OUT/pmc_sq.err"
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- $SHORT > /dev/null 2> "$OUT/pmc_fetch.err"
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d "$OUT/pmc_write" -- $SHORT > /dev/null 2> "$OUT/pmc_write.err"
python3 $ROOT/bench.py --steps 200 --warmup 20 $* > "$OUT/bench_plain.json" 2> "$OUT/bench_plain.err"
cat "$OUT/bench_plain.json"
