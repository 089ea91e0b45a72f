#!/bin/bash
# Development aid (GPU box): the headline step as a stream of batches (finalize inside the next rollout's launch) and with
# each form of the finalize as a launch of its own, interleaved repeats on one box,
# then a kernel trace of the default form.   usage: tools/finalize_ab.sh   ->  gpurun_out/finalize_ab.log
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/finalize_ab.log
: > $OUT
pick() { python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$1', 'ms_per_step %.4f kernel_ms %.4f ratio %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['ms_per_step']/d['roofline']['kernel_ms']))"; }
for rep in 1 2 3; do
  python3 $ROOT/bench.py --no-cpu-baseline --no-single-solve --steps 300 --warmup 30 2>/dev/null | pick stream >> $OUT
  python3 $ROOT/bench.py --no-stream --no-cpu-baseline --no-single-solve --steps 300 --warmup 30 2>/dev/null | pick groups >> $OUT
  ACMPC_FINALIZE_WAVES=1 python3 $ROOT/bench.py --no-stream --no-cpu-baseline --no-single-solve --steps 300 --warmup 30 2>/dev/null | pick four_waves >> $OUT
done
(cd /tmp && export TMPDIR=/tmp && rm -rf $ROOT/gpurun_out/finalize_ab_trace && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/finalize_ab_trace -- python3 $ROOT/bench.py --no-cpu-baseline --no-single-solve --steps 100 --warmup 10 > /dev/null 2> $ROOT/gpurun_out/finalize_ab_trace.err)
for f in $ROOT/gpurun_out/finalize_ab_trace/*/*_kernel_stats.csv; do cut -d, -f1-4 $f | sed -n 1,4p >> $OUT; done
cat $OUT
