#!/bin/bash
# Development probe (GPU box): SQ / L2 counters of the particle-scoring kernels (tools/bench_pf.py), one particle count per
# run.  usage: tools/pmc_pf.sh <tag> [library]   (library: an ACMPC_HIP_LIBRARY to profile instead of the tree's - the
# round-4 kernels for the before / after pair);  then here: python3 tools/summarize_pf_counters.py <tag>
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r05}
if [ -n "${2:-}" ]; then export ACMPC_HIP_LIBRARY=$2; fi
cd /tmp && export TMPDIR=/tmp
for P in 500 100000; do
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU \
    --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_${TAG}_pf_${P}_a -- python3 $ROOT/tools/bench_pf.py $P > $ROOT/gpurun_out/pmc_${TAG}_pf_$P.log 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum \
    --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_${TAG}_pf_${P}_b -- python3 $ROOT/tools/bench_pf.py $P >> $ROOT/gpurun_out/pmc_${TAG}_pf_$P.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/pf_${TAG}_$P -- python3 $ROOT/tools/bench_pf.py $P >> $ROOT/gpurun_out/pmc_${TAG}_pf_$P.log 2>&1
  tail -1 $ROOT/gpurun_out/pmc_${TAG}_pf_$P.log
done
