#!/bin/bash
# Development probe (GPU box): SQ counters of the fused sample + rollout kernel at bench.py's `sampled_fused_16M` scale.
# usage: tools/pmc_sampled.sh <tag>   then   python3 tools/summarize_sq_counters.py <tag> sampled
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
# (which sources these counters were collected from: bench.py compares it with the library it runs)
python3 -c "import sys; sys.path[:0] = ['$ROOT', '$ROOT/ac-mpc_amd']; import bench; print(bench.loaded_source_hash())" > $ROOT/gpurun_out/pmc_${TAG}_sampled_source_sha256.txt 2>/dev/null
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_ANY \
  --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_${TAG}_sampled_a -- python3 $ROOT/tools/run_sampled_fused.py > $ROOT/gpurun_out/pmc_${TAG}_sampled.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS \
  --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_${TAG}_sampled_b -- python3 $ROOT/tools/run_sampled_fused.py >> $ROOT/gpurun_out/pmc_${TAG}_sampled.log 2>&1
python3 $ROOT/tools/run_sampled_fused.py 8 > $ROOT/gpurun_out/pmc_${TAG}_sampled_unprofiled.log 2>&1
cat $ROOT/gpurun_out/pmc_${TAG}_sampled_unprofiled.log
