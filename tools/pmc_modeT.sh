#!/bin/bash
# Development probe (GPU box): SQ counters of the mode-T rollout (default launch shape), window 2,5 and 1,2.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
# (which sources these counters were collected from: bench.py compares it with the library it runs)
python3 -c "import sys; sys.path[:0] = ['$ROOT', '$ROOT/ac-mpc_amd']; import bench; print(bench.loaded_source_hash())" > $ROOT/gpurun_out/pmc_${TAG}_T_source_sha256.txt 2>/dev/null
for W in "2,5" "1,2" ""; do
  name=${W/,/_}; name=${name:-exhaustive}
  spec="T,1,256,4096,50${W:+,$W}"
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_ANY \
    --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_${TAG}_T_${name}_a -- python3 $ROOT/tools/sweep.py $spec > $ROOT/gpurun_out/pmc_${TAG}_T_${name}.log 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS \
    --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_${TAG}_T_${name}_b -- python3 $ROOT/tools/sweep.py $spec >> $ROOT/gpurun_out/pmc_${TAG}_T_${name}.log 2>&1
done
python3 $ROOT/tools/sweep.py T,1,256,4096,50,2,5 T,1,256,4096,50,1,2 T,1,256,4096,50 > $ROOT/gpurun_out/pmc_${TAG}_T_unprofiled.log 2>&1
cat $ROOT/gpurun_out/pmc_${TAG}_T_unprofiled.log
