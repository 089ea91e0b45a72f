#!/bin/bash
# Development probe (GPU box): SQ counter passes of the closed-loop tick (tools/tick_probe.py); means per launch of the
# fused round kernel by tools/pmc_means.py rollout_sampled gpurun_out/pmc_tick_<name>_{a,b,c}
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
NAME=${1:-r}
cd /tmp && export TMPDIR=/tmp
LOG=$ROOT/gpurun_out/pmc_tick_${NAME}.log
: > $LOG
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_ANY \
  --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_tick_${NAME}_a -- python3 $ROOT/tools/tick_probe.py >> $LOG 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS \
  --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_tick_${NAME}_b -- python3 $ROOT/tools/tick_probe.py >> $LOG 2>&1
grep "tick p50" $LOG
