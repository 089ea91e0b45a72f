#!/bin/bash
# Development probe (GPU box): SQ counter passes of ONE tools/sweep.py spec (any kernel), plus the un-profiled timing.
# usage: tools/pmc_sweep.sh <name> <spec>      (environment switches such as ACMPC_NO_TILE=1 are inherited)
# -> gpurun_out/pmc_<name>_{a,b,c}/ (raw), gpurun_out/pmc_<name>.log
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
NAME=$1
SPEC=$2
cd /tmp && export TMPDIR=/tmp
LOG=$ROOT/gpurun_out/pmc_${NAME}.log
: > $LOG
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_ANY \
  --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_${NAME}_a -- python3 $ROOT/tools/sweep.py $SPEC >> $LOG 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS \
  --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_${NAME}_b -- python3 $ROOT/tools/sweep.py $SPEC >> $LOG 2>&1
rocprofv3 --pmc SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_IFETCH SQ_INST_LEVEL_LDS \
  --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_${NAME}_c -- python3 $ROOT/tools/sweep.py $SPEC >> $LOG 2>&1
python3 $ROOT/tools/sweep.py $SPEC >> $LOG 2>&1
grep "kernel median" $LOG
