#!/bin/bash
# Development experiment (GPU box): mode T rollout, 1 M candidates per launch, launch-shape / packing variants and a
# build without the SLP vectoriser.  usage: tools/exp_modeT.sh  -> gpurun_out/exp_modeT.log
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
LOG=gpurun_out/exp_modeT.log
mkdir -p gpurun_out
: > $LOG
SPECS='T,1,256,4096,50,2,5 T,1,256,4096,50,1,2 T,1,256,4096,50'
run() { echo "== $1" >> $LOG; shift; env "$@" python3 tools/sweep.py $SPECS >> $LOG 2>&1; }
run "default (256,4 packed pairs)"            ACMPC_X=0
run "256,4 plain float32 source"               ACMPC_T_PACK=1
run "256,2 plain"                               ACMPC_SHAPE=256,2 ACMPC_T_PACK=1
# the A/B library goes through the package's own build (all sources, their per-source flags + the extra one) into a
# scratch copy of the package, so that the tree's library stays the default build
rm -rf /tmp/acmpc_noslp && mkdir -p /tmp/acmpc_noslp && cp -r ac-mpc_amd include /tmp/acmpc_noslp/ && rm -rf /tmp/acmpc_noslp/ac-mpc_amd/acmpc_amd/lib
(cd /tmp/acmpc_noslp/ac-mpc_amd && ACMPC_HIPCC_EXTRA=-fno-slp-vectorize python3 acmpc_amd/_build.py) >> $LOG 2>&1 || exit 1
cp /tmp/acmpc_noslp/ac-mpc_amd/acmpc_amd/lib/libacmpc_hip.so /tmp/libacmpc_noslp.so
run "no-slp: 256,4 plain"                       ACMPC_HIP_LIBRARY=/tmp/libacmpc_noslp.so ACMPC_T_PACK=1
run "no-slp: 256,2 plain"                       ACMPC_HIP_LIBRARY=/tmp/libacmpc_noslp.so ACMPC_SHAPE=256,2 ACMPC_T_PACK=1
run "no-slp: 256,1"                             ACMPC_HIP_LIBRARY=/tmp/libacmpc_noslp.so ACMPC_SHAPE=256,1
cd /tmp && export TMPDIR=/tmp
export ACMPC_HIP_LIBRARY=/tmp/libacmpc_noslp.so ACMPC_T_PACK=1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_ANY \
  --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_T_a -- python3 $ROOT/tools/sweep.py T,1,256,4096,50,2,5 >> $ROOT/$LOG 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT \
  --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_T_b -- python3 $ROOT/tools/sweep.py T,1,256,4096,50,2,5 >> $ROOT/$LOG 2>&1
cat $ROOT/$LOG
