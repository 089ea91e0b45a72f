#!/bin/bash
# Development aid (GPU box): the wave-by-wave timeline (tools/modeT_stamps.py) of several A/B builds of the mode T rollout.
# usage: tools/modeT_stamps_ab.sh <P> name1="<flags>" ...   -> gpurun_out/modeT_stamps_<name>.json
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
P=$1; shift
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}
  LIB=$($ROOT/tools/ab_build.sh st_$name "-DACMPC_T_STAMPS $flags" | tail -1)
  STAMP_REPEATS=${STAMP_REPEATS:-2} ACMPC_HIP_LIBRARY=$LIB python3 $ROOT/tools/modeT_stamps.py $P $ROOT/gpurun_out/modeT_stamps_$name.json > $ROOT/gpurun_out/modeT_stamps_$name.log 2>&1
  echo "$name done"
done
