#!/bin/bash
# Development aid (GPU box): A/B builds of the mode T rollout (flags of csrc/acmpc_kernels.hip) timed in turn on one box.
# usage: tools/modeT_ab.sh <P> name1="<flags>" name2="<flags>" ...     (name "tree" = the tree's library)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
P=$1; shift
declare -A LIBS
ORDER=()
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}
  if [ "$name" = "tree" ]; then LIBS[$name]=""; else LIBS[$name]=$($ROOT/tools/ab_build.sh $name "$flags" | tail -1); fi
  ORDER+=($name)
done
for round in 1 2 3; do
  for name in "${ORDER[@]}"; do
    echo "== $name (round $round)"
    if [ -z "${LIBS[$name]}" ]; then python3 $ROOT/tools/time_modeT.py $P; else ACMPC_HIP_LIBRARY=${LIBS[$name]} python3 $ROOT/tools/time_modeT.py $P; fi
  done
done
