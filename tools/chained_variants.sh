#!/bin/bash
# Development aid (GPU box): tools/chained_ab.py for A/B builds of the finalize's look-ahead inside the shared launch.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for v in "p2a4 -DACMPC_CHAINED_PASS=2 -DACMPC_CHAINED_AHEAD=4" "p4a8 -DACMPC_CHAINED_PASS=4 -DACMPC_CHAINED_AHEAD=8" "noprio -DACMPC_CHAINED_NO_PRIO"; do
  set -- $v; name=$1; shift
  bash $ROOT/tools/ab_build.sh $name "$*" > /dev/null
done
for rep in 1 2; do
  python3 $ROOT/tools/chained_ab.py
  for name in p2a4 p4a8 noprio; do ACMPC_HIP_LIBRARY=/tmp/ab_$name/libacmpc_hip.so python3 $ROOT/tools/chained_ab.py; done
done
