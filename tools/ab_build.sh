#!/bin/bash
# Development aid (GPU box): an A/B build of the library with extra hipcc flags, in a scratch copy of the package, so that
# the tree's library stays the default build.   usage: tools/ab_build.sh <name> "<flags>"  ->  /tmp/ab_<name>/libacmpc_hip.so
set -eu
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
NAME=$1; FLAGS=${2:-}
rm -rf /tmp/ab_$NAME && mkdir -p /tmp/ab_$NAME && cp -r "$ROOT/ac-mpc_amd" "$ROOT/include" /tmp/ab_$NAME/ && rm -rf /tmp/ab_$NAME/ac-mpc_amd/acmpc_amd/lib
(cd /tmp/ab_$NAME/ac-mpc_amd && ACMPC_HIPCC_EXTRA="$FLAGS" python3 acmpc_amd/_build.py > /tmp/ab_$NAME/build.log 2>&1)
cp /tmp/ab_$NAME/ac-mpc_amd/acmpc_amd/lib/libacmpc_hip.so /tmp/ab_$NAME/libacmpc_hip.so
echo /tmp/ab_$NAME/libacmpc_hip.so
