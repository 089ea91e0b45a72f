#!/bin/bash
# Development aid (GPU box): the library with some sources replaced by earlier versions kept in tools/_ab_old/ (not
# tracked: `git show <commit>:ac-mpc_amd/csrc/<file> > tools/_ab_old/<file>`), in a scratch copy of the package, for
# before / after pairs on one box (AB_OLD_DIR=<subdirectory of tools/_ab_old> for a second set).
# usage: tools/ab_old.sh <name> <file> ...   ->  /tmp/ab_<name>/libacmpc_hip.so
set -eu
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
NAME=$1; shift
rm -rf /tmp/ab_$NAME && mkdir -p /tmp/ab_$NAME && cp -r "$ROOT/ac-mpc_amd" "$ROOT/include" /tmp/ab_$NAME/ && rm -rf /tmp/ab_$NAME/ac-mpc_amd/acmpc_amd/lib
for f in "$@"; do cp "$ROOT/tools/_ab_old/${AB_OLD_DIR:-.}/$f" /tmp/ab_$NAME/ac-mpc_amd/csrc/$f; done
(cd /tmp/ab_$NAME/ac-mpc_amd && python3 acmpc_amd/_build.py > /tmp/ab_$NAME/build.log 2>&1)
cp /tmp/ab_$NAME/ac-mpc_amd/acmpc_amd/lib/libacmpc_hip.so /tmp/ab_$NAME/libacmpc_hip.so
echo /tmp/ab_$NAME/libacmpc_hip.so
