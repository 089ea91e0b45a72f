#!/bin/bash
# Development aid (GPU box): the library with round 4's particle-filter kernels (tools/_ab_old/acmpc_pf.hip) in a scratch
# copy of the package -> /tmp/ab_oldpf/libacmpc_hip.so, for before / after pairs on one box.
set -eu
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
rm -rf /tmp/ab_oldpf && mkdir -p /tmp/ab_oldpf && cp -r "$ROOT/ac-mpc_amd" "$ROOT/include" /tmp/ab_oldpf/ && rm -rf /tmp/ab_oldpf/ac-mpc_amd/acmpc_amd/lib
cp "$ROOT/tools/_ab_old/acmpc_pf.hip" /tmp/ab_oldpf/ac-mpc_amd/csrc/acmpc_pf.hip
(cd /tmp/ab_oldpf/ac-mpc_amd && python3 acmpc_amd/_build.py > /tmp/ab_oldpf/build.log 2>&1)
cp /tmp/ab_oldpf/ac-mpc_amd/acmpc_amd/lib/libacmpc_hip.so /tmp/ab_oldpf/libacmpc_hip.so
echo /tmp/ab_oldpf/libacmpc_hip.so
