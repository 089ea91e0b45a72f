#!/bin/bash
# Development aid (GPU box): the library with round 4's particle-filter kernels (tools/_ab_old/acmpc_pf.hip =
# `git show 50e7ed8:ac-mpc_amd/csrc/acmpc_pf.hip`) -> /tmp/ab_oldpf/libacmpc_hip.so, for before / after pairs on one box.
exec "$(dirname "$0")/ab_old.sh" oldpf acmpc_pf.hip
