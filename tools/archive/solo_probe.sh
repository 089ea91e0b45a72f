#!/bin/bash
# usage (GPU box): tools/solo_probe.sh [extra hipcc flags]   -> phase stamps of rollout_solo_kernel for the bench's single-solve shapes
set -eu
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DACMPC_STAMPS "$@" "$ROOT/tools/solo_probe.hip" -o /tmp/solo_probe 2> /tmp/solo_probe.build.log
for spec in "4096 49 1" "65536 49 1" "32768 79 1" "4096 49 0"; do
  /tmp/solo_probe $spec
done
