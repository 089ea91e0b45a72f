#!/bin/bash
# On the GPU box: rocprofv3 --kernel-trace --stats of the particle-scoring seam (tools/bench_pf.py), ONE particle count per
# run so that a kernel's average belongs to one size: 500 (the reference's, configs/monza.yaml:47 - the exhaustive scan,
# pf_score_kernel<1>) and 100 000 (the uniform grid: pf_nearest_kernel + pf_score_kernel<8>).
# usage: tools/profile_pf.sh <tag>   then here:   python3 tools/summarize_pf_profile.py <tag>
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp
for P in 500 100000; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/pf_${TAG}_$P -- python3 $ROOT/tools/bench_pf.py $P \
    > $ROOT/gpurun_out/pf_${TAG}_$P.log 2>&1
  tail -1 $ROOT/gpurun_out/pf_${TAG}_$P.log
done
