#!/bin/bash
# Development aid (GPU box): the particle-scoring kernels of the tree and of tools/_ab_old/head/acmpc_pf.hip (the last commit's)
# under rocprofv3 --kernel-trace --stats, alternating, on one box.   usage: tools/pf_ab.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OLD=$(AB_OLD_DIR=head "$ROOT/tools/ab_old.sh" headpf acmpc_pf.hip | tail -1)
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
  for which in new old; do
    if [ $which = old ]; then export ACMPC_HIP_LIBRARY=$OLD; else unset ACMPC_HIP_LIBRARY; fi
    for P in 500 100000; do
      rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/pfab_${which}_${rep}_$P -- python3 $ROOT/tools/bench_pf.py $P \
        > $ROOT/gpurun_out/pfab_${which}_${rep}_$P.log 2>&1
      echo "$which rep $rep P=$P"; grep -h "pf_" $ROOT/gpurun_out/pfab_${which}_${rep}_$P/*/*_kernel_stats.csv | cut -d, -f1,4 | sed 's/(anonymous namespace):://g'
    done
  done
done
