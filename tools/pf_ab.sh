#!/bin/bash
# Development aid (GPU box): the particle-scoring kernels under rocprofv3 --kernel-trace --stats for several settings of the
# library's A/B switches, alternating, on one box.   usage: tools/pf_ab.sh [label="VAR=1 ..."] ...
# default: the tree's default against ACMPC_PF_WORKGROUP_SCORE=1 (pf_score_kernel<8> behind the grid search).
# A label `head` runs tools/_ab_old/head/acmpc_pf.hip (an earlier source, see tools/ab_old.sh) instead; BUILD=<flag> in a
# label's settings builds the library with that compiler flag first (tools/ab_build.sh).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
[ $# -eq 0 ] && set -- tree= workgroup="ACMPC_PF_WORKGROUP_SCORE=1"
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
  for spec in "$@"; do
    label=${spec%%=*}; assigns=${spec#*=}
    (
      for kv in $assigns; do
        case "$kv" in
          BUILD=*) export ACMPC_HIP_LIBRARY=$("$ROOT/tools/ab_build.sh" pf_$label "${kv#BUILD=}" | tail -1) ;;   # one compiler flag
          *) export "$kv" ;;
        esac
      done
      if [ "$label" = head ]; then export ACMPC_HIP_LIBRARY=$(AB_OLD_DIR=head "$ROOT/tools/ab_old.sh" headpf acmpc_pf.hip | tail -1); fi
      for P in 500 100000; do
        rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/pfab_${label}_${rep}_$P -- python3 $ROOT/tools/bench_pf.py $P \
          > $ROOT/gpurun_out/pfab_${label}_${rep}_$P.log 2>&1
        echo "$label rep $rep P=$P: $(grep -h 'scoring call' $ROOT/gpurun_out/pfab_${label}_${rep}_$P.log | tail -1)"
        python3 - $ROOT/gpurun_out/pfab_${label}_${rep}_$P <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv"):
    for row in csv.DictReader(open(f)):
        if "pf_" in row["Name"]:
            print("    %-60s %8.1f us avg  %8.1f us min  x%s" % (row["Name"].replace("(anonymous namespace)::", "")[:60],
                  float(row["AverageNs"]) / 1e3, float(row["MinNs"]) / 1e3, row["Calls"]))
PY
      done
    )
  done
done
