#!/bin/bash
# Development probe (GPU box): kernel-by-kernel durations of the closed-loop tick (tools/tick_probe.py under
# rocprofv3 --kernel-trace): prints the median duration of each kernel, the sampled rounds split by their order in a tick.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
NAME=${1:-tick}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/trace_${NAME} -- python3 $ROOT/tools/tick_probe.py > $ROOT/gpurun_out/trace_${NAME}.log 2>&1
tail -1 $ROOT/gpurun_out/trace_${NAME}.log
python3 - "$ROOT/gpurun_out/trace_${NAME}" <<'PY'
import csv, glob, os, sys, statistics
rows = []
for path in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    with open(path, newline="") as handle:
        for row in csv.DictReader(handle):
            rows.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), row["Kernel_Name"]))
rows.sort()
names = {"prologue": [], "round_a": [], "round_b": [], "gap_prologue_to_a": [], "gap_a_to_b": []}
i = 0
while i + 2 < len(rows):
    if "prologue_kernel" in rows[i][2] and "rollout_sampled" in rows[i + 1][2] and "rollout_sampled" in rows[i + 2][2]:
        names["prologue"].append(rows[i][1] - rows[i][0])
        names["round_a"].append(rows[i + 1][1] - rows[i + 1][0])
        names["round_b"].append(rows[i + 2][1] - rows[i + 2][0])
        names["gap_prologue_to_a"].append(rows[i + 1][0] - rows[i][1])
        names["gap_a_to_b"].append(rows[i + 2][0] - rows[i + 1][1])
        i += 3
    else:
        i += 1
for key, values in names.items():
    if values:
        print("%-20s median %7.2f us   p10 %7.2f   p90 %7.2f   (%d)" % (key, statistics.median(values) / 1e3,
              sorted(values)[len(values) // 10] / 1e3, sorted(values)[9 * len(values) // 10] / 1e3, len(values)))
PY
