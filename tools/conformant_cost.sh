#!/bin/bash
# On the GPU box: what ACMPC_CONFORMANT_SYNC=1 costs.  The same short bench.py command twice in
# turn, twice over - the default forms and the conformant ones - and the three latencies / the headline step side by side.
# usage: tools/conformant_cost.sh [out.json]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=${1:-$ROOT/gpurun_out/conformant_cost.json}
for round in 1 2; do
  python3 $ROOT/bench.py --steps 100 --warmup 10 --cpu-seconds 1 > $ROOT/gpurun_out/conformant_default_$round.json 2> /dev/null
  ACMPC_CONFORMANT_SYNC=1 python3 $ROOT/bench.py --steps 100 --warmup 10 --cpu-seconds 1 > $ROOT/gpurun_out/conformant_on_$round.json 2> /dev/null
done
python3 - "$ROOT" "$OUT" <<'PY'
import json, sys
root, out = sys.argv[1], sys.argv[2]
def line(path):
    rows = [l for l in open(path) if l.startswith("{")]
    return json.loads(rows[-1])
table = {}
for form in ("default", "on"):
    runs = [line("%s/gpurun_out/conformant_%s_%d.json" % (root, form, k)) for k in (1, 2)]
    pick = lambda f: [f(r) for r in runs]
    table["default forms" if form == "default" else "ACMPC_CONFORMANT_SYNC=1"] = {
        "single_solve_us_p50 (4 096 x 49, device-resident)": pick(lambda r: r["single_solve"]["device_resident_us_p50"]),
        "config3_single_us_p50 (65 536 x 49)": pick(lambda r: r["config3_single"]["device_resident_us_p50"]),
        "tick_ms_p50 (get_control_at, mode S)": pick(lambda r: r["closed_loop_replay"]["solve_ms_p50"]),
        "tick_mode_T_ms_p50": pick(lambda r: r["closed_loop_replay_mode_T"]["solve_ms_p50"]),
        "tick_mode_T_window_2_5_ms_p50": pick(lambda r: r["closed_loop_replay_mode_T_window_2_5"]["solve_ms_p50"]),
        "headline ms_per_step": pick(lambda r: r["ms_per_step"]),
        "headline kernel_ms": pick(lambda r: r["roofline"]["kernel_ms"]),
        "headline kernel": runs[0]["roofline"]["kernel"],
    }
json.dump({"tool": "tools/conformant_cost.sh (two interleaved pairs of `bench.py --steps 100 --warmup 10` on one box)", "table": table}, open(out, "w"), indent=1)
print(json.dumps(table, indent=1))
PY
