"""Development tool: mean SQ counters per launch of the mode-T rollout kernel out of the raw rocprofv3 --pmc passes that
tools/pmc_modeT.sh leaves under gpurun_out/, written as profiles/<tag>_mode_T_sq_counters.json; with a second argument
`sampled`, of the fused sample + rollout kernel (tools/pmc_sampled.sh) -> profiles/<tag>_sampled_sq_counters.json.

usage: python3 tools/summarize_sq_counters.py r03 [sampled]"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEPS = 49
SIMDS = 1024


def counters(directory, kernel="::rollout_kernel<1"):
    sums, counts, micros = defaultdict(float), defaultdict(int), []
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as handle:
            for row in csv.DictReader(handle):
                if kernel not in row["Kernel_Name"]:
                    continue
                sums[row["Counter_Name"]] += float(row["Counter_Value"])
                counts[row["Counter_Name"]] += 1
    for path in glob.glob(os.path.join(directory, "**", "*kernel_trace.csv"), recursive=True):
        with open(path, newline="") as handle:
            for row in csv.DictReader(handle):
                if kernel in row["Kernel_Name"]:
                    micros.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    means = {name: sums[name] / counts[name] for name in sorted(sums)}
    launches = max(counts.values()) if counts else 0
    return means, launches, (sum(micros) / len(micros) if micros else None)


def collected_from(tag, kind):
    """sha256 of the sources the counters were collected from (written on the GPU box by tools/pmc_*.sh), or None."""
    path = os.path.join(ROOT, "gpurun_out", f"pmc_{tag}_{kind}_source_sha256.txt")
    try:
        value = open(path).read().strip()
    except OSError:
        return None
    return value if re.fullmatch(r"[0-9a-f]{64}", value) else None


def unprofiled(tag):
    out = {}
    path = os.path.join(ROOT, "gpurun_out", f"pmc_{tag}_T_unprofiled.log")
    if os.path.exists(path):
        for line in open(path):
            match = re.match(r"(T,\S+)\s+kernel median\s+([\d.]+) us\s+min\s+([\d.]+) us", line)
            if match:
                out[match.group(1)] = {"median_us": float(match.group(2)), "min_us": float(match.group(3))}
    return out


def sampled(tag, problems=1024, candidates=16384):
    """The fused sample + rollout round at bench.py's `sampled_fused_16M` scale: one wave = 64 candidates."""
    merged, launches, profiled_us = {}, 0, None
    for suffix in ("a", "b"):
        means, count, micros = counters(os.path.join(ROOT, "gpurun_out", f"pmc_{tag}_sampled_{suffix}"), "rollout_sampled_kernel<0")
        merged.update(means)
        launches = max(launches, count)
        profiled_us = profiled_us or micros
    if not merged:
        raise SystemExit("no counters under gpurun_out/pmc_%s_sampled_*" % tag)
    waves, valu = merged.get("SQ_WAVES", 0.0), merged.get("SQ_INSTS_VALU", 0.0)
    unprofiled_line = None
    path = os.path.join(ROOT, "gpurun_out", f"pmc_{tag}_sampled_unprofiled.log")
    if os.path.exists(path):
        lines = [line.strip() for line in open(path) if "kernel_us" in line]
        unprofiled_line = lines[-1] if lines else None
    derived = {
        "launches_averaged": launches,
        "candidates_per_wave": problems * candidates / waves if waves else None,
        # the horizon's steps are not the whole kernel: the Philox draws, the staging and the winner's re-roll are in the count
        "valu_instructions_per_candidate_step": valu / waves / STEPS if waves else None,
        "salu_instructions_per_wave_step": merged.get("SQ_INSTS_SALU", 0.0) / waves / STEPS if waves else None,
        "lds_instructions_per_wave_step": merged.get("SQ_INSTS_LDS", 0.0) / waves / STEPS if waves else None,
        "lds_bank_conflict_cycles": merged.get("SQ_LDS_BANK_CONFLICT"),
        "valu_issue_cycles_per_simd_at_4_per_instruction": valu * 4 / SIMDS,
        "gpu_active_cycles_per_xcd": (merged["GRBM_GUI_ACTIVE"] / 8 if "GRBM_GUI_ACTIVE" in merged else None),
        "share_of_wave_cycles_waiting_on_any_counter": (merged["SQ_WAIT_ANY"] / merged["SQ_WAVE_CYCLES"]
                                                        if "SQ_WAIT_ANY" in merged and "SQ_WAVE_CYCLES" in merged else None),
        "kernel_us_under_the_profiler": profiled_us,
        "unprofiled_same_box": unprofiled_line,
    }
    result = {
        "kernel": "rollout_sampled_kernel<mode S> (one wave per workgroup, 64 candidates drawn and rolled in the launch, "
                  "winner re-rolled by each problem's last workgroup) on %d problems x %d candidates x horizon 50" % (problems, candidates),
        "command": "tools/pmc_sampled.sh " + tag + ": two rocprofv3 --pmc passes with --kernel-trace round tools/run_sampled_fused.py, "
                   "then the same un-profiled",
        "fused_round": {"counters_mean_per_launch": merged, "derived": derived},
        "source_sha256": collected_from(tag, "sampled"),
    }
    path = os.path.join(ROOT, "profiles", f"{tag}_sampled_sq_counters.json")
    with open(path, "w") as handle:
        json.dump(result, handle, indent=1)
    print(json.dumps(derived, indent=1))


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
    if len(sys.argv) > 2 and sys.argv[2] == "sampled":
        return sampled(tag)
    plain = unprofiled(tag)
    result = {
        "kernel": "rollout_kernel<mode T, step-major, 2 candidates per lane, 256 threads, plain float32> on 256 poses x "
                  "4096 candidates x horizon 50 (1 048 576 candidates per launch)",
        "command": "tools/pmc_modeT.sh " + tag + ": two rocprofv3 --pmc passes (SQ issue counters; LDS / wait counters) with "
                   "--kernel-trace per search window, then the same tools/sweep.py specs un-profiled",
    }
    for name, spec in (("window_2_5", "T,1,256,4096,50,2,5"), ("window_1_2", "T,1,256,4096,50,1,2"),
                       ("exhaustive", "T,1,256,4096,50")):
        short = {"window_2_5": "2_5", "window_1_2": "1_2", "exhaustive": "exhaustive"}[name]
        merged, launches, profiled_us = {}, 0, None
        for suffix in ("a", "b"):
            means, count, micros = counters(os.path.join(ROOT, "gpurun_out", f"pmc_{tag}_T_{short}_{suffix}"))
            merged.update(means)
            launches = max(launches, count)
            profiled_us = profiled_us or micros
        if not merged:
            continue
        waves = merged.get("SQ_WAVES", 0.0)
        valu = merged.get("SQ_INSTS_VALU", 0.0)
        candidates_per_wave = 256 * 4096 / waves if waves else 0.0
        derived = {
            "launches_averaged": launches,
            "candidates_per_wave": candidates_per_wave,
            "valu_instructions_per_wave_step": valu / waves / STEPS if waves else None,
            "valu_instructions_per_candidate_step": (valu / waves / STEPS / (candidates_per_wave / 64)
                                                     if waves else None),
            "salu_instructions_per_wave_step": merged.get("SQ_INSTS_SALU", 0.0) / waves / STEPS if waves else None,
            "lds_instructions_per_wave_step": merged.get("SQ_INSTS_LDS", 0.0) / waves / STEPS if waves else None,
            "lds_bank_conflict_cycles": merged.get("SQ_LDS_BANK_CONFLICT"),
            "valu_issue_cycles_per_simd_at_4_per_instruction": valu * 4 / SIMDS,
            "gpu_active_cycles_per_xcd": (merged["GRBM_GUI_ACTIVE"] / 8 if "GRBM_GUI_ACTIVE" in merged else None),
            "share_of_wave_cycles_waiting_on_any_counter": (merged["SQ_WAIT_ANY"] / merged["SQ_WAVE_CYCLES"]
                                                            if "SQ_WAIT_ANY" in merged and "SQ_WAVE_CYCLES" in merged
                                                            else None),
            "kernel_us_under_the_profiler": profiled_us,
            "kernel_us_unprofiled_same_box": plain.get(spec),
        }
        result[name] = {"counters_mean_per_launch": merged, "derived": derived}
    result["source_sha256"] = collected_from(tag, "T")
    result["note"] = ("valu_issue_cycles_per_simd_at_4_per_instruction (SQ_INSTS_VALU x 4 cycles / 1024 SIMDs, against "
                      "GRBM_GUI_ACTIVE / 8 XCDs - the counter is summed over the dies) is the uniform four-cycle estimate of "
                      "rounds 2-3 and can exceed the active cycles: not every vector instruction takes four.  bench.py's "
                      "roofline_valu prices the COUNTED instructions of this file with the per-opcode issue times of "
                      "<tag>_valu_probe.json over the step loop's opcode mix of <tag>_isa_mix.json instead (DESIGN 4.1).  "
                      "Round 2's kernel counted 96.8 VALU instructions per candidate-step with window (2,5), round 3's 84.4 with "
                      "24 LDS instructions per wave-step and 21 % of the wave-cycles waiting for the LDS; this round's key table "
                      "(32-byte entries, two ds_read_b128 per four waypoints) leaves 16 and under 2 %.")
    path = os.path.join(ROOT, "profiles", f"{tag}_mode_T_sq_counters.json")
    with open(path, "w") as handle:
        json.dump(result, handle, indent=1)
    print(json.dumps({k: v["derived"] for k, v in result.items() if isinstance(v, dict)}, indent=1))


if __name__ == "__main__":
    main()
