"""bench.py end to end on the GPU box: the one-GPU line, and a two-rank rehearsal of the multi-GPU path (both ranks
share the card; the collective payloads go through gloo) which also runs bench.py's cross-rank self-check."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _last_json(text):
    lines = [line for line in text.splitlines() if line.startswith("{")]
    assert lines, text[-2000:]
    return json.loads(lines[-1])


def test_single_gpu_line_has_the_contract_fields():
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--poses", "32", "--steps", "5", "--warmup",
                           "2", "--cpu-seconds", "1"], capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stderr[-2000:]
    out = _last_json(proc.stdout)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in out, key
    assert out["n_gpus"] == 1 and out["steps"] == 5 and out["dtype"] == "f32" and out["vs_baseline"] is None
    assert out["value"] > 1e6 and out["scaling"] == "weak" and "workload" in out["config"]
    roof = out["roofline"]
    assert roof["bound"] == "hbm" and roof["peak"] == 8000.0 and 0 < roof["frac"] < 1
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-12
    cpu = out["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["cores"] >= 1 and cpu["value"] > 0
    assert out["closed_loop_replay"]["infeasible_solves"] == 0
    # every number DESIGN.md quotes travels in the driver's line
    assert cpu["numpy"]["cores"] == 1 and 0 < cpu["numpy"]["value"] < cpu["value"]
    assert cpu["numpy_all_cores"]["cores"] >= 1 and cpu["numpy_all_cores"]["value"] > 0      # SURVEY 8d's second figure
    second = out["secondary_kernels"]
    for name in ("mode_S_candidate_major", "mode_T_exhaustive_search", "mode_T_window_2_5", "mode_T_window_1_2",
                 "mode_S_step_major_1M", "mode_T_window_2_5_16M", "mode_T_exhaustive_search_16M",
                 "mode_S_candidate_major_16M"):
        assert 0 < second[name]["frac_of_hbm_roofline"] < 1 and second[name]["kernel_us"] > 0, name
    assert second["mode_S_candidate_major_16M"]["kernel_us"] > 4 * second["mode_S_candidate_major"]["kernel_us"]
    # instruction-bound kernels carry the roof they are under (VALU issue), the fused sampled round at scale too
    for name in ("mode_T_exhaustive_search", "mode_T_window_2_5", "mode_T_window_1_2", "mode_T_window_2_5_16M",
                 "mode_T_exhaustive_search_16M"):
        valu, lds = second[name]["roofline_valu"], second[name]["roofline_lds"]
        # an issue-time roof (per-opcode issue times measured at the kernel's occupancy x its opcode mix): a roof, so <= 1
        assert valu["bound"] == "valu_issue" and 0.3 < valu["frac"] <= 1.0, (name, valu["frac"])
        assert valu["valu_instructions_per_candidate_step"] > 20 and 0.9 < valu["mean_issue_ns_per_instruction_per_simd"] < 2.5
        assert valu["instruction_count_source"].startswith("profiles/") and valu["opcode_mix_source"].startswith("profiles/")
        assert valu["opcode_mix_matches_loaded_sources"] is True, "profiles/*_isa_mix.json is stale: python3 tools/isa_mix.py"
        assert lds["bound"] == "lds_array" and 0 < lds["frac"] <= 1.0
    fused = second["sampled_fused_16M"]
    assert fused["kernel_us"] > 0 and fused["candidate_trajectories_per_s"] > 1e9 and fused["winners_finite"]
    assert 0.3 < fused["roofline_valu"]["frac"] <= 1.0
    assert out["roofline"]["traffic_measured_in_this_run"] is False
    loop_t = out["closed_loop_replay_mode_T"]
    assert "rollout mode T" in loop_t["workload"] and loop_t["infeasible_solves"] == 0 and loop_t["solve_ms_p50"] > 0
    assert "nearest waypoint of all" in loop_t["workload"]             # the default: the verified search in the rounds
    loop_w = out["closed_loop_replay_mode_T_window_2_5"]
    assert "nn_window (2, 5)" in loop_w["workload"] and loop_w["infeasible_solves"] == 0
    assert loop_t["solve_ms_p50"] < 1.5 * loop_w["solve_ms_p50"]        # (a scan of every waypoint took 2.5 x the window)
    assert out["single_solve"]["device_resident_us_p50"] < 16.0      # one launch (round 2: two launches, 19 us)
    assert out["config3_single"]["device_resident_us_p50"] > 0 and out["config4_share"]["device_resident_us_p50"] > 0
    five = out["config5_host_pointer"]
    assert "10000 consecutive poses" in five["workload"] and 0 < five["solve_us_p50"] <= five["solve_us_p99"]
    pf = out["particle_filter"]
    assert pf["particles_500"]["update_us"] > 0 and pf["particles_100000"]["particles_per_s"] > 1e6
    # the driver keeps `roofline`'s scalar keys and only the names of the nested blocks: every secondary figure the review
    # reads is there as a scalar too, and equals the block it was copied from
    sys.path.insert(0, ROOT)
    import bench
    for key in bench.SECONDARY_SCALARS:
        assert key in roof and isinstance(roof[key], (int, float)) and not isinstance(roof[key], bool), key
    assert roof["mode_T_window_2_5_frac_1M"] == second["mode_T_window_2_5"]["frac_of_hbm_roofline"]
    assert roof["mode_T_exhaustive_frac_16M"] == second["mode_T_exhaustive_search_16M"]["frac_of_hbm_roofline"]
    assert roof["tick_ms_p50"] == out["closed_loop_replay"]["solve_ms_p50"] and roof["tick_infeasible_solves"] == 0
    assert roof["config5_us_p50"] == five["solve_us_p50"] and roof["pf_update_us_500"] == pf["particles_500"]["update_us"]
    assert roof["config3_single_us_p50"] == out["config3_single"]["device_resident_us_p50"]


def test_stream_of_batches_is_the_default_and_two_launches_the_option():
    """One rank: the previous batch's argmin and records run inside the next batch's rollout launch
    (acmpc_solve_stream_device, the default); the line names the kernel that carried them, and the step is that kernel and a
    launch boundary - `ms_per_step <= 1.01 x kernel_ms`, the round-3 review's bar.  `--no-stream`: two launches per batch."""
    common = ["--steps", "60", "--warmup", "10", "--no-cpu-baseline", "--no-single-solve"]
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stderr[-2000:]
    out = _last_json(proc.stdout)
    assert out["roofline"]["kernel"] == "rollout_chained_kernel" and "acmpc_solve_stream_device" in out["config"]["pipeline"]
    assert out["value"] > 1e9 and 0.5 < out["roofline"]["frac"] < 1
    assert out["ms_per_step"] <= 1.012 * out["roofline"]["kernel_ms"], (out["ms_per_step"], out["roofline"]["kernel_ms"])
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-stream"] + common, capture_output=True,
                          text=True, timeout=600)
    assert proc.returncode == 0, proc.stderr[-2000:]
    two = _last_json(proc.stdout)
    assert two["roofline"]["kernel"] == "rollout_kernel" and two["config"]["pipeline"] == "none (stream order)"
    assert two["ms_per_step"] > 1.005 * two["roofline"]["kernel_ms"]      # (the finalize and its boundary, behind the kernel)


def test_strong_scaling_mode_is_config_4():
    """`--scaling strong`: Nordschleife, 262 144 candidates x horizon 80 in total, one problem per step, latency reported."""
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--scaling", "strong", "--steps", "20",
                           "--warmup", "3", "--cpu-seconds", "1"], capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stderr[-2000:]
    out = _last_json(proc.stdout)
    assert out["scaling"] == "strong" and out["n_gpus"] == 1
    assert out["config"]["horizon"] == 80 and out["config"]["candidates_per_solve_per_gpu"] == 262144
    assert "configs[3]" in out["config"]["workload"] and out["step_latency"]["p50_us"] > 0
    assert out["roofline"]["algorithmic_bytes_per_launch"] == 262144 * (8 * 79 + 4)


def test_strong_scaling_two_rank_rehearsal():
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo",
           "--scaling", "strong", "--steps", "10", "--warmup", "2"]
    proc = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, (proc.stdout + proc.stderr)[-3000:]
    out = _last_json(proc.stdout)
    assert out["n_gpus"] == 2 and out["scaling"] == "strong"
    assert out["config"]["candidates_per_solve_per_gpu"] == 131072
    assert out["collective"]["world_size"] == 2 and out["step_latency"]["p50_us"] > 0


def test_two_rank_rehearsal_agrees_across_ranks():
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo",
           "--poses", "16", "--steps", "5", "--warmup", "2"]
    proc = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, (proc.stdout + proc.stderr)[-3000:]   # a rank disagreement exits non-zero
    out = _last_json(proc.stdout)
    assert out["n_gpus"] == 2 and out["value"] > 1e6
    assert "one all-reduce(MIN)" in out["config"]["parallelism"]
    # at N > 1 the line also carries the latency half of the metric: config 4 as one solve sharded over the ranks
    sharded = out["sharded_solve"]
    assert "131072 per GPU over 2 GPUs" in sharded["workload"] and 0 < sharded["solve_us_p50"] <= sharded["solve_us_p99"]
    assert out["mpc_solve_latency_ms"]["p50"] == pytest.approx(sharded["solve_us_p50"] * 1e-3)


def test_gpus_flag_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no torchrun on the command line: the parent starts the two ranks as a child
    torch.distributed.run before touching the GPU and relays rank 0's line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--poses",
                           "32", "--steps", "5", "--warmup", "2"], capture_output=True, text=True, timeout=600, env=env)
    assert proc.returncode == 0, (proc.stdout + proc.stderr)[-3000:]
    out = _last_json(proc.stdout)
    assert out["n_gpus"] == 2 and out["collective"]["world_size"] == 2 and out["steps"] == 5
    assert len([line for line in proc.stdout.splitlines() if line.startswith("{")]) == 1


def test_more_gpus_than_devices_is_refused():
    """`--gpus 8` over RCCL on a box with fewer devices exits non-zero with one line instead of running a smaller job."""
    import torch
    if torch.cuda.device_count() >= 8:
        pytest.skip("this box has 8 devices")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1"],
                          capture_output=True, text=True, timeout=300, env=env)
    assert proc.returncode != 0 and "GPU(s) visible" in proc.stderr
    assert not [line for line in proc.stdout.splitlines() if line.startswith("{")]
