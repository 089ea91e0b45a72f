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
