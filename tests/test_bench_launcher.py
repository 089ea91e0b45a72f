"""bench.py's launcher logic that needs no GPU: `--gpus N` never runs a smaller job under the label N."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(argv, **env_extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True,
                          timeout=300, env=env)


def test_more_gpus_than_devices_is_refused_before_any_rank_starts():
    import torch
    if torch.cuda.device_count() >= 8:
        import pytest
        pytest.skip("this box has 8 devices")
    proc = _run(["--gpus", "8", "--steps", "2"])
    assert proc.returncode != 0 and "GPU(s) visible" in proc.stderr and "{" not in proc.stdout


def test_world_size_must_match_the_gpus_flag():
    proc = _run(["--gpus", "2", "--steps", "2"], WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    assert proc.returncode != 0 and "WORLD_SIZE=1" in proc.stderr
    proc = _run(["--gpus", "1", "--steps", "2"], WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    assert proc.returncode != 0 and "WORLD_SIZE=2" in proc.stderr
