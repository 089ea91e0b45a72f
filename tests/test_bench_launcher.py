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


def test_gloo_rehearsal_starts_every_rank_and_returns_their_exit_code():
    """`--gpus 8 --backend gloo` on a box WITHOUT a GPU: the launcher starts eight ranks (torch.distributed.run as a child),
    the ranks receive the arguments and their place in the world, the first to look finds no GPU and says so, and the launcher hands the
    failure back as its own exit code - the spawn path, the argument forwarding and the exit-code propagation of the 8-GPU
    run, with nothing but RCCL itself left untested (on a box with a GPU the same command runs the bench: tests/test_gpu_bench.py)."""
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: the ranks would run the bench (tests/test_gpu_bench.py does that)")
    proc = _run(["--gpus", "8", "--backend", "gloo", "--steps", "2", "--warmup", "1"])
    assert proc.returncode != 0 and "{" not in proc.stdout
    # (the first rank to fail makes torch.distributed.run stop the others: its summary still lists all eight)
    assert "no GPU visible (rank " in proc.stderr and " of 8)" in proc.stderr, proc.stderr[-2000:]
    for rank in range(8):
        assert "(local_rank: %d)" % rank in proc.stderr, proc.stderr[-3000:]
