"""bench.py's launcher logic on a box WITHOUT a GPU: `python bench.py --gpus 2` with no launcher around it must start the
one-process-per-GPU job itself (python -m torch.distributed.run ... as a child process) and relay the child's exit code —
here the children stop at "bench.py needs a GPU", so the code is non-zero and no JSON line appears.  (The GPU variant,
tests/test_gpu_parity.py::test_bench_gpus_2_without_a_launcher, checks the line itself.)"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_starts_its_own_launcher_and_relays_the_exit_code():
    if os.path.exists("/dev/kfd"):
        pytest.skip("this is the no-GPU variant")
    try:
        import torch  # noqa: F401
    except Exception:
        pytest.skip("torch not installed")
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--size", "1MiB", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert "no launcher around --gpus 2" in p.stderr and "torch.distributed.run" in p.stderr, p.stderr[-2000:]
    assert p.returncode != 0
    assert "needs a GPU" in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]
    # a launcher-provided world that disagrees with --gpus is refused before anything else happens
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=dict(env, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=120)
    assert p.returncode == 2 and "WORLD_SIZE=3" in p.stderr
