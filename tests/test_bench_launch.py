"""`python bench.py --gpus N` must start N ranks by itself (the driver calls it exactly like the
N = 1 run) and must never print a line whose n_gpus differs from what was asked for."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(300)
def test_gpus_2_self_launches_two_ranks():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present: covered by the -m gpu run of bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, env=env, timeout=280)
    # no GPU here: both ranks must fail loudly (no CPU fallback), after having been launched
    assert p.returncode != 0
    assert "--nproc-per-node=2" in p.stderr
    assert "No HIP GPUs" in p.stderr or "no HIP device" in p.stderr or "NODEVICE" in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")], "a result line was printed without a GPU"


def test_world_size_mismatch_is_refused():
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True, text=True, env=env, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=4" in (p.stderr + p.stdout)
