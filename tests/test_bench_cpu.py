"""bench.py's launcher side, on a machine WITHOUT a GPU (this container): plain `python bench.py --gpus 2` must start its ranks as
child processes itself -- here they fail loudly, because the product has no CPU path, and the parent relays
that: non-zero exit code, no JSON line, the engine's own message on stderr."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _no_gpu_here():
    import torch
    return not torch.cuda.is_available()


@pytest.mark.skipif(not _no_gpu_here(), reason="the point of this test is a host without a GPU")
def test_plain_bench_gpus_2_launches_its_ranks_and_relays_their_failure():
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--scale", "0.01"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0                               # the children's failure is the parent's exit code
    assert r.stdout.strip() == ""                          # no line
    # the ranks were started (the launcher reports a failed local rank) and died of the engine's own loud check, not of a launcher error
    assert "engine_error" in r.stderr and "no CPU path" in r.stderr and "local_rank" in r.stderr, r.stderr[-1500:]
    assert "must be launched with" not in r.stderr        # round 3's refusal is gone
