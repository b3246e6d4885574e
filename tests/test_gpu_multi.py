"""RCCL path of bench.py on a node with at least two GPUs (-m gpu; skipped on a 1-GPU box): the N > 1 flow the driver runs
(`python -m torch.distributed.run ... bench.py --gpus N`) with backend nccl = RCCL over xGMI, weak (frames) and strong (bands)."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_bench_two_ranks_over_rccl():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "rccl_smoke.py"), "2"], capture_output=True, text=True, timeout=1800)
    assert p.returncode == 0, (p.stdout[-3000:], p.stderr[-3000:])
    assert p.stdout.count('"rccl_ranks": 2') == 2, p.stdout
