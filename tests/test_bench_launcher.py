"""bench.py's multi-rank contract: `--gpus N` must run N ranks (starting them itself when no launcher did), print
n_gpus = N, and refuse to report for a different world size."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_refuses_a_world_size_that_is_not_gpus():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=3" in (r.stderr + r.stdout)
    assert not any(line.startswith("{") for line in r.stdout.splitlines())            # no JSON line for the wrong GPU count


@pytest.mark.gpu
def test_bench_gpus_2_starts_two_ranks_itself():
    """`python bench.py --gpus 2` with no launcher around it: bench.py re-executes itself under torch.distributed.run with two
    ranks (both on cuda:0 with --dist-backend gloo: the N > 1 flow on a 1-GPU box), gathers the frames to rank 0 and prints
    ONE JSON line with n_gpus = 2."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--steps", "1", "--warmup", "1",
                        "--res", "128", "--samples", "16", "--no-cpu-baseline", "--no-secondary"], env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(line) for line in r.stdout.splitlines() if line.startswith("{")]
    assert len(lines) == 1
    d = lines[0]
    assert d["n_gpus"] == 2 and d["config"]["dist_world_size"] == 2 and d["config"]["dist_backend"] == "gloo"
    assert d["scaling"] == "weak" and d["value"] > 0 and d["config"]["rays_per_step"] == 128 * 128


@pytest.mark.gpu
def test_bench_strong_scaling_splits_one_frame_into_bands():
    """--scaling strong: both ranks render one band of rows of the SAME frame per step; value counts the frame's rays once."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--scaling", "strong", "--steps", "2",
                        "--warmup", "1", "--res", "128", "--samples", "16", "--no-cpu-baseline", "--no-secondary"], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(line) for line in r.stdout.splitlines() if line.startswith("{")]
    assert len(lines) == 1
    d = lines[0]
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["gather_ms_per_round_alone"] > 0
    assert abs(d["value"] - 128 * 128 * d["steps"] / (d["ms_per_step"] * 1e-3 * d["steps"])) < 1e-6 * d["value"]

