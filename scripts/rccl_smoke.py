#!/usr/bin/env python
"""First contact with RCCL on a multi-GPU node: runs `bench.py --gpus N --dist-backend nccl` (N = 2 by default, or the
argument) for two short steps in both scaling modes with NCCL_DEBUG=VERSION, checks that rank 0's JSON line reports N GPUs and
N RCCL ranks, and prints the lines.  On a 1-GPU box it says so and exits 0 (tests/test_gpu_multi.py skips the same way)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(n, scaling):
    env = dict(os.environ, NCCL_DEBUG="VERSION", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--dist-backend", "nccl", "--steps", "2", "--warmup", "1",
           "--no-secondary", "--no-cpu-baseline", "--scaling", scaling]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    version = [l for l in (p.stdout + p.stderr).splitlines() if "RCCL version" in l or "NCCL version" in l]
    assert p.returncode == 0 and lines, (p.returncode, p.stdout[-2000:], p.stderr[-2000:])
    d = json.loads(lines[-1])
    assert d["n_gpus"] == n and d["config"]["rccl_ranks"] == n and d["config"]["dist_backend"] == "nccl", d["config"]
    assert d["scaling"] == scaling
    return d, version[:1]


def main():
    import torch
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    if torch.cuda.device_count() < n:
        print(f"rccl_smoke: {torch.cuda.device_count()} GPU(s) visible, {n} needed: nothing to run")
        return 0
    for scaling in ("weak", "strong"):
        d, version = run(n, scaling)
        print(json.dumps({"scaling": scaling, "n_gpus": d["n_gpus"], "rays_per_sec": d["value"], "ms_per_step": d["ms_per_step"],
                          "rccl_ranks": d["config"]["rccl_ranks"], "dist_library": d["config"]["dist_library"],
                          "gather_ms_per_round_alone": d["config"]["gather_ms_per_round_alone"], "nccl_debug_version": version}))
    return 0


if __name__ == "__main__":
    sys.exit(main())
