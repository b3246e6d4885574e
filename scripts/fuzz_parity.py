#!/usr/bin/env python
"""Randomised parity sweep on the GPU box at any size: the scenes and the gate of tests/test_gpu_fuzz.py (HIP render through the
C ABI vs the C oracle; every ray within 1e-4 unless the oracle's own conditioning probe explains it, tests/parity_gate.py).
Writes gpurun_out/fuzz_parity.json: a summary plus the rays that needed the widened bar, each with its error and the oracle's
envelope.  MEASUREMENT / TEST INFRASTRUCTURE (imports the oracle).  Usage: fuzz_parity.py [n_scenes] [seed] [--fp32]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from keypointnerf_amd import ops  # noqa: E402
from tests.test_gpu_fuzz import fuzz_scene, run_scene  # noqa: E402


def main():
    fp32 = "--fp32" in sys.argv          # the fp32-MFMA kernels in both field kernels (kernel selection has no environment variable)
    argv = [a for a in sys.argv[1:] if not a.startswith("--")]
    if fp32:
        ops.set_geo_rows_mode(0); ops.set_fuse_mode(0)
    n_scenes = int(argv[0]) if len(argv) > 0 else 40
    rng = np.random.default_rng(int(argv[1]) if len(argv) > 1 else 2024)
    t0 = time.time()
    rays = above = 0
    widened, worst = [], {}
    for i in range(n_scenes):
        cfg = fuzz_scene(rng)
        rep = run_scene(ops, cfg)          # raises on a ray the gate does not accept
        rays += rep["rays"]
        above += rep["above_bar"]
        for k, v in rep["max_err"].items():
            worst[k] = max(worst.get(k, 0.0), v)
        for w in rep["widened"]:
            widened.append(dict(scene=i, **{k: cfg[k] for k in ("V", "Sc", "Sf", "mask", "fine")}, **w))
        print(f"scene {i}: {cfg['V']} views, {cfg['tar']} rays, {cfg['Sc']}+{cfg['Sf'] if cfg['fine'] else 0} {cfg['mask']}: max err "
              + ", ".join(f"{k} {v:.1e}" for k, v in rep["max_err"].items()) + f"; above 1e-4: {rep['above_bar']}", flush=True)
    res = {"scenes": n_scenes, "rays": rays, "rows_mode": ops.get_geo_rows_mode(), "rays_above_1e-4": above, "rays_explained_by_the_oracle_envelope": len(widened),
           "widened_rays_conditional_ok": sum(1 for w in widened if w.get("conditional_ok")),
           "fuse_mode": ops.get_fuse_mode(), "rays_unexplained": 0, "max_error": worst, "seconds": time.time() - t0,
           "what": "keypointnerf_amd HIP render vs C oracle, random scenes / weights / cameras / sample counts; gate = tests/parity_gate.py"}
    print(json.dumps(res))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump({"summary": res, "widened_rays": widened}, open(os.path.join(ROOT, "gpurun_out", "fuzz_parity.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
