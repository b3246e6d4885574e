#!/usr/bin/env python
"""Training iteration (bench.py's time_training: forward + backward at 1024 and 4096 rays, with the backward's per-kernel roofline)
for the product library or, with KPN_EXPERIMENT_LIB=path, an experimental build.  MEASUREMENT INFRASTRUCTURE."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from keypointnerf_amd import lib as kl  # noqa: E402

if os.environ.get("KPN_EXPERIMENT_LIB"):
    kl._default = kl.KpnLibrary(os.environ["KPN_EXPERIMENT_LIB"])
from keypointnerf_amd import ops  # noqa: E402
from keypointnerf_amd.synthetic import random_hotpath_state_dict  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda", 0)
sd = random_hotpath_state_dict(seed=3)
for patch, steps in ((32, 10), (64, 5)):
    t = bench.time_training(ops, torch, dev, sd, steps=steps, patch=patch)
    k = t["backward_roofline"]["kernels"]
    print(json.dumps({"rays": patch * patch, "forward_ms": round(t["forward_ms"], 3), "backward_ms": round(t["backward_ms"], 3),
                      **{n: [round(v["ms_per_iteration"], 3), round(v.get("dump_GBps", 0.0))] for n, v in k.items()}}))
