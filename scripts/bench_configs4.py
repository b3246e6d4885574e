#!/usr/bin/env python
"""BASELINE configs[4] at full size alone (bench.py's secondary.configs4_full), e.g. under A/B environment knobs:
KPN_NO_POOL=1 python scripts/bench_configs4.py [--no-parity]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from keypointnerf_amd import lib as kl, ops  # noqa: E402
from keypointnerf_amd.synthetic import random_hotpath_state_dict  # noqa: E402

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
L = kl.get_library()
r = bench.time_configs4(L, ops, torch, dev, random_hotpath_state_dict(seed=3), ops.get_geo_rows_mode(), with_parity="--no-parity" not in sys.argv)
print(json.dumps(r))
