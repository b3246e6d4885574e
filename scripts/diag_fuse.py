#!/usr/bin/env python
"""Diagnosis of k_fuse_color_h on the GPU: golden query in fuse mode 1 vs 0 (rows mode 0), per output column. [lib.so]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from keypointnerf_amd import lib as kl
if len(sys.argv) > 1:
    kl._default = kl.KpnLibrary(sys.argv[1])
from keypointnerf_amd import ops
from keypointnerf_amd.synthetic import to_device
from tests.golden_io import CASES, load_case, load_weights
scene, cfg, g = load_case(CASES[0])
s = to_device(scene, "cuda")
ps = ops.PreparedScene(s["img"], s["cam"], s["feat_geo"], s["feat_tex"], s["sp_data"], s["src_foreground_mask"])
sd = load_weights()
w = ops.PackedWeights(sd)
pts, view = torch.from_numpy(g["query.0.pts"]).cuda(), torch.from_numpy(g["query.0.view"]).cuda()
ops.set_geo_rows_mode(0)
res = {}
for fm in (0, 1):
    ops.set_fuse_mode(fm)
    o, v = ops.query(ps, w, pts, view, mode=0)
    res[fm] = (o[0].cpu().numpy(), v.cpu().numpy().reshape(-1).astype(bool))
o0, v0 = res[0]; o1, v1 = res[1]
d = np.abs(o1 - o0)[v0]
print(os.path.basename(sys.argv[1]) if len(sys.argv) > 1 else "product", "valid", int(v0.sum()), "nan points", int(np.isnan(o1[v0]).any(-1).sum()),
      "max |f16 - fp32| per column", [float(np.nanmax(d[:, c])) for c in range(5)], "points off by > 1e-4:", int((np.nan_to_num(d, nan=1.0)[:, 2:].max(-1) > 1e-4).sum()))
