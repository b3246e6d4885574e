#!/usr/bin/env python
"""Render one synthetic frame with an experimental build of the library and compare it with the oracle: per output the largest
error, whether two runs are bit-identical, and where the wrong rays are.   usage: lib_vs_lib.py <lib.so> [V]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from keypointnerf_amd import lib as kl  # noqa: E402

kl._default = kl.KpnLibrary(sys.argv[1])
from keypointnerf_amd import ops  # noqa: E402
from keypointnerf_amd.synthetic import make_scene, random_hotpath_state_dict, to_device  # noqa: E402
from oracle import oracle  # noqa: E402

V = int(sys.argv[2]) if len(sys.argv) > 2 else 3
sd = random_hotpath_state_dict(seed=5)
scene = make_scene(n_views=V, src_hw=(128, 128), tar_hw=(48, 48), mask="ellipsoid", seed=11)
s = to_device(scene, "cuda")
th = tw = 48
yy, xx = np.meshgrid(np.arange(th), np.arange(tw), indexing="ij")
pix = np.stack([xx.reshape(-1), yy.reshape(-1)], -1).astype(np.int32)
ref = oracle.render_rays(oracle.OracleScene(scene), oracle.flat_weights(sd), scene["cam_tar"], scene["bounds"], pix, 32, 32, fine=True)
runs = []
for rep in range(3):
    ps = ops.PreparedScene(s["img"], s["cam"], s["feat_geo"], s["feat_tex"], s["sp_data"], s["src_foreground_mask"])
    out = ops.render_rays(ps, ops.PackedWeights(sd), s["cam_tar"], s["bounds"], grid=(0, 0, 1, tw, th), n_coarse=32, n_fine=32, fine=True)
    runs.append({k: (out[k][0].permute(1, 2, 0).reshape(-1, 3) if k.startswith("tex") else out[k].reshape(-1)).cpu().numpy()
                 for k in ("tex_fg", "alpha", "tex_fg_fine", "alpha_fine")})
for k in runs[0]:
    err = np.abs(runs[0][k] - ref[k]); err = err.max(-1) if err.ndim == 2 else err
    bad = np.nonzero(err > 1e-4)[0]
    print(f"{os.path.basename(sys.argv[1])} V={V} {k}: max err {err.max():.3e}, rays above 1e-4: {len(bad)} of {len(err)}; runs bit-identical: "
          f"{all(np.array_equal(runs[0][k], r[k]) for r in runs[1:])}; first wrong rays {bad[:8].tolist()}")
