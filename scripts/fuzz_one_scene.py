#!/usr/bin/env python
"""One scene of the randomised parity sweep (scripts/fuzz_parity.py) in every kernel selection, ray by ray: which rays are above
the bar, their error per output and the oracle's envelope.  usage: fuzz_one_scene.py <scene index> [sweep seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from keypointnerf_amd import ops  # noqa: E402
from keypointnerf_amd.synthetic import make_scene, random_hotpath_state_dict, to_device  # noqa: E402
from oracle import oracle  # noqa: E402
from tests.test_gpu_fuzz import fuzz_scene  # noqa: E402

idx, seed = int(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 2024
rng = np.random.default_rng(seed)
for _ in range(idx + 1):
    cfg = fuzz_scene(rng)
print(cfg)
sd = random_hotpath_state_dict(seed=cfg["seed"], density_bias=cfg["bias"])
scene = make_scene(n_views=cfg["V"], src_hw=cfg["src"], tar_hw=cfg["tar"], mask=cfg["mask"], seed=cfg["seed"] + 1, tar_angle=cfg["angle"],
                   tar_focal_at_512=cfg["focal"])
s = to_device(scene, "cuda")
th, tw = cfg["tar"]
yy, xx = np.meshgrid(np.arange(th), np.arange(tw), indexing="ij")
pix = np.stack([xx.reshape(-1), yy.reshape(-1)], -1).astype(np.int32)
osc, wf = oracle.OracleScene(scene), oracle.flat_weights(sd)
ref = oracle.render_rays(osc, wf, scene["cam_tar"], scene["bounds"], pix, cfg["Sc"], cfg["Sf"], fine=cfg["fine"])
env = oracle.render_envelope(osc, wf, scene["cam_tar"], scene["bounds"], pix, cfg["Sc"], cfg["Sf"], fine=cfg["fine"], ref=ref, trials=64)
keys = ("tex_fg", "alpha") + (("tex_fg_fine", "alpha_fine") if cfg["fine"] else ())
for rm, fm in ((3, 1), (0, 0), (2, 0), (0, 1), (0, 0)):
    ops.set_geo_rows_mode(rm); ops.set_fuse_mode(fm)
    for chunk in (cfg["chunk"], 0):
        ps = ops.PreparedScene(s["img"], s["cam"], s["feat_geo"], s["feat_tex"], s["sp_data"], s["src_foreground_mask"])
        out = ops.render_rays(ps, ops.PackedWeights(sd), s["cam_tar"], s["bounds"], grid=(0, 0, 1, tw, th), n_coarse=cfg["Sc"], n_fine=cfg["Sf"],
                              fine=cfg["fine"], chunk_rays=chunk)
        got = {k: (out[k][0].permute(1, 2, 0).reshape(-1, 3) if k.startswith("tex") else out[k].reshape(-1)).cpu().numpy() for k in keys}
        err = {k: (np.abs(got[k] - ref[k]).max(-1) if got[k].ndim == 2 else np.abs(got[k] - ref[k])) for k in keys}
        bad = np.nonzero(np.any([err[k] > 1e-4 for k in keys], axis=0))[0]
        print(f"rows mode {rm} fuse mode {fm} chunk {chunk}: max", {k: float(err[k].max()) for k in keys}, "rays above 1e-4:",
              [(int(r), {k: float(err[k][r]) for k in keys}, {k: float(env[k][r]) for k in keys}) for r in bad[:4]])
