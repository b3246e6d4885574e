#!/usr/bin/env python
"""Diagnostic (GPU box): the configs[4] frame's lattice rays (bench.py secondary.configs4_full.parity) in every kernel selection
against the oracle, and the worst ray sample by sample (field values of its 128 points: HIP query vs oracle query)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keypointnerf_amd import ops  # noqa: E402
from keypointnerf_amd.synthetic import make_scene, random_hotpath_state_dict, to_device  # noqa: E402
from oracle import oracle  # noqa: E402

res, views, S = 4096, 10, 128
n, step, off = (int(sys.argv[1]) if len(sys.argv) > 1 else 96), 42, 53
dev = torch.device("cuda", 0)
sd = random_hotpath_state_dict(seed=3)
scene_cpu = make_scene(n_views=views, src_hw=(res, res), tar_hw=(res, res), mask="dense", seed=1, tar_focal_at_512=800.0)
scene = to_device(scene_cpu, dev)
w = ops.PackedWeights(sd, device=dev)
ps = ops.PreparedScene(scene["img"], scene["cam"], scene["feat_geo"], scene["feat_tex"], scene["sp_data"], scene["src_foreground_mask"])
ys, xs = np.meshgrid(np.arange(n) * step + off, np.arange(n) * step + off, indexing="ij")
pix = np.stack([xs.reshape(-1), ys.reshape(-1)], -1).astype(np.int32)
osc, wflat = oracle.OracleScene(scene_cpu), oracle.flat_weights(sd)
ref = oracle.render_rays(osc, wflat, scene_cpu["cam_tar"], scene_cpu["bounds"], pix, S, S, fine=False, stages=True)
worst = None
for rk, fk in (("f16x2", "f16x2"), ("f32", "f32"), ("f16x2", "f32"), ("f32", "f16x2"), ("bf16x3", "f32")):
    plan = ops.RenderPlan(ps, (off, off, step, n, n), S, S, fine=False, rows_kernel=rk, fuse_kernel=fk)
    out = ops.render_rays(ps, w, scene["cam_tar"], scene["bounds"], plan=plan)
    torch.cuda.synchronize()
    tex = out["tex_fg"][0].permute(1, 2, 0).reshape(-1, 3).cpu().numpy()
    al = out["alpha"].reshape(-1).cpu().numpy()
    e = np.abs(tex - ref["tex_fg"]).max(-1)
    print(f"rows {rk:7s} fuse {fk:6s}: rgb err max {e.max():.3e} at ray {int(e.argmax())}, rays > 1e-4: {int((e > 1e-4).sum())}, > 5e-5: {int((e > 5e-5).sum())}; alpha err max {np.abs(al - ref['alpha']).max():.3e}", flush=True)
    if worst is None:
        worst = int(e.argmax())
    del plan
# the worst ray of the default kernels, sample by sample: the oracle's own points through the HIP field query
r = worst
z = ref["z_c"][r]
K, RT = scene_cpu["cam_tar"]["K"].reshape(4, 4).numpy(), scene_cpu["cam_tar"]["RT"].reshape(4, 4).numpy()
dirs, cam_pos, near, far = oracle.make_rays(scene_cpu["cam_tar"], scene_cpu["bounds"], pix[r:r + 1])
pts = (cam_pos[None, :] + dirs[0][None, :] * z[:, None]).astype(np.float32)      # (model.py:1057; the kernels form it the same way)
view = np.repeat(dirs[:1], S, 0).astype(np.float32)
oq, ov = oracle.query(osc, wflat, pts, view, apply_eval_func=True)
oq = np.asarray(oq).reshape(S, 5)
rg = oracle.rgba2out(ref["rgba_c"][r:r + 1], ref["z_c"][r:r + 1])
contrib = rg[3][0]
P = torch.from_numpy(pts)[None].to(dev); Vw = torch.from_numpy(view)[None].to(dev)
np.set_printoptions(precision=3, linewidth=220, suppress=False)
print(f"ray {r} pixel {pix[r]}: oracle query vs oracle render stage rgba: {np.abs(oq - ref['rgba_c'][r]).max():.2e}; contrib top {np.sort(contrib)[-5:]}")
for rm, fm in ((3, 1), (0, 0), (3, 0), (0, 1)):
    ops.set_geo_rows_mode(rm); ops.set_fuse_mode(fm)
    g, val = ops.query(ps, w, P, Vw, mode=1)
    g = g[0].cpu().numpy()
    d = np.abs(g - oq)
    wd = (d[:, 2:5] * contrib[:, None]).sum(0)
    i = int((d[:, 2:5].max(-1) * contrib).argmax())
    print(f"query rows {rm} fuse {fm}: per-point max |d| sigma {d[:,0].max():.2e} sdf {d[:,1].max():.2e} rgb {d[:,2:5].max():.2e}; contrib-weighted rgb error sum {wd}; "
          f"worst weighted sample {i}: contrib {contrib[i]:.3e} d_rgb {d[i,2:5]} sigma {oq[i,0]:.3f} valid mismatch {int((val.reshape(-1).cpu().numpy() != np.asarray(ov).reshape(-1)).sum())}")
ops.set_geo_rows_mode(3); ops.set_fuse_mode(1)
