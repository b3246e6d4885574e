#!/usr/bin/env python
"""Diagnostic: for one scene of the parity sweep (scripts/fuzz_parity.py's sequence: seed, index) print, for every ray the gate
widens, the stage excesses of the conditional check and — for the field stages — the worst sample: kernel value, oracle value at
the kernel's depth, the oracle's sensitivity there.  MEASUREMENT INFRASTRUCTURE (imports the oracle).
usage: diag_conditional.py SEED INDEX"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from keypointnerf_amd import ops  # noqa: E402
from keypointnerf_amd.synthetic import make_scene, random_hotpath_state_dict, to_device  # noqa: E402
from oracle import oracle  # noqa: E402
from tests import parity_gate  # noqa: E402
from tests.test_gpu_fuzz import fuzz_scene  # noqa: E402

seed, index = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
for _ in range(index + 1):
    cfg = fuzz_scene(rng)
print(cfg)
sd = random_hotpath_state_dict(seed=cfg["seed"], density_bias=cfg["bias"])
scene = make_scene(n_views=cfg["V"], src_hw=cfg["src"], tar_hw=cfg["tar"], mask=cfg["mask"], seed=cfg["seed"] + 1, tar_angle=cfg["angle"], tar_focal_at_512=cfg["focal"])
s = to_device(scene, "cuda")
ps = ops.PreparedScene(s["img"], s["cam"], s["feat_geo"], s["feat_tex"], s["sp_data"], s["src_foreground_mask"])
w = ops.PackedWeights(sd)
th, tw = cfg["tar"]
Sc, Sf, fine = cfg["Sc"], cfg["Sf"], cfg["fine"]
out = ops.render_rays(ps, w, s["cam_tar"], s["bounds"], grid=(0, 0, 1, tw, th), n_coarse=Sc, n_fine=Sf, fine=fine, chunk_rays=cfg["chunk"])
yy, xx = np.meshgrid(np.arange(th), np.arange(tw), indexing="ij")
pix = np.stack([xx.reshape(-1), yy.reshape(-1)], -1).astype(np.int32)
osc, wf = oracle.OracleScene(scene), oracle.flat_weights(sd)
ref = oracle.render_rays(osc, wf, scene["cam_tar"], scene["bounds"], pix, Sc, Sf, fine=fine)
keys = ("tex_fg", "alpha") + (("tex_fg_fine", "alpha_fine") if fine else ())
got = {k: (out[k][0].permute(1, 2, 0).reshape(-1, 3) if k.startswith("tex") else out[k].reshape(-1)).cpu().numpy() for k in keys}
rep = parity_gate.check_rays(got, ref, parity_gate.oracle_envelope(oracle, osc, wf, scene["cam_tar"], scene["bounds"], pix, Sc, Sf, fine=fine), keys=keys,
                             max_widened_fraction=1.0, what="diag")
one = parity_gate.product_render_one(ops, ps, w, s["cam_tar"], s["bounds"], Sc, Sf, fine)
for row in rep["widened"]:
    x, y = (int(v) for v in pix[row["ray"]])
    o1, st = one(x, y)
    c = parity_gate.conditional_check(oracle, osc, wf, scene["cam_tar"], scene["bounds"], np.array([[x, y]], np.int32), o1, st, Sc, Sf, fine=fine)[0]
    print("ray", row["ray"], "err", row["err"], "\n  stages", c["stages"])
    dirs, cam_pos, near, far = oracle.make_rays(scene["cam_tar"], scene["bounds"], np.array([[x, y]], np.int32))
    for name, zk, rk in (("coarse", st["z_coarse"], st["rgba_coarse"]),) + ((("fine", st["z_fine"], st["rgba_fine"]),) if fine else ()):
        z = np.asarray(zk, np.float32).reshape(1, -1)
        rk = np.asarray(rk, np.float32).reshape(1, -1, 5)
        P = (cam_pos[None, None] + dirs[:, None] * z[..., None]).astype(np.float32)
        view = np.repeat(dirs[:, None], z.shape[1], 1).reshape(-1, 3)
        q = lambda pts: oracle.query(osc, wf, np.ascontiguousarray(pts.reshape(-1, 3), np.float32), view, apply_eval_func=True)
        o, valid = q(P)
        d = np.abs(rk[0] - o)
        d[:, 2:] *= (rk[0, :, :1] > 0)
        i = int(np.argmax((d - 2e-5 * (1 + np.abs(o) * (np.arange(5) < 2))).max(-1)))
        spread = np.zeros(5)
        for sg in ([1, 1, 1], [-1, -1, -1], [1, -1, 1], [-1, 1, -1], [1, 1, -1], [-1, -1, 1]):
            o2, v2 = q(P * (1 + 2.4e-7 * np.array(sg, np.float32))[None, None])
            spread = np.maximum(spread, np.abs(o2[i] - o[i]))
        print(f"  {name}: worst sample {i} of {z.shape[1]} z={z[0, i]:.6f} valid={bool(valid[i])}\n    kernel {rk[0, i]}\n    oracle {o[i]}\n    |diff| {d[i]}\n    spread {spread}")
        if i > 0:
            print(f"    neighbours: kernel sigma {rk[0, max(0, i - 2):i + 3, 0]}, oracle sigma {o[max(0, i - 2):i + 3, 0]}")

    if fine:   # the resampling stage: the kernel's new depths against the oracle's from the kernel's coarse weights
        zc, rc = np.asarray(st["z_coarse"], np.float32).reshape(1, -1), np.asarray(st["rgba_coarse"], np.float32).reshape(1, -1, 5)
        _, _, _, contrib, _ = oracle.rgba2out(rc, zc)
        zmid = 0.5 * (zc[:, 1:] + zc[:, :-1])
        cin = np.ascontiguousarray(contrib[:, 1:Sc - 1])
        zn = oracle.importance_sample(cin, zmid, Sf)
        zf_o = np.sort(np.concatenate([zc, zn], -1), -1)[0]
        zf_k = np.asarray(st["z_fine"], np.float32).reshape(-1)
        w = cin[0].astype(np.float64) + 1e-5
        cdf = np.concatenate([[0.0], np.cumsum(w / w.sum())])
        u = np.linspace(0.0, 1.0, Sf)
        bad = np.nonzero(np.abs(zf_k - zf_o) > 5e-6)[0]
        print("  resample: differing merged samples", bad, "kernel", zf_k[bad], "oracle", zf_o[bad])
        print("    bin widths (cdf):", np.diff(cdf)[:8], "... min", np.diff(cdf).min(), "sum of weights", w.sum())
        j = np.searchsorted(cdf, u, side="right")
        print("    distance of every u to its nearest cdf entry:", np.abs(u[:, None] - cdf[None]).min(-1).min(), "; den of the bins used:", (cdf[np.minimum(j, len(cdf) - 1)] - cdf[np.maximum(j - 1, 0)]))
