"""GPU-side diagnosis: case P (headline tile) per-key error statistics, worst rays, per-sample comparison vs the oracle."""
import numpy as np, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keypointnerf_amd import ops
from keypointnerf_amd.synthetic import to_device
from oracle import oracle
from tests.golden_io import load_case, load_weights, pixel_list, out_as_rays
case = sys.argv[1] if len(sys.argv) > 1 else "case_p_v3_headline_tile"
scene, cfg, g = load_case(case)
sd = load_weights()
s = to_device(scene, "cuda")
ps = ops.PreparedScene(s["img"], s["cam"], s["feat_geo"], s["feat_tex"], s["sp_data"], s["src_foreground_mask"])
w = ops.PackedWeights(sd)
step = 2 ** (cfg["level"] - 1)
out = ops.render_rays(ps, w, s["cam_tar"], s["bounds"], grid=(cfg["stride_j"], cfg["stride_i"], step, 64, 64), n_coarse=cfg["Sc"], n_fine=cfg["Sf"], fine=True)
pix, _ = pixel_list(cfg, scene["cam_tar"])
wf = oracle.flat_weights(sd)
osc = oracle.OracleScene(scene)
o = oracle.render_rays(osc, wf, scene["cam_tar"], scene["bounds"], pix, cfg["Sc"], cfg["Sf"], fine=True, stages=True)
for k in ("tex_fg", "alpha", "tex_fg_fine", "alpha_fine"):
    got = out[k][0].cpu().numpy()
    got = got.transpose(1, 2, 0).reshape(-1, 3) if got.ndim == 3 else got.reshape(-1)
    e = np.abs(got - out_as_rays(g, k)); e = e.max(-1) if e.ndim == 2 else e
    eo = np.abs(got - o[k]); eo = eo.max(-1) if eo.ndim == 2 else eo
    print(k, "vs ref max %.2e  n>1e-4: %d   vs oracle max %.2e n>1e-4: %d" % (e.max(), (e > 1e-4).sum(), eo.max(), (eo > 1e-4).sum()), "worst rays", np.argsort(-e)[:5])
k = "tex_fg"
got = out[k][0].cpu().numpy().transpose(1, 2, 0).reshape(-1, 3)
r = int(np.argmax(np.abs(got - o[k]).max(-1)))
print("worst coarse ray", r, "pix", pix[r])
d, cp, near, far = oracle.make_rays(scene["cam_tar"], scene["bounds"], pix[r:r + 1])
z = o["z_c"][r]
pts = cp[None] + d * z[:, None]
q, v = ops.query(ps, w, torch.from_numpy(pts.astype(np.float32)).cuda()[None], torch.from_numpy(np.repeat(d, len(z), 0)).cuda()[None], mode=1)
q = q[0].cpu().numpy()
ref = o["rgba_c"][r]
err = np.abs(q - ref)
print("per-sample max err per column", err.max(0), "valid gpu", v.reshape(-1).cpu().numpy().astype(int).tolist())
print("oracle sigma>0", (ref[:, 0] > 0).astype(int).tolist())
bad = np.where(err.max(1) > 1e-5)[0]
for i in bad[:10]:
    print(i, "gpu", q[i], "oracle", ref[i])
