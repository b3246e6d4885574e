import numpy as np, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keypointnerf_amd import ops
from keypointnerf_amd.synthetic import make_scene, random_hotpath_state_dict, to_device
from oracle import oracle
n_views, mask, src_hw, tar_hw, Sc, Sf = 16, "dense", (48, 48), (10, 10), 8, 8
sd = random_hotpath_state_dict(seed=11 + n_views)
scene = make_scene(n_views=n_views, src_hw=src_hw, tar_hw=tar_hw, mask=mask, seed=20 + n_views)
s = to_device(scene, "cuda")
ps = ops.PreparedScene(s["img"], s["cam"], s["feat_geo"], s["feat_tex"], s["sp_data"], s["src_foreground_mask"])
w = ops.PackedWeights(sd)
H, W = tar_hw
R = C = H * W
plan = ops.RenderPlan(ps, (0, 0, 1, W, H), Sc, Sf, fine=True)
out = ops.render_rays(ps, w, s["cam_tar"], s["bounds"], plan=plan)
torch.cuda.synchronize()
yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
pix = np.stack([xx.reshape(-1), yy.reshape(-1)], -1).astype(np.int32)
o = oracle.render_rays(oracle.OracleScene(scene), oracle.flat_weights(sd), scene["cam_tar"], scene["bounds"], pix, Sc, Sf, stages=True)
ws = plan.ws.cpu().numpy()
al = lambda b: (b + 255) // 256 * 256
off, o_ = {}, 0
for name, nb in (("cam_pos", 64), ("dirs", R * 12), ("nearv", R * 4), ("farv", R * 4), ("zc", C * Sc * 4), ("zf", C * (Sc + Sf) * 4), ("rgba", C * (Sc + Sf) * 20), ("rgba_c", C * Sc * 20)):
    off[name] = o_; o_ += al(nb)
get = lambda name, n: ws[off[name]:off[name] + n * 4].view(np.float32).copy()
zc, zf, rc = get("zc", C * Sc).reshape(C, Sc), get("zf", C * (Sc + Sf)).reshape(C, Sc + Sf), get("rgba_c", C * Sc * 5).reshape(C, Sc, 5)
e = np.abs(out["alpha_fine"].reshape(-1).cpu().numpy() - o["alpha_fine"])
r = int(e.argmax())
print("worst ray", r, "alpha_fine err", e[r], "alpha (coarse) err", np.abs(out["alpha"].reshape(-1).cpu().numpy() - o["alpha"])[r])
print("zc diff", np.abs(zc - o["z_c"]).max(), "rgba_c diff max", np.abs(rc - o["rgba_c"]).max(0))
print("zf gpu ", zf[r]); print("zf orcl", o["z_f"][r]); print("rgba_c sigma gpu", rc[r, :, 0], "orcl", o["rgba_c"][r, :, 0])
