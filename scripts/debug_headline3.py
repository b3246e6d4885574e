import numpy as np, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keypointnerf_amd import ops
from keypointnerf_amd.synthetic import to_device
from oracle import oracle
from tests.golden_io import load_case, load_weights, pixel_list
scene, cfg, g = load_case("case_p_v3_headline_tile")
sd = load_weights()
s = to_device(scene, "cuda")
ps = ops.PreparedScene(s["img"], s["cam"], s["feat_geo"], s["feat_tex"], s["sp_data"], s["src_foreground_mask"])
w = ops.PackedWeights(sd)
pix, _ = pixel_list(cfg, scene["cam_tar"])
wf = oracle.flat_weights(sd); osc = oracle.OracleScene(scene)
o = oracle.render_rays(osc, wf, scene["cam_tar"], scene["bounds"], pix, 64, 64, fine=True, stages=True)
d, cp, near, far = oracle.make_rays(scene["cam_tar"], scene["bounds"], pix)
step = 8
plan = ops.RenderPlan(ps, (cfg["stride_j"], cfg["stride_i"], step, 64, 64), 64, 64, fine=True)
out = ops.render_rays(ps, w, s["cam_tar"], s["bounds"], plan=plan)
torch.cuda.synchronize()
ws = plan.ws.cpu().numpy()
al = lambda b: (b + 255) // 256 * 256
R = C = 4096
off = {}
o_ = 0
for name, nb in (("cam_pos", 64), ("dirs", R * 12), ("nearv", R * 4), ("farv", R * 4), ("zc", C * 64 * 4), ("zf", C * 128 * 4), ("rgba", C * 128 * 20), ("rgba_c", C * 64 * 20), ("rgba_n", C * 64 * 20), ("zn", C * 64 * 4)):
    off[name] = o_; o_ += al(nb)
get = lambda name, n: ws[off[name]:off[name] + n * 4].view(np.float32)
gd = get("dirs", R * 3).reshape(R, 3); gz = get("zc", C * 64).reshape(C, 64); grc = get("rgba_c", C * 64 * 5).reshape(C, 64, 5)
gn, gf = get("nearv", R), get("farv", R)
print("cam_pos", get("cam_pos", 3), cp)
print("dirs max diff", np.abs(gd - d).max(), "near/far diff", np.abs(gn - near).max(), np.abs(gf - far).max(), "z diff", np.abs(gz - o["z_c"]).max())
ref = o["rgba_c"]
e = np.abs(grc - ref); val = ref[..., 0] > 0
e[~val, 2:] = 0
bad = np.argwhere(e.max(-1) > 1e-4)
print("bad (ray,sample):", bad[:20].tolist(), "count", len(bad))
for r, sidx in bad[:6]:
    print(r, sidx, "gpu", grc[r, sidx], "ref", ref[r, sidx], "z gpu %.8f ref %.8f" % (gz[r, sidx], o["z_c"][r, sidx]), "dir diff", np.abs(gd[r] - d[r]).max())
