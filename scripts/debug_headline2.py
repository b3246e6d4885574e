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
z = o["z_c"]
pts = (cp[None, None] + d[:, None] * z[..., None]).astype(np.float32).reshape(-1, 3)
view = np.repeat(d, 64, 0)
P, V = torch.from_numpy(pts).cuda()[None], torch.from_numpy(view).cuda()[None]
ref = o["rgba_c"].reshape(-1, 5)
res = []
for it in range(3):
    q, v = ops.query(ps, w, P, V, mode=1)
    q = q[0].cpu().numpy(); v = v.reshape(-1).cpu().numpy()
    val = ref[:, 0] > 0
    e = np.abs(q - ref); e[~val, 2:] = 0          # masked rgb: render path writes 0, query the average; ignore
    bad = np.where(e.max(1) > 1e-4)[0]
    print("iter", it, "N", len(ref), "bad points", len(bad), bad[:10], "rays", (bad // 64)[:10], "samples", (bad % 64)[:10], "max err", e.max(0))
    for b in bad[:5]:
        print("   ", b, "gpu", q[b], "ref", ref[b])
    res.append(q)
print("deterministic:", np.array_equal(res[0], res[1]), np.array_equal(res[1], res[2]))
# subsets: only the rays 3800..3900
sel = slice(3800 * 64, 3900 * 64)
q, v = ops.query(ps, w, P[:, sel], V[:, sel], mode=1)
e = np.abs(q[0].cpu().numpy() - ref[sel]); e[~(ref[sel][:, 0] > 0), 2:] = 0
print("subset max err", e.max(0))
