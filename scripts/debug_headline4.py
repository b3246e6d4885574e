import numpy as np, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keypointnerf_amd import ops
from keypointnerf_amd.synthetic import to_device
from tests.golden_io import load_case, load_weights
scene, cfg, g = load_case("case_p_v3_headline_tile")
sd = load_weights()
s = to_device(scene, "cuda")
ps = ops.PreparedScene(s["img"], s["cam"], s["feat_geo"], s["feat_tex"], s["sp_data"], s["src_foreground_mask"])
w = ops.PackedWeights(sd)
al = lambda b: (b + 255) // 256 * 256
def dump(grid):
    R = C = grid[3] * grid[4]
    plan = ops.RenderPlan(ps, grid, 64, 64, fine=True)
    ops.render_rays(ps, w, s["cam_tar"], s["bounds"], plan=plan)
    torch.cuda.synchronize()
    ws = plan.ws.cpu().numpy()
    off, o_ = {}, 0
    for name, nb in (("cam_pos", 64), ("dirs", R * 12), ("nearv", R * 4), ("farv", R * 4), ("zc", C * 64 * 4), ("zf", C * 128 * 4), ("rgba", C * 128 * 20), ("rgba_c", C * 64 * 20)):
        off[name] = o_; o_ += al(nb)
    get = lambda name, n: ws[off[name]:off[name] + n * 4].view(np.float32).copy()
    return get("zc", C * 64).reshape(C, 64), get("rgba_c", C * 64 * 5).reshape(C, 64, 5), get("dirs", R * 3).reshape(R, 3)
zf, rf, df = dump((3, 5, 8, 64, 64))
r = 3866
print("full tile  ray 3866 s41..44 sigma", rf[r, 41:45, 0], "z", zf[r, 42].view(np.uint32) if False else zf[r, 42].tobytes().hex())
z1, r1, d1 = dump((211, 485, 1, 1, 1))
print("single ray s41..44 sigma", r1[0, 41:45, 0], "z", z1[0, 42].tobytes().hex(), "dir eq", np.array_equal(d1[0], df[r]))
z2, r2, d2 = dump((3, 485, 8, 64, 1))
print("row        s41..44 sigma", r2[26, 41:45, 0], "z", z2[26, 42].tobytes().hex(), "dir eq", np.array_equal(d2[26], df[r]))
# explicit query at the GPU's own point
cam = torch.tensor(np.array([2.2641287, 0.19999999, -1.9681772], np.float32))
P = (torch.from_numpy(df[r])[None] * torch.from_numpy(zf[r])[:, None]) + cam[None]
q, v = ops.query(ps, w, P[None].cuda(), torch.from_numpy(df[r])[None].expand(64, -1).contiguous()[None].cuda(), mode=1)
print("query      s41..44 sigma", q[0, 41:45, 0].cpu().numpy(), "valid", v.reshape(-1)[41:45].cpu().numpy())
for it in range(3):
    zf2, rf2, _ = dump((3, 5, 8, 64, 64))
    print("repeat full tile", rf2[r, 41:45, 0], "n diff vs first", int((rf2 != rf).any(-1).sum()))
