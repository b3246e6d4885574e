"""Eager-PyTorch restatement of the ray-march path — TEST / MEASUREMENT INFRASTRUCTURE ONLY.

What it is for: the reference itself cannot travel to the GPU box (/root/reference exists in the build container only), so
"the unfused PyTorch-ROCm path on the MI355X"  cannot be measured with the reference's own files.  This
module restates the same arithmetic with the same ATen operators the reference uses (grid_sample, softplus, elu, cumprod,
searchsorted, sort, ...) so that scripts/bench_torch_eager.py can time an unfused eager implementation on the same GPU,
and tests/test_torch_eager.py pins it to the reference's recorded outputs (tests/golden) — it is a second, independent
oracle, written from SURVEY.md §8 / Appendix A, not a copy of the reference's code: parameters come as the flat effective
vector (weights.flatten_plain), views are a leading batch axis, one function per stage.

Nothing under keypointnerf_amd/ imports this module.

Reference lines restated: src/model.py:690-843 (query, query_color), :942-1108 (batch_render_pifu_nerf, eval branch),
:1110-1176 (importance_sample, rgba2out), :1178-1237 (ray_bbox_intersection), :1267-1302 (IBRRenderingHead.forward);
src/spatial.py:63-118; src/utils.py:74-95, 500-517, 577-587, 612-647, 691-748.
"""
import math

import torch
import torch.nn.functional as F

from keypointnerf_amd.synthetic import HOTPATH_LAYERS


def unpack_plain(plain):
    """flat effective-parameter vector -> {name: (W (out,in), b (out,))}, 'ani_al' -> 0-dim tensor."""
    out, off = {}, 0
    for name, _, (o, i), _ in HOTPATH_LAYERS:
        W = plain[off:off + o * i].view(o, i); off += o * i
        b = plain[off:off + o]; off += o
        out[name] = (W, b)
    out["ani_al"] = plain[off]
    return out


def softplus100(x):
    return F.softplus(x, beta=100.0, threshold=20.0)


def sample_maps(maps, xy):
    """maps (V,C,h,w), xy (V,N,2) in [-1,1] -> (V,N,C): bilinear, border padding, align_corners (src/utils.py:74-89)."""
    return F.grid_sample(maps, xy[:, :, None, :], mode="bilinear", padding_mode="border", align_corners=True)[..., 0].transpose(1, 2)


def keypoint_encoding(pts_v, kpt3d, extrin, sigma):
    """rel_z_decay (src/spatial.py:63-118): pts_v (V,N,3) world points per view -> (V,N,168)."""
    R, t = extrin[:, :3, :3], extrin[:, :3, 3]
    cam_pts = pts_v @ R.transpose(1, 2) + t[:, None]                       # (V,N,3)
    cam_kpt = kpt3d[None] @ R.transpose(1, 2) + t[:, None]                 # (V,K,3)
    dz = cam_pts[:, :, None, 2] - cam_kpt[:, None, :, 2]                   # (V,N,K)
    d2 = ((cam_pts[:, :, None] - cam_kpt[:, None]) ** 2).sum(-1)
    w = torch.exp(-d2 / (2.0 * sigma ** 2))
    freqs = torch.tensor([math.pi * 2 ** k for k in range(3)], dtype=torch.float32, device=pts_v.device).float()
    ang = dz[:, :, None, :] * freqs[None, None, :, None]                   # (V,N,3,K)
    pe = torch.cat([torch.sin(ang), torch.cos(ang)], -1).reshape(*dz.shape[:2], -1)   # per level [sin K | cos K]
    enc = torch.cat([dz, pe], -1).view(*dz.shape[:2], 7, -1) * w[:, :, None]
    return enc.reshape(*dz.shape[:2], -1)


def field(P, pts, view, scene, sigma=0.1, disable_fg_mask=False):
    """KeypointNeRF.query + eval_func (src/model.py:690-843, 978-997), eval mode, batch 1.
    pts, view (N,3) -> rgba (N,5) = [density, sdf, r, g, b]."""
    cam = scene["cam"]
    KRT = cam["KRT"]
    V, N = KRT.shape[0], pts.shape[0]
    v = pts[None].expand(V, -1, -1)
    vh = v @ KRT[:, :3, :3].transpose(1, 2) + KRT[:, :3, 3][:, None]
    z = vh[..., 2:3]
    xy = vh[..., :2] / z
    xy = torch.stack([2.0 * (xy[..., 0] / (cam["width"] - 1.0)) - 1.0, 2.0 * (xy[..., 1] / (cam["height"] - 1.0)) - 1.0], -1)
    zn = 2.0 * (z - cam["znear"]) / (cam["zfar"] - cam["znear"]) - 1.0
    inside = ((xy >= -1.01) & (xy <= 1.01)).all(-1, keepdim=True) & (zn >= -1.0)          # (V,N,1)
    mask = inside.all(0, keepdim=True)
    if not disable_fg_mask:
        fg = sample_maps(scene["src_foreground_mask"].reshape(V, 1, *scene["img"].shape[-2:]).float(), xy)
        mask = mask & (fg > 0.1).all(0, keepdim=True)
    mask = (inside & mask).float()                                                          # (V,N,1), the same for every view
    xyz01 = 0.5 * torch.cat([xy, zn], -1) + 0.5
    border = torch.minimum(xyz01, 1.0 - xyz01)
    pw = torch.sigmoid(5.0 * (border / 0.1 - 1.0)).prod(-1, keepdim=True) * mask
    pw = pw / (pw.sum(0, keepdim=True) + 1e-6)
    g0, g1 = sample_maps(scene["feat_geo"][0], xy), sample_maps(scene["feat_geo"][1], xy)
    enc = keypoint_encoding(v, scene["sp_data"]["kpt3d"].reshape(-1, 3), scene["sp_data"]["extrin"], sigma)
    lin = lambda name, x: F.linear(x, *P[name])
    h = softplus100(lin("g1_0", torch.cat([enc, g0], -1)))
    h = softplus100(lin("g1_1", h))
    h = softplus100(lin("g1_2", torch.cat([h, g1], -1)))
    xv = lin("g1_3", h)                                                                      # (V,N,64)
    mean = (pw * xv).sum(0)
    var = (pw * (xv - mean[None]) ** 2).sum(0)
    fused = torch.cat([mean, var], -1)                                                       # (N,128)
    g = lin("g2_2", softplus100(lin("g2_1", softplus100(lin("g2_0", fused)))))               # (N,2) = [sdf, rad]
    valid = (mask.sum(0) > 0).float()                                                        # (N,1)
    # colour head
    src_rgb = sample_maps(scene["img"], xy)
    tex = sample_maps(scene["feat_tex"], xy)
    lat = lin("cmp", fused)[None].expand(V, -1, -1)
    feat = torch.cat([src_rgb, tex, lat], -1)                                                # (V,N,35)
    centre = torch.inverse(KRT)[:, :3, 3]                                                    # (V,3)
    cam_rays = F.normalize(v - centre[:, None], dim=-1)
    diff = view[None] - cam_rays
    rd = torch.cat([diff / diff.norm(dim=-1, keepdim=True).clamp(min=1e-6), (cam_rays * view[None]).sum(-1, keepdim=True)], -1)
    feat = feat + F.elu(lin("re_1", F.elu(lin("re_0", rd))))
    e = torch.exp(P["ani_al"].abs() * (rd[..., 3:4] - 1.0))
    bw = (e - e.min(0, keepdim=True)[0]) * mask
    bw = bw / (bw.sum(0, keepdim=True) + 1e-8)
    m = (feat * bw).sum(0, keepdim=True)
    s2 = (bw * (feat - m) ** 2).sum(0, keepdim=True)
    x = F.elu(lin("bl_1", F.elu(lin("bl_0", torch.cat([m.expand(V, -1, -1), s2.expand(V, -1, -1), feat], -1)))))
    rv = F.elu(lin("v1_1", F.elu(lin("v1_0", x * bw))))
    x = x + rv[..., :32]
    vis = torch.sigmoid(lin("v2_1", F.elu(lin("v2_0", x * torch.sigmoid(rv[..., 32:33]) * mask)))) * mask
    logit = lin("o_2", F.elu(lin("o_1", F.elu(lin("o_0", torch.cat([x, vis, rd], -1))))))
    logit = logit.masked_fill(mask == 0, -1e9)
    rgb = (src_rgb * torch.softmax(logit, 0)).sum(0)
    sdf = valid * g[:, :1] + (1.0 - valid) * (0.1 / cam["nml_scale"])
    return torch.cat([valid * F.relu(g[:, 1:2]), sdf, rgb], -1)


def composite(rgba, z):
    """rgba2out (src/model.py:1150-1176): (R,S,5), (R,S) -> colour (R,3), depth, alpha, contrib (R,S), sdf."""
    dist = torch.cat([z[:, 1:] - z[:, :-1], torch.full_like(z[:, :1], 1e10)], -1)
    a = 1.0 - torch.exp(-rgba[..., 0] * dist)
    c = a * torch.cumprod(torch.cat([torch.ones_like(a[:, :1]), 1.0 - a[:, :-1]], -1), -1)
    alpha = c.sum(-1)
    return (rgba[..., 2:] * c[..., None]).sum(1), (z * c).sum(-1) / (alpha + 1e-8), alpha, c, (rgba[..., 1] * c).sum(-1) / (alpha + 1e-8)


def resample(contrib, z_mid, n):
    """importance_sample with uniform u (src/model.py:1110-1148): contrib (R,D-2), z_mid (R,D-1) -> (R,n)."""
    pdf = contrib + 1e-5
    pdf = pdf / pdf.sum(-1, keepdim=True)
    cdf = torch.cat([torch.zeros_like(pdf[:, :1]), torch.cumsum(pdf, -1)], -1)
    u = torch.linspace(0.0, 1.0, n, device=z_mid.device)[None].expand(cdf.shape[0], -1).contiguous()
    idx = torch.searchsorted(cdf, u, right=True)
    lo, hi = (idx - 1).clamp(min=0), idx.clamp(max=cdf.shape[-1] - 1)
    c0, c1, z0, z1 = cdf.gather(1, lo), cdf.gather(1, hi), z_mid.gather(1, lo), z_mid.gather(1, hi)
    den = c1 - c0
    den = torch.where(den < 1e-5, torch.ones_like(den), den)
    return z0 + (u - c0) / den * (z1 - z0)


def box_near_far(bounds, origin, dirs):
    """ray_bbox_intersection (src/model.py:1178-1237): exactly-two-faces rule; misses get near = far = 1, hit = False."""
    b = bounds.reshape(2, 3) + torch.tensor([-0.01, 0.01], device=bounds.device)[:, None]
    d = torch.where(dirs.abs() < 1e-5, torch.full_like(dirs, 1e-5), dirs)
    t = (b[:, None, :] - origin[None, None, :]) / d[None]                                   # (2,R,3)
    t = t.permute(1, 0, 2).reshape(-1, 6)                                                    # (R,6): min faces xyz, max faces xyz
    hitp = origin[None, None] + t[..., None] * d[:, None]                                    # (R,6,3)
    ok = ((hitp >= b[0] - 1e-6) & (hitp <= b[1] + 1e-6)).all(-1)
    two = ok.sum(-1) == 2
    dist = (hitp - origin).norm(dim=-1) / d.norm(dim=-1, keepdim=True)
    near = torch.where(two, torch.where(ok, dist, torch.full_like(dist, 1e30)).min(-1)[0], torch.ones_like(dist[:, 0]))
    far = torch.where(two, torch.where(ok, dist, torch.full_like(dist, -1e30)).max(-1)[0], torch.ones_like(dist[:, 0]))
    return near, far, two


def render_rays(P, scene, pix, n_coarse=64, n_fine=64, fine=True, sigma=0.1, disable_fg_mask=False):
    """batch_render_pifu_nerf, eval branch with uniform=True (src/model.py:1019-1108), for integer pixels pix (R,2) = (x,y).
    Returns the out dict with (R,3) / (R,) tensors."""
    tar = scene["cam_tar"]
    K, RT = tar["K"][0], tar["RT"][0]
    R = pix.shape[0]
    g = torch.cat([pix.float(), torch.ones(R, 1, device=pix.device)], -1)
    inv_K = torch.inverse(K[:3, :3]).t()
    cam = g @ inv_K
    near0, far0 = (tar["znear"] * cam).norm(dim=-1), (tar["zfar"] * cam).norm(dim=-1)
    dirs = F.normalize(cam @ RT[:3, :3], dim=-1)
    origin = -(RT[:3, 3] @ RT[:3, :3])
    z1, z2, hit = box_near_far(scene["bounds"], origin, dirs)
    near = torch.where(hit & (z1 > near0), z1, near0)
    far = torch.where(hit & (z2 < far0), z2, far0)
    t = torch.linspace(0.0, 1.0, n_coarse, device=pix.device)
    z = near[:, None] + (far - near)[:, None] * t[None]

    def march(zs):
        S = zs.shape[1]
        pts = origin[None, None] + dirs[:, None] * zs[..., None]
        rgba = field(P, pts.reshape(-1, 3), dirs[:, None].expand(-1, S, -1).reshape(-1, 3), scene, sigma, disable_fg_mask)
        return composite(rgba.view(R, S, 5), zs)

    col, dep, alp, contrib, _ = march(z)
    out = {"tex_fg": col, "depth": dep, "alpha": alp}
    if fine:
        z_mid = 0.5 * (z[:, 1:] + z[:, :-1])
        z_all = torch.sort(torch.cat([z, resample(contrib[:, 1:-1], z_mid, n_fine)], -1), -1)[0]
        col, dep, alp, _, sdf = march(z_all)
        out.update({"tex_fg_fine": col, "depth_fine": dep, "alpha_fine": alp, "sdf": sdf})
    return out
