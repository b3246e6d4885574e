"""Seeded synthetic scenes for the ray-march path.

ZJU-MoCap frames and the pretrained checkpoint are not available offline, so every parity test,
``smoke()`` and ``bench.py`` run on scenes from this generator: V source cameras on a ring of
radius 3 m looking at the origin, a target camera between two sources, 24 keypoints in a
0.6 x 1.6 x 0.4 m box, bounds = keypoint AABB +-0.1 m, foreground masks = projected ellipsoid
("ellipsoid") or all ones ("dense"), uniform-random source images and normal-random feature maps of
exactly the shapes the reference's encoders hand over (reference src/model.py:653-680):
``feat_geo = [(V,64,H/8,W/8), (V,8,H/2,W/2)]``, ``feat_tex = (V,8,H/4,W/4)`` in NCHW.

The dictionaries mirror the ones ``decode_batch`` builds in the reference (src/model.py:336-388).
Nothing here depends on the reference or on the oracle.
"""
import math

import numpy as np
import torch

N_KPT = 24


def _look_at(pos, target=(0.0, 0.0, 0.0), up=(0.0, -1.0, 0.0)):
    """World->camera rotation/translation, OpenCV convention (x right, y down, z forward)."""
    pos = np.asarray(pos, np.float64)
    fwd = np.asarray(target, np.float64) - pos
    fwd /= np.linalg.norm(fwd)
    upv = np.asarray(up, np.float64)
    right = np.cross(fwd, -upv)
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    R = np.stack([right, down, fwd], 0)  # rows = camera axes in world coords
    t = -R @ pos
    return R, t


def _ring_camera(angle_deg, radius=3.0, height=0.0):
    a = math.radians(angle_deg)
    pos = (radius * math.sin(a), height, -radius * math.cos(a))
    return _look_at(pos)


def _intrinsics(H, W, focal_at_512=600.0):
    f = focal_at_512 * (W / 512.0)
    K = np.eye(4, dtype=np.float64)
    K[0, 0] = f
    K[1, 1] = f
    K[0, 2] = W / 2.0
    K[1, 2] = H / 2.0
    return K


def make_scene(n_views=3, src_hw=(512, 512), tar_hw=(512, 512), mask="ellipsoid", seed=1,
               tar_near_far=(2.0, 8.0), tar_angle=None, device="cpu", dtype=torch.float32, tar_focal_at_512=600.0):
    """Returns a dict with the tensors/dicts the reference's hot path consumes.

    keys: img (V,3,H,W) in [0,1]; feat_geo [list of 2]; feat_tex; cam (source cameras dict);
    cam_tar (target camera dict); sp_data {kpt3d (1,24,3), extrin (V,4,4)}; bounds (1,2,3);
    src_foreground_mask (1,V,1,H,W) bool; n_views.

    tar_focal_at_512: focal length of the TARGET camera in pixels at a 512-wide image.  600 (the sources' focal) leaves the
    1.8 m subject 70 % of the frame height; 800 frames it like the reference's orbit does (render_video_zju: 1337.6 px at
    5 m = 94 % of the height, src/model.py:178-187), which is what brings the fraction of valid sample points to the
    30-40 % SURVEY.md §8(d) specifies for the ellipsoid scene.
    """
    H, W = src_hw
    Ht, Wt = tar_hw
    g = torch.Generator().manual_seed(seed)
    V = n_views

    angles = [i * (360.0 / V) for i in range(V)]
    if tar_angle is None:
        tar_angle = 0.5 * (angles[0] + (angles[1] if V > 1 else 40.0)) * 0.7 + 7.0
    Ks, exts = [], []
    for a in angles:
        R, t = _ring_camera(a, height=0.15 * math.sin(math.radians(a * 1.7)))
        E = np.eye(4)
        E[:3, :3] = R
        E[:3, 3] = t
        exts.append(E)
        Ks.append(_intrinsics(H, W))
    K = torch.tensor(np.stack(Ks), dtype=dtype)
    extrin = torch.tensor(np.stack(exts), dtype=dtype)
    KRT = torch.bmm(K, extrin)  # same op order as reference src/model.py:342
    cam = {"KRT": KRT, "K": K, "Rt": extrin[:, :3, :4].contiguous(), "extrin": extrin,
           "znear": 2.0, "zfar": 5.0, "width": W, "height": H, "nml_scale": 100.0}

    Rt_, tt_ = _ring_camera(tar_angle, height=0.2)
    Et = np.eye(4)
    Et[:3, :3] = Rt_
    Et[:3, 3] = tt_
    Kt = torch.tensor(_intrinsics(Ht, Wt, tar_focal_at_512)[None], dtype=dtype)
    RTt = torch.tensor(Et[None], dtype=dtype)
    cam_tar = {"K": Kt, "RT": RTt, "KRT": torch.bmm(Kt, RTt), "width": Wt, "height": Ht,
               "nml_scale": 100.0, "znear": float(tar_near_far[0]), "zfar": float(tar_near_far[1])}

    box = torch.tensor([0.6, 1.6, 0.4], dtype=dtype)
    kpt3d = (torch.rand(1, N_KPT, 3, generator=g, dtype=dtype) - 0.5) * box
    bounds = torch.stack([kpt3d[0].min(0)[0] - 0.1, kpt3d[0].max(0)[0] + 0.1], 0)[None]

    img = torch.rand(V, 3, H, W, generator=g, dtype=dtype)
    feat_geo = [torch.randn(V, 64, H // 8, W // 8, generator=g, dtype=dtype),
                torch.randn(V, 8, H // 2, W // 2, generator=g, dtype=dtype)]
    feat_tex = torch.randn(V, 8, H // 4, W // 4, generator=g, dtype=dtype)

    if mask == "dense":
        fg = torch.ones(1, V, 1, H, W, dtype=torch.bool)
    elif mask == "ellipsoid":
        # pixel is foreground iff its camera ray hits the ellipsoid (x/a)^2+(y/b)^2+(z/c)^2=1
        semi = np.array([0.42, 0.95, 0.36])
        ys, xs = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
        fg_list = []
        for v in range(V):
            Kv = Ks[v][:3, :3]
            R = exts[v][:3, :3]
            t = exts[v][:3, 3]
            o = -R.T @ t
            d = np.stack([xs, ys, np.ones_like(xs)], -1) @ np.linalg.inv(Kv).T @ R  # world dirs
            os_, ds_ = o / semi, d / semi
            a = (ds_ * ds_).sum(-1)
            b = 2.0 * (ds_ * os_).sum(-1)
            c = (os_ * os_).sum() - 1.0
            fg_list.append((b * b - 4 * a * c) >= 0.0)
        fg = torch.tensor(np.stack(fg_list))[None, :, None]
    else:
        raise ValueError(f"unknown mask kind {mask!r}")

    scene = {"img": img, "feat_geo": feat_geo, "feat_tex": feat_tex, "cam": cam, "cam_tar": cam_tar,
             "sp_data": {"kpt3d": kpt3d, "extrin": extrin}, "bounds": bounds,
             "src_foreground_mask": fg, "n_views": V}
    return to_device(scene, device)


def to_device(obj, device):
    if isinstance(obj, torch.Tensor):
        return obj.to(device)
    if isinstance(obj, dict):
        return {k: to_device(v, device) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(to_device(v, device) for v in obj)
    return obj


# --------------------------------------------------------------------------------------------
# Hot-path parameters.  The layout below is the *effective* (weight-norm folded) parameter set
# of the reference modules mlp_geo / ibr_compress_gfeat / mlp_tex.
# name, reference state-dict prefix, (out, in), weight-normed?
HOTPATH_LAYERS = [
    ("g1_0", "mlp_geo.layers1.layers.0.linear", (128, 232), True),
    ("g1_1", "mlp_geo.layers1.layers.1.linear", (128, 128), True),
    ("g1_2", "mlp_geo.layers1.layers.2.linear", (120, 136), True),
    ("g1_3", "mlp_geo.layers1.layers.3.linear", (64, 120), False),
    ("g2_0", "mlp_geo.layers2.layers.0.linear", (64, 128), True),
    ("g2_1", "mlp_geo.layers2.layers.1.linear", (64, 64), True),
    ("g2_2", "mlp_geo.layers2.layers.2.linear", (2, 64), False),
    ("cmp", "ibr_compress_gfeat", (24, 128), False),
    ("re_0", "mlp_tex.ray_encoder.0", (16, 4), False),
    ("re_1", "mlp_tex.ray_encoder.2", (35, 16), False),
    ("bl_0", "mlp_tex.base_layer.0", (64, 105), False),
    ("bl_1", "mlp_tex.base_layer.2", (32, 64), False),
    ("v1_0", "mlp_tex.vis_layer1.0", (32, 32), False),
    ("v1_1", "mlp_tex.vis_layer1.2", (33, 32), False),
    ("v2_0", "mlp_tex.vis_layer2.0", (32, 32), False),
    ("v2_1", "mlp_tex.vis_layer2.2", (1, 32), False),
    ("o_0", "mlp_tex.out_layer.0", (16, 37), False),
    ("o_1", "mlp_tex.out_layer.2", (8, 16), False),
    ("o_2", "mlp_tex.out_layer.4", (1, 8), False),
]


def random_hotpath_state_dict(seed=0, bias_std=0.05, density_gain=25.0, density_bias=0.0):
    """A random state dict with the reference's parameter names/shapes for the hot-path modules.

    Used where the reference itself is not importable (GPU box): kaiming-like weights, non-zero
    biases (the reference's init leaves every bias at exactly 0, which would hide bias bugs) and a
    density head scaled so that alpha along a ray is neither ~0 nor saturated.  density_bias is added to the bias of the
    density output `rad` (mlp_geo.layers2.layers.2, row 1): a negative value makes relu(rad) exactly 0 in part of the visual
    hull, as a trained density is in the free space between the silhouettes' hull and the surface.
    """
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, prefix, (o, i), wn in HOTPATH_LAYERS:
        w = torch.randn(o, i, generator=g) * math.sqrt(2.0 / i)
        b = torch.randn(o, generator=g) * bias_std
        if wn:
            sd[prefix + ".weight_v"] = w
            sd[prefix + ".weight_g"] = w.norm(dim=1, keepdim=True) * (1.0 + 0.1 * torch.randn(o, 1, generator=g))
        else:
            sd[prefix + ".weight"] = w
        sd[prefix + ".bias"] = b
    sd["mlp_geo.layers2.layers.2.linear.weight"] *= density_gain
    sd["mlp_geo.layers2.layers.2.linear.bias"] = torch.tensor([0.0, 0.5 * density_gain * 0.05 + float(density_bias)])
    sd["mlp_tex.ani_al"] = torch.tensor(0.2)
    return sd


def perturb_reference_net(net, seed=7, bias_std=0.05, density_gain=25.0):
    """In-place: give a reference-constructed net non-zero biases and a usable density scale
    (same intent as ``random_hotpath_state_dict``; keeps the reference's seeded weights)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in net.named_parameters():
            if not name.startswith(("mlp_geo.", "mlp_tex.", "ibr_compress_gfeat.")):
                continue
            if name.endswith(".bias"):
                p.add_(torch.randn(p.shape, generator=g) * bias_std)
            if name.endswith("weight_g"):
                p.mul_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
        w = net.mlp_geo.layers2.layers[2].linear.weight
        w.mul_(density_gain)
        net.mlp_geo.layers2.layers[2].linear.bias.add_(torch.tensor([0.0, 0.5 * density_gain * 0.05]))
        net.mlp_tex.ani_al.fill_(0.37)
    return net
