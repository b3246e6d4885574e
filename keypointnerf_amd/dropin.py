"""Drop-in for the reference's ray-march seam.

The reference has no plugin interface; its hot path is reached through attribute lookup on the ``net``
object: ``net.batch_render_pifu_nerf`` (reference src/model.py:866,922),
``net.query`` (:979), ``net.ray_bbox_intersection`` (:1039), ``net.rgba2out`` (:1065,1085),
``net.importance_sample`` (:1075) and, one level up, ``net.render_pifu_nerf`` (:453).  ``install(net)``
rebinds exactly those attributes on a live ``KeypointNeRF`` instance — same names, same argument lists,
same return conventions — and leaves the module tree / parameter names (checkpoint format) untouched:
weights are read from ``net.state_dict()`` and re-packed whenever a parameter changes.

Eval path (``net.training == False``, ``uniform=True`` sampling as used by render_full_nerf_image,
src/model.py:453-473): ``kpn_render_rays``.  Training path (``net.training == True``, batch size 1 as
configs/zju.json:12): ``batch_render_pifu_nerf`` draws the patch centre, the stratified jitter, the
view-dropout masks, the density noise and the importance ``u`` with the same calls, shapes and order as the reference
(src/model.py:1008-1017, 1049-1053, 742-748, 993-994, 1129), renders with ``kpn_render_rays_train`` and is
differentiable: ``loss.backward()`` runs ``kpn_render_rays_train_backward`` and reaches ``weight_g`` / ``weight_v`` /
``bias`` / ``ani_al`` and the image encoders (through ``feat_geo`` / ``feat_tex``) by autograd.  Anything outside that
envelope (larger batches, more views, ``separate_cf``) is forwarded to the reference's own method — that is the
reference itself, not a fallback of this library.
"""
import types

import numpy as np
import torch

from . import ops
from .weights import plain_tensor_from_module

_OUT_KEYS = ("tex_fg", "depth", "alpha", "tex_fg_fine", "depth_fine", "alpha_fine", "sdf")


class _TrainRender(torch.autograd.Function):
    """kpn_render_rays_train / kpn_render_rays_train_backward as one differentiable op.  Differentiable inputs: the
    flat effective-parameter vector and the three feature maps; everything else (cameras, draws) rides in `cfg`."""

    @staticmethod
    def forward(ctx, plain, geo0, geo1, tex, cfg):
        scene = ops.PreparedScene(cfg["img"], cfg["cam"], [geo0.detach(), geo1.detach()], tex.detach(), cfg["sp_data"],
                                  cfg["fg_mask"], disable_fg_mask=cfg["disable_fg_mask"], sigma=cfg["sigma"])
        w = ops.PackedWeights.from_plain(plain, device=geo0.device)
        out = ops.render_rays_train(scene, w, cfg["tar"], cfg["bounds"], cfg["pix"], cfg["u_c"], cfg["u_f"], cfg["keep_c"],
                                    cfg["keep_f"], noise_coarse=cfg["noise_c"], noise_fine=cfg["noise_f"],
                                    rand_noise_std=cfg["noise_std"], n_coarse=cfg["Sc"], n_fine=cfg["Sf"])
        ctx.scene, ctx.w, ctx.cfg = scene, w, cfg
        return tuple(out[k].clone() for k in _OUT_KEYS)

    @staticmethod
    def backward(ctx, *grads):
        cfg = ctx.cfg
        g = {k: (None if gi is None else gi.contiguous()) for k, gi in zip(_OUT_KEYS, grads)}
        d_plain, d_g0, d_g1, d_tx = ops.render_rays_train_backward(
            ctx.scene, ctx.w, cfg["tar"], cfg["bounds"], cfg["pix"], cfg["u_c"], cfg["u_f"], cfg["keep_c"], cfg["keep_f"], g,
            noise_coarse=cfg["noise_c"], noise_fine=cfg["noise_f"], rand_noise_std=cfg["noise_std"], n_coarse=cfg["Sc"],
            n_fine=cfg["Sf"])
        return d_plain, d_g0.contiguous(), d_g1.contiguous(), d_tx.contiguous(), None

_SEAMS = ("batch_render_pifu_nerf", "render_pifu_nerf", "query", "rgba2out", "importance_sample", "ray_bbox_intersection")


class _State:
    def __init__(self, net):
        self.net = net
        self.weights = None
        self.weights_key = None
        self.scene = None
        self.scene_key = None
        self.plans = {}

    def packed_weights(self):
        params = [p for n, p in self.net.named_parameters() if n.startswith(("mlp_geo.", "mlp_tex.", "ibr_compress_gfeat."))]
        key = tuple((p.data_ptr(), p._version) for p in params)
        if key != self.weights_key:
            dev = params[0].device if params and params[0].is_cuda else "cuda"
            self.weights = ops.PackedWeights(self.net.state_dict(), device=dev)
            self.weights_key = key
        return self.weights

    def prepared_scene(self, img, cam, feat_geo, feat_tex, sp_data, fg_mask):
        tensors = [img, cam["KRT"], feat_geo[0], feat_geo[1], feat_tex, sp_data["kpt3d"], fg_mask]
        key = tuple((t.data_ptr(), t._version, tuple(t.shape)) for t in tensors)
        if key != self.scene_key:
            enc = getattr(self.net, "sp_encoder", None)
            sigma = float(getattr(enc, "kwargs", {}).get("sigma", 0.1)) if enc is not None else 0.1  # configs/zju.json:43
            self.scene = ops.PreparedScene(img, cam, feat_geo, feat_tex, sp_data, fg_mask,
                                           disable_fg_mask=getattr(self.net, "disable_fg_mask", False), sigma=sigma)
            self.scene_key = key
            self.plans = {}
        return self.scene


def install(net):
    """Rebinds the hot-path attributes of a reference ``KeypointNeRF`` instance to the HIP operators.
    Returns ``net``.  ``uninstall(net)`` restores the reference's methods."""
    st = _State(net)
    cls = type(net)
    ref = {k: getattr(cls, k) for k in _SEAMS if hasattr(cls, k)}

    def query(self, pts, cam, feat_geo=None, feat_tex=None, n_views=1, sp_data={}, tx_data={}, view=None,
              n_pts_samples=-1, **kwargs):
        if self.training:
            return ref["query"](self, pts, cam, feat_geo, feat_tex, n_views=n_views, sp_data=sp_data, tx_data=tx_data,
                                view=view, n_pts_samples=n_pts_samples, **kwargs)
        if pts.shape[0] != 1:
            raise NotImplementedError("eval requires batch size 1 (reference src/model.py:938,1191)")
        feat_geo = self.feat_geo if feat_geo is None else feat_geo
        scene = st.prepared_scene(tx_data["img"], cam, feat_geo, feat_tex, sp_data, kwargs["src_foreground_mask"])
        return ops.query(scene, st.packed_weights(), pts, view, mode=0)

    def train_render(net_, img_in, cam_in, n_views, cam_tar, tar_img, feat_geo, feat_tex, sp_data, **config):
        """Train branch of batch_render_pifu_nerf (src/model.py:1008-1108), batch size 1."""
        dev = img_in.device
        if feat_geo is None:
            feat_geo = net_.attach_geo_feat(img_in, return_val=True)
        if feat_tex is None:
            feat_tex = net_.attach_tex_feat(img_in, return_val=True)
        width = cam_tar.get("width", cam_in["width"])
        height = cam_tar.get("height", cam_in["height"])
        Sc, Sf = config.get("sample_per_ray_c", 64), config.get("sample_per_ray_f", 64)
        std = float(config.get("rand_noise_std", 0.0))
        out_h, out_w = net_.train_out_h, net_.train_out_w
        # patch around a random foreground pixel, src/model.py:1010-1016 (numpy RNG, as the reference)
        msk = config["msk"].squeeze()
        msk_coords = torch.stack(torch.where(msk)[::-1], -1)
        center = msk_coords[np.random.randint(0, msk_coords.shape[0], 1)]
        yg, xg = torch.meshgrid(torch.arange(0, out_h, device=dev), torch.arange(0, out_w, device=dev), indexing="ij")
        grids = torch.stack([xg, yg], -1).view(-1, 2) + (center.to(dev) - out_h // 2)
        grids = grids.clamp(0, min(width - 1, height - 1))
        R = grids.shape[0]
        index = (grids[:, 0] + grids[:, 1] * width)[None]

        def keep_bits():                                            # src/model.py:742-748
            if n_views == 1:
                return 1
            d = torch.zeros(1, n_views, 1, 1, device=dev)
            d[:, :1] = 1.0
            d[:, 1:] = (torch.rand_like(d[:, 1:]) > 0.5).float()
            perm = torch.rand_like(d).argsort(dim=1)
            k = torch.gather(d, 1, perm).reshape(-1)
            return int(sum(1 << i for i, x in enumerate(k.tolist()) if x > 0.5))

        # the reference's draws, in its order: jitter (:1052), coarse dropout (:745-746), coarse noise (:994),
        # importance u (:1129), fine dropout, fine noise
        u_c = torch.rand(1, R, Sc, device=dev)
        keep_c = keep_bits()
        noise_c = torch.randn(1, R * Sc, 1, device=dev) if std > 0.0 else None
        u_f = torch.rand(1, R, Sf).to(dev)                          # th.rand(...).to(device): a CPU draw, as :1129
        keep_f = keep_bits()
        noise_f = torch.randn(1, R * (Sc + Sf), 1, device=dev) if std > 0.0 else None
        enc = getattr(net_, "sp_encoder", None)
        cfg = dict(img=img_in, cam=cam_in, sp_data=sp_data, fg_mask=config["src_foreground_mask"],
                   disable_fg_mask=getattr(net_, "disable_fg_mask", False),
                   sigma=float(getattr(enc, "kwargs", {}).get("sigma", 0.1)) if enc is not None else 0.1,
                   tar={"K": cam_tar["K"], "RT": cam_tar["RT"], "znear": cam_tar.get("znear", cam_in["znear"]),
                        "zfar": cam_tar.get("zfar", cam_in["zfar"])},
                   bounds=config["bounds"], pix=grids.to(torch.int32), u_c=u_c, u_f=u_f, keep_c=keep_c, keep_f=keep_f,
                   noise_c=noise_c, noise_f=noise_f, noise_std=std, Sc=Sc, Sf=Sf)
        res = _TrainRender.apply(plain_tensor_from_module(net_), feat_geo[0], feat_geo[1], feat_tex, cfg)
        out = {}
        for k, v in zip(_OUT_KEYS, res):                            # (1,3,R) / (1,R) in patch order
            out[k] = v.view(1, 3, out_h, out_w) if k.startswith("tex") else v.view(1, out_h, out_w)
        if tar_img is not None:                                     # src/model.py:1097-1107
            with torch.no_grad():
                t = tar_img.reshape(*tar_img.shape[:2], -1)
                out["tar_img"] = torch.gather(t, 2, index[:, None].expand(-1, 3, -1)).view(*t.shape[:2], out_h, out_w)
                a = config["msk"].reshape(1, 1, -1)
                out["tar_alpha"] = torch.gather(a, 2, index[:, None].expand(-1, 1, -1)).view(1, 1, out_h, out_w).float()
        return out

    def batch_render_pifu_nerf(net_, img_in, cam_in, n_views, cam_tar, level=2, stride=0, tar_img=None, feat_geo=None,
                               feat_tex=None, sp_data={}, objcenter=None, **config):
        if net_.training:
            if (img_in.shape[0] // n_views == 1 and n_views <= 16 and config.get("fine", False) and not config.get("uniform", False)
                    and not config.get("separate_cf", False) and "msk" in config):
                return train_render(net_, img_in, cam_in, n_views, cam_tar, tar_img, feat_geo, feat_tex, sp_data, **config)
            return ref["batch_render_pifu_nerf"](net_, img_in, cam_in, n_views, cam_tar, level, stride, tar_img, feat_geo,
                                                 feat_tex, sp_data, objcenter, **config)
        if not config.get("uniform", False):
            return ref["batch_render_pifu_nerf"](net_, img_in, cam_in, n_views, cam_tar, level, stride, tar_img, feat_geo,
                                                 feat_tex, sp_data, objcenter, **config)
        if img_in.shape[0] // n_views != 1:
            raise NotImplementedError("eval requires batch size 1 (reference src/model.py:938,1191)")
        if config.get("separate_cf", False):
            raise NotImplementedError("separate_cf is not used by configs/zju.json")
        if feat_geo is None:
            feat_geo = net_.attach_geo_feat(img_in, return_val=True)
        if feat_tex is None:
            feat_tex = net_.attach_tex_feat(img_in, return_val=True)
        width = cam_tar.get("width", cam_in["width"])
        height = cam_tar.get("height", cam_in["height"])
        step = 2 ** (level - 1)
        assert width % step == 0 and height % step == 0          # reference src/model.py:999
        if isinstance(stride, int):
            assert stride < step
            x0 = y0 = int(stride)
        elif isinstance(stride, torch.Tensor):
            assert stride.max().item() < step
            x0, y0 = int(stride.reshape(-1, 2)[0, 0].item()), int(stride.reshape(-1, 2)[0, 1].item())
        else:
            raise NotImplementedError("unsupported stride type")    # reference src/model.py:1006
        nx, ny = width // step, height // step
        scene = st.prepared_scene(img_in, cam_in, feat_geo, feat_tex, sp_data, config["src_foreground_mask"])
        tar = {"K": cam_tar["K"], "RT": cam_tar["RT"], "znear": cam_tar.get("znear", cam_in["znear"]),
               "zfar": cam_tar.get("zfar", cam_in["zfar"])}
        fine = bool(config.get("fine", False))
        key = (x0, y0, step, nx, ny, config.get("sample_per_ray_c", 64), config.get("sample_per_ray_f", 64), fine)
        # a fresh plan per call keeps the reference's "new tensors out" contract; geometry is cached only
        plan = ops.RenderPlan(scene, key[:5], key[5], key[6], fine=fine)
        res = ops.render_rays(scene, st.packed_weights(), tar, config["bounds"], plan=plan)
        out = dict(res)
        if tar_img is not None:                                     # reference src/model.py:1097-1107
            ys = torch.arange(ny, device=img_in.device) * step + y0
            xs = torch.arange(nx, device=img_in.device) * step + x0
            index = (ys[:, None] * width + xs[None, :]).reshape(1, -1).long()
            with torch.no_grad():
                t = tar_img.reshape(*tar_img.shape[:2], -1)
                out["tar_img"] = torch.gather(t, 2, index[:, None].expand(-1, 3, -1)).view(*t.shape[:2], ny, nx)
                if "msk" in config:
                    a = config["msk"].reshape(1, 1, -1)
                    out["tar_alpha"] = torch.gather(a, 2, index[:, None].expand(-1, 1, -1)).view(1, 1, ny, nx).float()
        return out

    def render_pifu_nerf(net_, img_in, cam_in, cam_tar, level=5, sp_data={}, bkg_emb=None, camcenter=None, objcenter=None,
                         tar_img=None, **config):
        """The reference renders stride^2 strided tiles and re-assembles them with pixel_shuffle
        (src/model.py:916-938); the same rays are marched here as ONE full-frame pass (step 1)."""
        n_views = img_in.shape[0]
        feat_geo = net_.attach_geo_feat(img_in, return_val=True)
        feat_tex = net_.attach_tex_feat(img_in, return_val=True)
        out = batch_render_pifu_nerf(net_, img_in, cam_in, n_views, cam_tar, 1, 0, tar_img, feat_geo, feat_tex, sp_data,
                                     objcenter, **config)
        ret = {}
        for k, v in out.items():                                   # same filtering as src/model.py:924-938
            if v is None or len(v.shape) < 3:
                continue
            if len(v.shape) == 3:
                v = v[:, None]
            ret[k] = v.detach().cpu()[0]
        return ret

    net._kpnerf_reference_methods = {k: net.__dict__.get(k) for k in _SEAMS}
    net.query = types.MethodType(query, net)
    net.batch_render_pifu_nerf = batch_render_pifu_nerf            # static in the reference: called as net.f(net, ...)
    net.render_pifu_nerf = render_pifu_nerf
    net.rgba2out = lambda rgba, z: ops.rgba2out(rgba, z)
    net.importance_sample = lambda contrib, z, n, uniform=False: ops.importance_sample(contrib, z, n, uniform=uniform)
    net.ray_bbox_intersection = lambda bounds, orig, direct: ops.ray_bbox_intersection(bounds, orig, direct)
    net._kpnerf_state = st
    return net


def uninstall(net):
    for k in _SEAMS:
        if k in net.__dict__:
            del net.__dict__[k]
    for k in ("_kpnerf_state", "_kpnerf_reference_methods"):
        net.__dict__.pop(k, None)
    return net
