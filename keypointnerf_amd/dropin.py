"""Drop-in for the reference's ray-march seam.

The reference has no plugin interface; its hot path is reached through attribute lookup on the ``net``
object: ``net.batch_render_pifu_nerf`` (reference src/model.py:866,922),
``net.query`` (:979), ``net.ray_bbox_intersection`` (:1039), ``net.rgba2out`` (:1065,1085),
``net.importance_sample`` (:1075) and, one level up, ``net.render_pifu_nerf`` (:453).  ``install(net)``
rebinds exactly those attributes on a live ``KeypointNeRF`` instance — same names, same argument lists,
same return conventions — and leaves the module tree / parameter names (checkpoint format) untouched:
weights are read from ``net.state_dict()`` and re-packed whenever a parameter changes.

Scope (round 1): the eval path (``net.training == False``, ``uniform=True`` sampling as used by
render_full_nerf_image, src/model.py:453-473).  In training mode the rebinding forwards to the
reference's own methods (view dropout, stratified jitter, density noise and autograd are SURVEY.md
§8 config 4, not built yet) — that is the reference itself, not a fallback of this library.
"""
import types

import torch

from . import ops

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

    def batch_render_pifu_nerf(net_, img_in, cam_in, n_views, cam_tar, level=2, stride=0, tar_img=None, feat_geo=None,
                               feat_tex=None, sp_data={}, objcenter=None, **config):
        if net_.training or not config.get("uniform", False):
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
