"""Drop-in for the reference's ray-march seam.

The reference has no plugin interface; its hot path is reached through attribute lookup on the ``net``
object: ``net.batch_render_pifu_nerf`` (reference src/model.py:866,922),
``net.query`` (:979), ``net.ray_bbox_intersection`` (:1039), ``net.rgba2out`` (:1065,1085),
``net.importance_sample`` (:1075) and, one level up, ``net.render_pifu_nerf`` (:453).  ``install(net)``
rebinds exactly those attributes on a live ``KeypointNeRF`` instance — same names, same argument lists
(the reference calls both render methods with ``net=`` by keyword, :454 and :866), same return
conventions — and leaves the module tree / parameter names (checkpoint format) untouched: weights are
read from the live parameters and re-packed whenever one changes.

Served by the library (batch size 1, as configs/zju.json:12 and src/model.py:938,1191):

* eval, ``uniform=True`` (render_full_nerf_image, src/model.py:453-473): ``kpn_render_rays``;
* eval, ``uniform=False`` (validation_step -> KeypointNeRF.forward in eval mode, src/model.py:509-526 with
  dr_kwargs of configs/zju.json:101-108): strided pixel grid, stratified jitter, density noise and random
  importance samples drawn with the reference's calls in the reference's order, no view dropout
  (src/model.py:742 is train-only): ``kpn_render_rays_train`` with every view kept;
* train (``net.training``): patch centre, jitter, view-dropout masks, density noise and importance ``u`` drawn
  as the reference does (src/model.py:1008-1017, 1049-1053, 742-748, 993-994, 1129); differentiable:
  ``loss.backward()`` runs ``kpn_render_rays_train_backward`` and reaches ``weight_g`` / ``weight_v`` / ``bias`` /
  ``ani_al`` and the image encoders (through ``feat_geo`` / ``feat_tex``) by autograd
  (``torch.ops.kpnerf.render_rays_train``, keypointnerf_amd/torch_ops.py).

Anything outside that envelope (larger batches, ``separate_cf``, a spatial encoder other than the shipped
``rel_z_decay``) is refused at ``install`` time or forwarded to the reference's own method — that is the
reference itself, not a fallback of this library.
"""
import types

import numpy as np
import torch

from . import ops
from .weights import plain_tensor_from_module

_OUT_KEYS = ("tex_fg", "depth", "alpha", "tex_fg_fine", "depth_fine", "alpha_fine", "sdf")
_SEAMS = ("batch_render_pifu_nerf", "render_pifu_nerf", "query", "rgba2out", "importance_sample", "ray_bbox_intersection")
_HOT_PREFIXES = ("mlp_geo.", "mlp_tex.", "ibr_compress_gfeat.")
# rays of one stochastic render whose pass state is kept for the backward (the shipped patch is 64 x 64 = 4096, configs/zju.json:36-37)
_KEEP_STATE_MAX_RAYS = 16384


def _version_key(tensors):
    """(identity, version) key of a list of tensors, or None when a version counter is unavailable (inference
    tensors — Lightning runs validate/test under torch.inference_mode — raise on ._version)."""
    key = []
    for t in tensors:
        if t is None:
            key.append(None)
            continue
        if t.is_inference():
            return None
        key.append((id(t), t.data_ptr(), t._version, tuple(t.shape)))
    return tuple(key)


def check_supported(net):
    """The kernels implement the shipped configuration (reference configs/zju.json:39-45): keypoint-relative depth
    encoding with Gaussian decay, 3 octaves, 24 keypoints, scale 1.  A model built with another SpatialEncoder
    branch of the same feature width would pass the weight-shape check and render silently wrong: refuse it."""
    enc = getattr(net, "sp_encoder", None)
    if enc is None:
        return
    want = {"sp_type": "rel_z_decay", "sp_level": 3, "n_kpt": 24}
    for k, v in want.items():
        got = getattr(enc, k, v)
        if got != v:
            raise NotImplementedError(f"keypointnerf_amd serves sp_encoder.{k} == {v!r} only (got {got!r}; "
                                      f"reference src/spatial.py:88-133 has other branches, configs/zju.json selects this one)")
    if float(getattr(enc, "scale", 1.0)) != 1.0:
        raise NotImplementedError("keypointnerf_amd serves sp_encoder.scale == 1.0 only (configs/zju.json:42)")


def encoder_sigma(net):
    """sp_args['sigma'] of the live module; the reference's fallback is 150.0 (src/spatial.py:112)."""
    enc = getattr(net, "sp_encoder", None)
    return float(getattr(enc, "kwargs", {}).get("sigma", 150.0)) if enc is not None else 150.0


class _State:
    def __init__(self, net):
        self.net = net
        self.weights = None
        self.weights_key = None
        self.plans = {}
        self.feats = None            # (encoder key, (img, image key), feat_geo, feat_tex): encoder outputs of the last source set
        # the module's own attach_im_feat(im) (no return_val): encoder key at that moment + the very feat_geo / feat_tex OBJECTS it
        # left on the module.  The reference overwrites net.feat_geo / net.feat_tex on EVERY attach_*_feat call, return_val or not
        # (src/model.py:664,678), while net.im only changes without return_val: the shortcut in encoder_features therefore also
        # requires that the module still holds these very objects.
        self.attached = None         # (encoder key, feat_geo object)
        self.attached_tex = None     # feat_tex object computed by attach_tex_feat(im) for the SAME im, else None

    def packed_weights(self):
        params = [p for n, p in self.net.named_parameters() if n.startswith(_HOT_PREFIXES)]
        key = _version_key(params)
        if key is None or key != self.weights_key:
            dev = params[0].device if params and ops._on_gpu(params[0]) else "cuda"
            # packed on the device from the live parameters (weight-norm fold + MFMA operand order): no host round trip
            with torch.no_grad():
                self.weights = ops.PackedWeights.from_plain(plain_tensor_from_module(self.net).to(dev), device=dev)
            self.weights_key = key
        return self.weights

    def prepared_scene(self, img, cam, feat_geo, feat_tex, sp_data, fg_mask):
        """A fresh PreparedScene per call: kpn_scene_prepare is ~0.1 ms next to a render of tens of ms, and keying a
        cache on data_ptr/_version is unsafe (inference tensors have no version counter; a recycled address with the
        same shape would serve a stale workspace).  PreparedScene keeps every input tensor alive."""
        if "transf" in cam:
            raise NotImplementedError("cam['transf'] (src/model.py:716-718) is not produced by decode_batch and not served")
        return ops.PreparedScene(img, cam, feat_geo, feat_tex, sp_data, fg_mask,
                                 disable_fg_mask=getattr(self.net, "disable_fg_mask", False), sigma=encoder_sigma(self.net))

    def plan(self, scene, grid, Sc, Sf, fine):
        """Render plans (outputs + workspace) are reused across calls with the same geometry; the outputs handed to
        the caller are clones, so the reference's "new tensors out" contract holds."""
        d = scene.desc
        key = (tuple(grid), Sc, Sf, fine, d.n_views, d.src_h, d.src_w, d.geo0_h, d.geo0_w, d.geo1_h, d.geo1_w, d.tex_h, d.tex_w,
               str(scene.ws.device))
        plan = self.plans.get(key)
        if plan is None:
            if len(self.plans) >= 4:
                self.plans.clear()
            plan = self.plans[key] = ops.RenderPlan(scene, grid, Sc, Sf, fine=fine)
        return plan

    def encoder_key(self):
        """(identity, storage, version) of every parameter AND buffer of the two encoders.  They are ordinary tensors (never
        inference tensors), so the version counter is always there: an optimizer step, load_state_dict or a BatchNorm update
        changes the key even when the source images are inference tensors (Lightning validate / test)."""
        net = self.net
        key = []
        for name in ("geo_encoder", "tex_encoder"):
            m = getattr(net, name, None)
            if m is None:
                continue
            for t in list(m.parameters()) + list(m.buffers()):
                if t.is_inference():
                    return None                       # no version counter: never reuse
                key.append((id(t), t.data_ptr(), t._version))
        return tuple(key)

    @staticmethod
    def _same_images(cached, img_in):
        """`cached` = (tensor, version key or None).  Versioned tensors compare by identity + version, inference tensors by
        content against the clone that was kept."""
        img0, k0 = cached
        k = _version_key([img_in])
        if k is not None and k0 is not None:
            return k == k0
        return img0.numel() == img_in.numel() and img0.device == img_in.device and bool(torch.equal(img0.reshape(img_in.shape), img_in))

    def note_attached(self, im):
        """Called by the wrapped attach_geo_feat (install) after a call WITHOUT return_val: the module now holds net.im = im.clone()
        and feat_geo of `im`, computed with the encoder state of this moment (src/model.py:653-666).  feat_tex is noted by the
        wrapped attach_tex_feat when it is computed for the same images."""
        self.attached = (self.encoder_key(), getattr(self.net, "feat_geo", None))
        self.attached_tex = None

    def note_attached_tex(self, im):
        """Wrapped attach_tex_feat without return_val: feat_tex belongs to the attached source set only if `im` is net.im's content
        (a bare attach_tex_feat(B) after attach_im_feat(A) must not pair F_tex(B) with F_geo(A))."""
        net = self.net
        im0 = getattr(net, "im", None)
        ok = (self.attached is not None and im0 is not None and im0.numel() == im.numel() and im0.device == im.device
              and bool(torch.equal(im0.reshape(im.shape), im)))
        self.attached_tex = getattr(net, "feat_tex", None) if ok else None

    def encoder_features(self, img_in):
        """feat_geo / feat_tex of the source images.  render_novel_views runs both encoders once per source set
        (attach_im_feat, src/model.py:479) and render_pifu_nerf then runs them AGAIN for every target camera
        (:913-914) — 28 M parameters of convolutions per orbit frame for identical inputs.  In eval mode the encoders are
        deterministic, so (1) the maps the module already holds from attach_im_feat are taken when they belong to these images
        and to the encoders' current state (render_video_zju: every frame is a new source set, attached once), and (2) otherwise
        the maps of the last call are kept per (source images, encoder state)."""
        net = self.net
        if net.training or (torch.is_grad_enabled() and any(p.requires_grad for p in net.parameters())):
            return net.attach_geo_feat(img_in, return_val=True), net.attach_tex_feat(img_in, return_val=True)
        ekey = self.encoder_key()
        if ekey is not None:
            im0 = getattr(net, "im", None)
            # the attached maps: same encoder state, the module still holds the very objects attach_im_feat left (any later
            # attach_*_feat(other, return_val=True) — render_pifu_nerf, an eval forward — replaces them), same images
            fg, ft = getattr(net, "feat_geo", None), getattr(net, "feat_tex", None)
            if (self.attached is not None and self.attached[0] == ekey and im0 is not None and fg is not None and fg is self.attached[1]
                    and (getattr(net, "tex_encoder", None) is None or (ft is not None and ft is self.attached_tex))
                    and self._same_images((im0, None), img_in)):
                return fg, ft
            if self.feats is not None and self.feats[0] == ekey and self._same_images(self.feats[1], img_in):
                return self.feats[2], self.feats[3]
        g = net.attach_geo_feat(img_in, return_val=True)
        t = net.attach_tex_feat(img_in, return_val=True)
        ikey = _version_key([img_in])
        # versioned tensors: the strong reference pins identity; inference tensors: a clone pins the content
        self.feats = (ekey, (img_in if ikey is not None else img_in.clone(), ikey), g, t) if ekey is not None else None
        return g, t


def _draw_keep_bits(n_views, dev):
    """Per-view dropout of one query call, drawn exactly like src/model.py:742-748."""
    if n_views == 1:
        return 1
    d = torch.zeros(1, n_views, 1, 1, device=dev)
    d[:, :1] = 1.0
    d[:, 1:] = (torch.rand_like(d[:, 1:]) > 0.5).float()
    perm = torch.rand_like(d).argsort(dim=1)
    k = torch.gather(d, 1, perm).reshape(-1)
    return int(sum(1 << i for i, x in enumerate(k.tolist()) if x > 0.5))


def install(net, rows_mode=None):
    """Rebinds the hot-path attributes of a reference ``KeypointNeRF`` instance to the HIP operators.
    Returns ``net``.  ``uninstall(net)`` restores the reference's methods.  ``rows_mode`` (optional) selects the rows
    kernel process-wide (``ops.set_geo_rows_mode``: 3 = default, two fp16 pieces per operand; 2 = three bf16 pieces; 0 = fp32 MFMA)."""
    from . import torch_ops  # noqa: F401  (registers torch.ops.kpnerf.*)
    check_supported(net)
    if rows_mode is not None:
        ops.set_geo_rows_mode(rows_mode)
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

    def tar_dict(cam_in, cam_tar):
        return {"K": cam_tar["K"], "RT": cam_tar["RT"], "znear": cam_tar.get("znear", cam_in["znear"]),
                "zfar": cam_tar.get("zfar", cam_in["zfar"])}

    def gt_gather(out, tar_img, index, config, h, w):                 # src/model.py:1097-1107
        if tar_img is None:
            return
        with torch.no_grad():
            assert tar_img.shape[0] == index.shape[0]
            t = tar_img.reshape(*tar_img.shape[:2], -1)
            out["tar_img"] = torch.gather(t, 2, index[:, None].expand(-1, 3, -1)).view(*t.shape[:2], h, w)
            if "msk" in config:
                a = config["msk"].reshape(1, 1, -1)
                out["tar_alpha"] = torch.gather(a, 2, index[:, None].expand(-1, 1, -1)).view(1, 1, h, w).float()

    def stochastic_render(net, img_in, cam_in, n_views, cam_tar, grids, out_h, out_w, tar_img, feat_geo, feat_tex, sp_data,
                          dropout, **config):
        """The `uniform=False` sampling of batch_render_pifu_nerf for the pixels `grids` (R,2): train branch
        (dropout=True, differentiable) and eval-mode validation (dropout=False)."""
        dev = img_in.device
        Sc, Sf = config.get("sample_per_ray_c", 64), config.get("sample_per_ray_f", 64)
        std = float(config.get("rand_noise_std", 0.0))
        R = grids.shape[0]
        width = cam_tar.get("width", cam_in["width"])
        index = (grids[:, 0] + grids[:, 1] * width)[None].long()
        all_views = (1 << n_views) - 1
        # the reference's draws, in its order: jitter (:1052), coarse dropout (:745-746), coarse noise (:994),
        # importance u (:1129), fine dropout, fine noise
        u_c = torch.rand(1, R, Sc, device=dev)
        keep_c = _draw_keep_bits(n_views, dev) if dropout else all_views
        noise_c = torch.randn(1, R * Sc, 1, device=dev) if std > 0.0 else None
        u_f = torch.rand(1, R, Sf).to(dev)                          # th.rand(...).to(device): a CPU draw, as :1129
        keep_f = _draw_keep_bits(n_views, dev) if dropout else all_views
        noise_f = torch.randn(1, R * (Sc + Sf), 1, device=dev) if std > 0.0 else None
        tar = tar_dict(cam_in, cam_tar)
        fg = config["src_foreground_mask"]
        extrin = sp_data["extrin"] if "extrin" in sp_data else cam_in["extrin"]
        if "transf" in cam_in:
            raise NotImplementedError("cam['transf'] (src/model.py:716-718) is not served")
        if torch.is_grad_enabled():
            plain = plain_tensor_from_module(net)
        else:
            with torch.no_grad():
                plain = plain_tensor_from_module(net)
        res = torch.ops.kpnerf.render_rays_train(
            plain, feat_geo[0], feat_geo[1], feat_tex, img_in, cam_in["KRT"], extrin, sp_data["kpt3d"],
            None if getattr(net, "disable_fg_mask", False) else fg,
            [float(cam_in["znear"]), float(cam_in["zfar"]), float(cam_in.get("nml_scale", 100.0)), encoder_sigma(net)],
            tar["K"], tar["RT"], config["bounds"], float(tar["znear"]), float(tar["zfar"]), grids.to(torch.int32), u_c, u_f,
            noise_c, noise_f, int(keep_c), int(keep_f), std, int(Sc), int(Sf),
            # training: keep the forward's pass state (about 1 KB per field evaluation) so that loss.backward() does not repeat
            # the forward — within the training budget only: a whole validation frame rendered with gradients enabled
            # (net.eval() without no_grad) stays differentiable, its backward repeats the forward instead of pinning tens of GB
            bool(torch.is_grad_enabled() and R <= _KEEP_STATE_MAX_RAYS
                 and (plain.requires_grad or feat_geo[0].requires_grad or feat_tex.requires_grad)))
        out = {}
        for k, v in zip(_OUT_KEYS, res):                            # (1,3,R) / (1,R) in pixel-list order
            out[k] = v.view(1, 3, out_h, out_w) if k.startswith("tex") else v.view(1, out_h, out_w)
        gt_gather(out, tar_img, index, config, out_h, out_w)
        return out

    def batch_render_pifu_nerf(net, img_in, cam_in, n_views, cam_tar, level=2, stride=0, tar_img=None, feat_geo=None,
                               feat_tex=None, sp_data={}, objcenter=None, **config):
        def reference():
            return ref["batch_render_pifu_nerf"](net, img_in, cam_in, n_views, cam_tar, level, stride, tar_img, feat_geo,
                                                 feat_tex, sp_data, objcenter, **config)
        batch_size = img_in.shape[0] // n_views
        uniform, fine = bool(config.get("uniform", False)), bool(config.get("fine", False))
        served = batch_size == 1 and n_views <= 16 and not config.get("separate_cf", False)
        if net.training:
            if not (served and fine and not uniform and "msk" in config):
                return reference()
        elif not served:
            if batch_size != 1:
                raise NotImplementedError("eval requires batch size 1 (reference src/model.py:938,1191)")
            raise NotImplementedError("separate_cf / more than 16 source views are not used by configs/zju.json")
        elif not uniform and not fine:
            return reference()                                       # stratified coarse-only: not a shipped configuration
        if feat_geo is None:
            feat_geo = net.attach_geo_feat(img_in, return_val=True)
        if feat_tex is None:
            feat_tex = net.attach_tex_feat(img_in, return_val=True)
        dev = img_in.device
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
        if net.training:
            # patch around a random foreground pixel, src/model.py:1008-1017 (numpy RNG, as the reference)
            out_h, out_w = net.train_out_h, net.train_out_w
            msk = config["msk"].squeeze()
            msk_coords = torch.stack(torch.where(msk)[::-1], -1)
            center = msk_coords[np.random.randint(0, msk_coords.shape[0], 1)]
            yg, xg = torch.meshgrid(torch.arange(0, out_h, device=dev), torch.arange(0, out_w, device=dev), indexing="ij")
            grids = torch.stack([xg, yg], -1).view(-1, 2) + (center.to(dev) - out_h // 2)
            grids = grids.clamp(0, min(width - 1, height - 1))
            return stochastic_render(net, img_in, cam_in, n_views, cam_tar, grids, out_h, out_w, tar_img, feat_geo, feat_tex,
                                     sp_data, True, **config)
        nx, ny = width // step, height // step
        if not uniform:                                              # validation: src/model.py:1018-1022 + :1049-1053
            yg, xg = torch.meshgrid(torch.arange(0, height, step, device=dev), torch.arange(0, width, step, device=dev), indexing="ij")
            grids = torch.stack([xg, yg], -1).view(-1, 2) + torch.tensor([x0, y0], device=dev)
            # like the reference's eval-mode forward, differentiable when the caller has gradients enabled (fine-tuning with
            # net.eval() to freeze the norm layers); Lightning's validation loop runs it under no_grad / inference_mode
            return stochastic_render(net, img_in, cam_in, n_views, cam_tar, grids, ny, nx, tar_img, feat_geo, feat_tex,
                                     sp_data, False, **config)
        scene = st.prepared_scene(img_in, cam_in, feat_geo, feat_tex, sp_data, config["src_foreground_mask"])
        Sc, Sf = config.get("sample_per_ray_c", 64), config.get("sample_per_ray_f", 64)
        plan = st.plan(scene, (x0, y0, step, nx, ny), Sc, Sf, fine)
        res = ops.render_rays(scene, st.packed_weights(), tar_dict(cam_in, cam_tar), config["bounds"], plan=plan)
        out = {k: v.clone() for k, v in res.items()}
        if tar_img is not None:
            ys = torch.arange(ny, device=dev) * step + y0
            xs = torch.arange(nx, device=dev) * step + x0
            index = (ys[:, None] * width + xs[None, :]).reshape(1, -1).long()
            gt_gather(out, tar_img, index, config, ny, nx)
        return out

    def render_pifu_nerf(net, img_in, cam_in, cam_tar, level=5, sp_data={}, bkg_emb=None, camcenter=None, objcenter=None,
                         tar_img=None, **config):
        """The reference renders stride^2 strided tiles and re-assembles them with pixel_shuffle
        (src/model.py:916-938); the same rays are marched here as ONE full-frame pass (step 1).  The encoders are
        not re-run when the source images are the ones already encoded (:913-914 vs :479)."""
        n_views = img_in.shape[0]
        feat_geo, feat_tex = st.encoder_features(img_in)
        out = batch_render_pifu_nerf(net, img_in, cam_in, n_views, cam_tar, 1, 0, tar_img, feat_geo, feat_tex, sp_data,
                                     objcenter, **config)
        ret = {}
        for k, v in out.items():                                   # same filtering as src/model.py:924-938
            if v is None or len(v.shape) < 3:
                continue
            if len(v.shape) == 3:
                v = v[:, None]
            ret[k] = v.detach().cpu()[0]
        return ret

    def importance_sample(contrib, z, sample_per_ray, uniform=False):
        u = None
        if not uniform:                                            # a CPU draw moved to the device, as src/model.py:1129
            u = torch.rand(*contrib.shape[:-1], sample_per_ray).to(contrib.device)
        return ops.importance_sample(contrib, z, sample_per_ray, uniform=uniform, u=u)

    def attach_geo_feat(self, im, return_val=False):
        r = cls.attach_geo_feat(self, im, return_val)
        if not return_val:                                         # the module keeps im / feat_geo (src/model.py:653-666)
            st.note_attached(im)
        return r

    def attach_tex_feat(self, im, return_val=False):
        r = cls.attach_tex_feat(self, im, return_val)
        if not return_val:
            st.note_attached_tex(im)
        return r

    net._kpnerf_reference_methods = {k: net.__dict__.get(k) for k in _SEAMS}
    if hasattr(cls, "attach_geo_feat"):
        net.attach_geo_feat = types.MethodType(attach_geo_feat, net)
    if hasattr(cls, "attach_tex_feat"):
        net.attach_tex_feat = types.MethodType(attach_tex_feat, net)
    net.query = types.MethodType(query, net)
    net.batch_render_pifu_nerf = batch_render_pifu_nerf            # static in the reference: called as net.f(net=net, ...)
    net.render_pifu_nerf = render_pifu_nerf
    net.rgba2out = lambda rgba, z: ops.rgba2out(rgba, z)
    net.importance_sample = importance_sample
    net.ray_bbox_intersection = lambda bounds, orig, direct: ops.ray_bbox_intersection(bounds, orig, direct)
    net._kpnerf_state = st
    return net


def uninstall(net):
    for k in _SEAMS + ("attach_geo_feat", "attach_tex_feat"):
        if k in net.__dict__:
            del net.__dict__[k]
    for k in ("_kpnerf_state", "_kpnerf_reference_methods"):
        net.__dict__.pop(k, None)
    return net
