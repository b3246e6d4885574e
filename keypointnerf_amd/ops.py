"""PyTorch-facing operators of the gfx950 ray-march library (device tensors in, device tensors out).

Each function mirrors one callable of the reference (names, argument meaning, shapes, error
behaviour) and forwards to the C ABI in include/kpnerf.h on torch's CURRENT HIP stream:

    ray_bbox_intersection  <- KeypointNeRF.ray_bbox_intersection  (reference src/model.py:1178-1237)
    importance_sample      <- KeypointNeRF.importance_sample      (src/model.py:1110-1148)
    rgba2out               <- KeypointNeRF.rgba2out               (src/model.py:1150-1176)
    query                  <- KeypointNeRF.query                  (src/model.py:690-843, eval mode)
    render_rays            <- KeypointNeRF.batch_render_pifu_nerf (src/model.py:942-1108, eval branch)

torch is plumbing here (device memory, streams); the arithmetic is in csrc/*.hip.  Every op raises if
its tensors are not on a HIP device or the library is missing — there is no CPU/eager fallback.
"""
import ctypes

import numpy as np
import torch

from . import lib as kl
from .weights import effective_weights, flatten_plain

_f32 = torch.float32


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _on_gpu(t):
    return t.is_cuda


def _dev(t, name, dtype=_f32):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not _on_gpu(t):
        raise RuntimeError(f"{name} must live on the GPU (got {t.device}); keypointnerf_amd has no CPU path")
    if t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous()


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


# ------------------------------------------------------------------------------------------------
class PackedWeights:
    """Hot-path parameters packed for the kernels; built from the reference's state dict / module."""

    def __init__(self, state_dict_or_module, device="cuda"):
        sd = state_dict_or_module.state_dict() if hasattr(state_dict_or_module, "state_dict") else state_dict_or_module
        L = kl.get_library()
        plain = flatten_plain(effective_weights(sd))
        if plain.size != L.kpn_plain_weight_floats():
            raise ValueError("unexpected hot-path parameter count")
        packed = np.zeros(L.kpn_packed_weight_floats(), np.float32)
        L.check(L.kpn_pack_weights(plain.ctypes.data_as(ctypes.c_void_p), packed.ctypes.data_as(ctypes.c_void_p)))
        # the packers' count of weights beyond fp16's range (include/kpnerf.h kpn_packed_f16_range_check).  Nothing to do here:
        # the two-fp16-piece kernels read the same count on the device and leave the work to the fp32-range kernels (range guard)
        self.f16_beyond = int(packed[-4])
        self.tensor = torch.from_numpy(packed).to(device)

    @classmethod
    def from_plain(cls, plain, device="cuda"):
        """From the flat effective-parameter vector (weights.flatten_plain / plain_tensor_from_module layout).  A CUDA
        tensor is packed on the device (kpn_pack_weights_device: no host round trip, asynchronous)."""
        L = kl.get_library()
        if isinstance(plain, torch.Tensor) and _on_gpu(plain):
            flat = plain.detach().to(_f32).contiguous()
            if flat.numel() != L.kpn_plain_weight_floats():
                raise ValueError("unexpected hot-path parameter count")
            self = cls.__new__(cls)
            self.tensor = torch.empty(L.kpn_packed_weight_floats(), dtype=_f32, device=flat.device)
            L.check(L.kpn_pack_weights_device(_p(flat), _p(self.tensor), _stream()))
            self._plain = flat  # keeps the source alive until the stream has consumed it
            return self
        flat = np.ascontiguousarray(plain.detach().float().cpu().numpy() if isinstance(plain, torch.Tensor) else plain, dtype=np.float32)
        if flat.size != L.kpn_plain_weight_floats():
            raise ValueError("unexpected hot-path parameter count")
        packed = np.zeros(L.kpn_packed_weight_floats(), np.float32)
        L.check(L.kpn_pack_weights(flat.ctypes.data_as(ctypes.c_void_p), packed.ctypes.data_as(ctypes.c_void_p)))
        self = cls.__new__(cls)
        self.tensor = torch.from_numpy(packed).to(device)
        return self


class PreparedScene:
    """kpn_scene_desc + the prepared (channels-last) device workspace for one set of source views.

    Arguments are the reference's own objects (reference src/model.py:336-355, 653-680):
    img (V,3,H,W); cam dict {KRT,(K),extrin|sp_data['extrin'],width,height,znear,zfar,nml_scale};
    feat_geo [ (V,64,h0,w0), (V,8,h1,w1) ]; feat_tex (V,8,ht,wt); sp_data {kpt3d (1,24,3), extrin (V,4,4)};
    src_foreground_mask (1,V,1,H,W) bool.
    """

    def __init__(self, img, cam, feat_geo, feat_tex, sp_data, src_foreground_mask, disable_fg_mask=False, sigma=0.1):
        L = kl.get_library()
        self.img = _dev(img, "img")
        V, C, H, W = self.img.shape
        if C != 3:
            raise ValueError("img must be (V,3,H,W)")
        if not isinstance(feat_geo, (list, tuple)) or len(feat_geo) != 2:
            raise ValueError("feat_geo must be the list [ (V,64,h,w), (V,8,h,w) ] of HGFilterV2")
        self.geo0, self.geo1, self.tex = _dev(feat_geo[0], "feat_geo[0]"), _dev(feat_geo[1], "feat_geo[1]"), _dev(feat_tex, "feat_tex")
        if self.geo0.shape[:2] != (V, 64) or self.geo1.shape[:2] != (V, 8) or self.tex.shape[:2] != (V, 8):
            raise ValueError("feature maps must be (V,64,..), (V,8,..), (V,8,..)")
        self.KRT = _dev(cam["KRT"], "cam['KRT']").reshape(V, 4, 4)
        extrin = sp_data["extrin"] if "extrin" in sp_data else cam["extrin"]
        self.extrin = _dev(extrin, "extrin").reshape(V, 4, 4)
        kpt = _dev(sp_data["kpt3d"], "kpt3d")
        if kpt.numel() != 72:
            raise ValueError("kpt3d must be (1,24,3): batch size 1 and 24 keypoints (reference src/model.py:938, configs/zju.json:44)")
        self.kpt3d = kpt.reshape(24, 3)
        if int(cam["width"]) != W or int(cam["height"]) != H:
            raise ValueError("cam width/height must match the source images")
        if disable_fg_mask:
            self.fg = None
        else:
            m = src_foreground_mask
            if not _on_gpu(m):
                raise RuntimeError("src_foreground_mask must live on the GPU")
            self.fg = (m.reshape(V, H, W) != 0).to(torch.uint8).contiguous()
        d = kl.SceneDesc()
        d.n_views, d.src_h, d.src_w = V, H, W
        d.geo0_h, d.geo0_w = self.geo0.shape[-2:]
        d.geo1_h, d.geo1_w = self.geo1.shape[-2:]
        d.tex_h, d.tex_w = self.tex.shape[-2:]
        d.disable_fg_mask = int(bool(disable_fg_mask))
        d.znear, d.zfar = float(cam["znear"]), float(cam["zfar"])
        d.nml_scale, d.sigma = float(cam.get("nml_scale", 100.0)), float(sigma)
        d.KRT, d.extrin, d.kpt3d = self.KRT.data_ptr(), self.extrin.data_ptr(), self.kpt3d.data_ptr()
        d.img, d.geo0, d.geo1, d.tex = self.img.data_ptr(), self.geo0.data_ptr(), self.geo1.data_ptr(), self.tex.data_ptr()
        d.fg_mask = self.fg.data_ptr() if self.fg is not None else None
        self.desc = d
        self.n_views = V
        nbytes = L.kpn_scene_workspace_bytes(ctypes.byref(d))
        if nbytes == 0:
            raise kl.KpnError(L.kpn_last_error().decode())
        self.ws = torch.empty(nbytes // 4, dtype=_f32, device=self.img.device)
        L.check(L.kpn_scene_prepare(ctypes.byref(d), _p(self.ws), _stream()))

    def as_op_args(self):
        """(scene_ws, scene_dims, scene_scalars) for torch.ops.kpnerf.field_query (keypointnerf_amd/torch_ops.py)."""
        d = self.desc
        return (self.ws, [d.n_views, d.src_h, d.src_w, d.geo0_h, d.geo0_w, d.geo1_h, d.geo1_w, d.tex_h, d.tex_w, d.disable_fg_mask],
                [d.znear, d.zfar, d.nml_scale, d.sigma])


# ------------------------------------------------------------------------------------------------
def ray_bbox_intersection(bounds, orig, direct):
    """bounds (1,2,3), orig (1,1,3), direct (1,R,3) -> near (1,R,1), far (1,R,1), hit (1,R,1) bool."""
    L = kl.get_library()
    b, o, d = _dev(bounds, "bounds").reshape(2, 3), _dev(orig, "orig").reshape(3), _dev(direct, "direct").reshape(-1, 3)
    R = d.shape[0]
    near, far = torch.empty(R, dtype=_f32, device=d.device), torch.empty(R, dtype=_f32, device=d.device)
    hit = torch.empty(R, dtype=torch.uint8, device=d.device)
    L.check(L.kpn_ray_bbox_intersection(_p(b), _p(o), _p(d), R, _p(near), _p(far), _p(hit), _stream()))
    return near.view(1, R, 1), far.view(1, R, 1), hit.view(1, R, 1).bool()


def importance_sample(contrib, z, sample_per_ray, uniform=False, u=None):
    """contrib (B,R,D-2), z (B,R,D-1) -> (B,R,sample_per_ray).  uniform=True uses linspace(0,1,n)
    (reference src/model.py:1126); otherwise `u` (B,R,n) — drawn with torch.rand if not given (:1129)."""
    L = kl.get_library()
    c, zz = _dev(contrib, "contrib"), _dev(z, "z")
    assert c.shape[-1] == zz.shape[-1] - 1  # same assert as the reference, src/model.py:1119
    B, R, Dm2 = c.shape
    n = int(sample_per_ray)
    if uniform:
        uu = None
    else:
        uu = _dev(u, "u").reshape(B * R, n) if u is not None else torch.rand(B * R, n, device=c.device)
    out = torch.empty(B, R, n, dtype=_f32, device=c.device)
    L.check(L.kpn_importance_sample(_p(c), _p(zz), _p(uu), B * R, Dm2, n, _p(out), _stream()))
    return out


class _Rgba2Out(torch.autograd.Function):
    """rgba2out with its hand-written backward (kpn_rgba2out_backward): d_rgba from the upstream gradients of
    color/depth/alpha/sdf; contrib and z carry no gradient, as in the reference (:1038,1118 run under no_grad)."""

    @staticmethod
    def forward(ctx, rgba, z):
        out = _rgba2out_fwd(rgba, z)
        ctx.save_for_backward(rgba.detach(), z.detach())
        ctx.mark_non_differentiable(out[3])
        return out

    @staticmethod
    def backward(ctx, d_color, d_depth, d_alpha, d_contrib, d_sdf):
        L = kl.get_library()
        rgba, z = ctx.saved_tensors
        q, zz = _dev(rgba, "rgba"), _dev(z, "z")
        B, R, S = zz.shape
        g = [None if x is None else _dev(x, "grad") for x in (d_color, d_depth, d_alpha, d_sdf)]
        d_rgba = torch.empty_like(q)
        L.check(L.kpn_rgba2out_backward(_p(q), _p(zz), B * R, S, _p(g[0]), _p(g[1]), _p(g[2]), _p(g[3]), _p(d_rgba), _stream()))
        return d_rgba, None


def rgba2out(rgba, z):
    """rgba (B,R,S,5), z (B,R,S) -> color (B,R,3), depth (B,R), alpha (B,R), contrib (B,R,S), sdf (B,R).
    Differentiable w.r.t. rgba (hand-written backward kernel)."""
    if torch.is_grad_enabled() and isinstance(rgba, torch.Tensor) and rgba.requires_grad:
        return _Rgba2Out.apply(rgba, z)
    return _rgba2out_fwd(rgba, z)


def _rgba2out_fwd(rgba, z):
    L = kl.get_library()
    q, zz = _dev(rgba, "rgba"), _dev(z, "z")
    B, R, S = zz.shape
    if q.shape != (B, R, S, 5):
        raise ValueError("rgba must be (B,R,S,5)")
    dv = q.device
    color = torch.empty(B, R, 3, dtype=_f32, device=dv)
    depth, alpha, sdf = (torch.empty(B, R, dtype=_f32, device=dv) for _ in range(3))
    contrib = torch.empty(B, R, S, dtype=_f32, device=dv)
    L.check(L.kpn_rgba2out(_p(q), _p(zz), B * R, S, _p(color), _p(depth), _p(alpha), _p(contrib), _p(sdf), _stream()))
    return color, depth, alpha, contrib, sdf


def query(scene, weights, pts, view, mode=0):
    """pts (1,N,3), view (1,N,3) -> out (1,N,5), valid (1,N,1) bool.  mode 0 = KeypointNeRF.query's
    [sdf_raw, rad, rgb]; mode 1 = eval_func(query) = [sigma, sdf, rgb] (reference src/model.py:978-997)."""
    L = kl.get_library()
    p, v = _dev(pts, "pts").reshape(-1, 3), _dev(view, "view").reshape(-1, 3)
    if p.shape != v.shape:
        raise ValueError("pts and view must have the same shape")
    N = p.shape[0]
    out = torch.empty(N, 5, dtype=_f32, device=p.device)
    valid = torch.empty(N, dtype=torch.uint8, device=p.device)
    if N == 0:
        return out.view(1, 0, 5), valid.view(1, 0, 1).bool()
    nb = L.kpn_query_workspace_bytes(N, scene.n_views)
    ws = torch.empty(max(nb, 1), dtype=torch.uint8, device=p.device)
    L.check(L.kpn_query(ctypes.byref(scene.desc), _p(scene.ws), _p(weights.tensor), N, _p(p), _p(v), int(mode), _p(out),
                        _p(valid), _p(ws), nb, _stream()))
    return out.view(1, N, 5), valid.view(1, N, 1).bool()


def geo_rows_backward(scene, weights, pts, d_x, keep_mask=0xFFFFFFFF):
    """Reverse pass of the per-(point,view) geometry rows (kpn_geo_rows_backward): MLPUNet.layers1 and the
    feat_geo gathers (reference src/utils.py:691-716, src/model.py:763-765).
    pts (1,N,3) or (N,3); d_x (N,V,64) = d loss / d layers1-output.  Returns (d_plain, d_geo0, d_geo1):
    d_plain flat like weights.flatten_plain (feed it to weights.plain_grads_to_state_dict), d_geo* shaped
    like feat_geo[*] (NCHW views of the channels-last accumulators)."""
    L = kl.get_library()
    p = _dev(pts, "pts").reshape(-1, 3)
    N, V = p.shape[0], scene.n_views
    g = _dev(d_x, "d_x")
    if tuple(g.shape) != (N, V, 64):
        raise ValueError(f"d_x must be (N, V, 64) = {(N, V, 64)}, got {tuple(g.shape)}")
    d = scene.desc
    d_plain = torch.zeros(L.kpn_plain_weight_floats(), dtype=_f32, device=p.device)
    d_g0 = torch.zeros(V, d.geo0_h, d.geo0_w, 64, dtype=_f32, device=p.device)
    d_g1 = torch.zeros(V, d.geo1_h, d.geo1_w, 8, dtype=_f32, device=p.device)
    if N > 0:
        nb = L.kpn_geo_rows_backward_workspace_bytes(N, V)
        ws = torch.empty(max(nb, 1), dtype=torch.uint8, device=p.device)
        L.check(L.kpn_geo_rows_backward(ctypes.byref(d), _p(scene.ws), _p(weights.tensor), N, _p(p), int(keep_mask) & 0xFFFFFFFF,
                                        _p(g), _p(d_plain), _p(d_g0), _p(d_g1), _p(ws), nb, _stream()))
    return d_plain, d_g0.permute(0, 3, 1, 2), d_g1.permute(0, 3, 1, 2)


def query_backward_geometry(scene, weights, pts, d_out, mode=1, keep_mask=0xFFFFFFFF, noise=None, noise_std=0.0):
    """Reverse pass of the field evaluation w.r.t. its two geometry outputs (kpn_query_backward_geometry): pooling,
    layers2, layers1 and the feat_geo gathers.  d_out (N,5) or (1,N,5): columns 0,1 are propagated (mode 1:
    [sigma, sdf] of eval_func, the training path; mode 0: query's raw [sdf_raw, rad]); the colour columns are not yet.
    Returns (d_plain, d_geo0, d_geo1) like geo_rows_backward."""
    L = kl.get_library()
    p = _dev(pts, "pts").reshape(-1, 3)
    N, V = p.shape[0], scene.n_views
    g = _dev(d_out, "d_out").reshape(-1, 5)
    if g.shape[0] != N:
        raise ValueError(f"d_out must have {N} rows")
    nz = None if noise is None else _dev(noise, "noise").reshape(-1)
    if nz is not None and nz.shape[0] != N:
        raise ValueError("noise must have one value per point")
    d = scene.desc
    d_plain = torch.zeros(L.kpn_plain_weight_floats(), dtype=_f32, device=p.device)
    d_g0 = torch.zeros(V, d.geo0_h, d.geo0_w, 64, dtype=_f32, device=p.device)
    d_g1 = torch.zeros(V, d.geo1_h, d.geo1_w, 8, dtype=_f32, device=p.device)
    if N > 0:
        nb = L.kpn_query_backward_geometry_workspace_bytes(N, V)
        ws = torch.empty(max(nb, 1), dtype=torch.uint8, device=p.device)
        L.check(L.kpn_query_backward_geometry(ctypes.byref(d), _p(scene.ws), _p(weights.tensor), N, _p(p), int(mode),
                                              int(keep_mask) & 0xFFFFFFFF, None if nz is None else _p(nz), float(noise_std), _p(g),
                                              _p(d_plain), _p(d_g0), _p(d_g1), _p(ws), nb, _stream()))
    return d_plain, d_g0.permute(0, 3, 1, 2), d_g1.permute(0, 3, 1, 2)


def query_backward(scene, weights, pts, view, d_out, mode=1, keep_mask=0xFFFFFFFF, noise=None, noise_std=0.0):
    """Reverse pass of the whole field evaluation incl. the colour head (kpn_query_backward):
    what loss.backward() does for KeypointNeRF.query + eval_func in training_step (reference src/model.py:128-155).
    pts, view (N,3)/(1,N,3); d_out (N,5) = d loss / d [sigma, sdf, r, g, b] (mode 1) or d [sdf_raw, rad, r, g, b] (mode 0).
    Returns (d_plain, d_geo0, d_geo1, d_tex): flat effective-parameter gradient (weights.plain_grads_to_state_dict maps it
    to weight_g / weight_v / bias / ani_al) and the feature-map gradients shaped like feat_geo[0], feat_geo[1], feat_tex."""
    L = kl.get_library()
    p, vw = _dev(pts, "pts").reshape(-1, 3), _dev(view, "view").reshape(-1, 3)
    N, V = p.shape[0], scene.n_views
    g = _dev(d_out, "d_out").reshape(-1, 5)
    if g.shape[0] != N or vw.shape[0] != N:
        raise ValueError(f"view and d_out must have {N} rows")
    nz = None if noise is None else _dev(noise, "noise").reshape(-1)
    if nz is not None and nz.shape[0] != N:
        raise ValueError("noise must have one value per point")
    d = scene.desc
    d_plain = torch.zeros(L.kpn_plain_weight_floats(), dtype=_f32, device=p.device)
    d_g0 = torch.zeros(V, d.geo0_h, d.geo0_w, 64, dtype=_f32, device=p.device)
    d_g1 = torch.zeros(V, d.geo1_h, d.geo1_w, 8, dtype=_f32, device=p.device)
    d_tx = torch.zeros(V, d.tex_h, d.tex_w, 8, dtype=_f32, device=p.device)
    if N > 0:
        nb = L.kpn_query_backward_workspace_bytes(N, V)
        ws = torch.empty(max(nb, 1), dtype=torch.uint8, device=p.device)
        L.check(L.kpn_query_backward(ctypes.byref(d), _p(scene.ws), _p(weights.tensor), N, _p(p), _p(vw), int(mode),
                                     int(keep_mask) & 0xFFFFFFFF, None if nz is None else _p(nz), float(noise_std), _p(g), _p(d_plain),
                                     _p(d_g0), _p(d_g1), _p(d_tx), _p(ws), nb, _stream()))
    return d_plain, d_g0.permute(0, 3, 1, 2), d_g1.permute(0, 3, 1, 2), d_tx.permute(0, 3, 1, 2)


class RenderPlan:
    """Pre-allocated outputs + workspace for repeated renders of one pixel grid (no per-call allocation)."""

    def __init__(self, scene, grid, n_coarse, n_fine, fine=True, chunk_rays=0, device=None, rows_kernel=None, fuse_kernel=None):
        """rows_kernel / fuse_kernel: the kernels of the calls made with THIS plan ("f32" / "bf16x3" / "f16x2", "f32" / "f16x2";
        None = the process-wide selection of set_geo_rows_mode / set_fuse_mode) — include/kpnerf.h kpn_render_args."""
        L = kl.get_library()
        # grid = (x0, y0, step, nx, ny) as the reference's strided grids, or (x0, y0, step, nx, ny, step_y): rows advance by step_y
        # (a frame's rows dealt round-robin to the ranks of a render job: parallel.rows_of_rank)
        x0, y0, step, nx, ny = (int(g) for g in grid[:5])
        step_y = int(grid[5]) if len(grid) > 5 else 0
        dv = device or scene.ws.device
        self.grid, self.fine = (x0, y0, step, nx, ny, step_y), bool(fine)
        self.out = {"tex_fg": torch.empty(1, 3, ny, nx, dtype=_f32, device=dv), "depth": torch.empty(1, ny, nx, dtype=_f32, device=dv),
                    "alpha": torch.empty(1, ny, nx, dtype=_f32, device=dv)}
        if fine:
            self.out.update({"tex_fg_fine": torch.empty(1, 3, ny, nx, dtype=_f32, device=dv),
                             "depth_fine": torch.empty(1, ny, nx, dtype=_f32, device=dv),
                             "alpha_fine": torch.empty(1, ny, nx, dtype=_f32, device=dv),
                             "sdf": torch.empty(1, ny, nx, dtype=_f32, device=dv)})
        a = kl.RenderArgs()
        a.x0, a.y0, a.step, a.nx, a.ny, a.step_y = x0, y0, step, nx, ny, step_y
        a.n_coarse, a.n_fine, a.fine, a.chunk_rays = int(n_coarse), int(n_fine), int(bool(fine)), int(chunk_rays)
        a.rows_kernel = {None: 0, "f32": 1, "bf16x3": 2, "f16x2": 3}[rows_kernel]
        a.fuse_kernel = {None: 0, "f32": 1, "f16x2": 2}[fuse_kernel]
        for k, v in self.out.items():
            setattr(a, k, v.data_ptr())
        self.args = a
        # K/RT/bounds pointers are filled per call; sizes do not depend on them
        dummy = torch.zeros(16, dtype=_f32, device=dv)
        a.K = a.RT = a.bounds = dummy.data_ptr()
        self.nbytes = L.kpn_render_workspace_bytes(ctypes.byref(scene.desc), ctypes.byref(a))
        if self.nbytes == 0:
            raise kl.KpnError(L.kpn_last_error().decode())
        self.ws = torch.empty(self.nbytes, dtype=torch.uint8, device=dv)

    def n_rays(self):
        return self.grid[3] * self.grid[4]


def render_rays(scene, weights, cam_tar, bounds, grid=None, n_coarse=64, n_fine=64, fine=True, chunk_rays=0, plan=None, stages=False):
    """Eval-mode batch_render_pifu_nerf for the pixel grid (x0, y0, step, nx, ny) of the target camera
    cam_tar {K (1,4,4), RT (1,4,4), znear, zfar}.  Returns the reference's out dict (B=1):
    tex_fg (1,3,ny,nx), depth/alpha (1,ny,nx) [, tex_fg_fine, depth_fine, alpha_fine, sdf].
    stages=True: -> (out, {dirs (R,3), cam_pos (3), z_coarse (R,Sc), rgba_coarse (R,Sc,5) [, z_fine (R,Sc+Sf), rgba_fine (R,Sc+Sf,5)]}), the rays, the per-sample depths
    and eval_func'ed field values of the call in ray order (kpn_render_stages: the conditional parity check of tests/parity_gate.py)."""
    L = kl.get_library()
    if plan is None:
        plan = RenderPlan(scene, grid, n_coarse, n_fine, fine, chunk_rays)
    K, RT, b = _dev(cam_tar["K"], "cam_tar['K']").reshape(4, 4), _dev(cam_tar["RT"], "cam_tar['RT']").reshape(4, 4), _dev(bounds, "bounds").reshape(2, 3)
    a = plan.args
    a.K, a.RT, a.bounds = K.data_ptr(), RT.data_ptr(), b.data_ptr()
    a.znear, a.zfar = float(cam_tar["znear"]), float(cam_tar["zfar"])
    st = None
    if stages:
        R, Sc, Sf = plan.n_rays(), int(a.n_coarse), int(a.n_fine) if plan.fine else 0
        st = {"z_coarse": torch.empty(R, Sc, dtype=_f32, device=plan.ws.device), "rgba_coarse": torch.empty(R, Sc, 5, dtype=_f32, device=plan.ws.device),
              "dirs": torch.empty(R, 3, dtype=_f32, device=plan.ws.device), "cam_pos": torch.empty(3, dtype=_f32, device=plan.ws.device)}
        if plan.fine:
            st.update({"z_fine": torch.empty(R, Sc + Sf, dtype=_f32, device=plan.ws.device),
                       "rgba_fine": torch.empty(R, Sc + Sf, 5, dtype=_f32, device=plan.ws.device)})
        cst = kl.RenderStages()
        for k, v in st.items():
            setattr(cst, k, v.data_ptr())
        a.stages = ctypes.pointer(cst)
    try:
        L.check(L.kpn_render_rays(ctypes.byref(scene.desc), _p(scene.ws), _p(weights.tensor), ctypes.byref(a), _p(plan.ws),
                                  plan.nbytes, _stream()))
    finally:
        if stages:
            a.stages = ctypes.POINTER(kl.RenderStages)()
    plan._keep = (K, RT, b)  # keep the small tensors alive until the stream has consumed them
    return (plan.out, st) if stages else plan.out


def render_rays_train(scene, weights, cam_tar, bounds, pix, u_coarse, u_fine, keep_coarse, keep_fine, noise_coarse=None,
                      noise_fine=None, rand_noise_std=0.0, n_coarse=64, n_fine=64, chunk_rays=0, keep_state=False):
    """TRAIN branch of batch_render_pifu_nerf, forward only, with the random draws passed in (reference
    src/model.py:1008-1017,1049-1053,993-994,742-748,1129): pix (R,2) int32 patch pixels (x,y); u_coarse (R,Sc);
    u_fine (R,Sf); keep_* = (V,) 0/1 view-dropout vectors (or bit masks) of the coarse / fine query; noise_* flat
    density noise.  Returns the out dict with (1,3,R) / (1,R) tensors in patch order (reshape to (out_h,out_w))."""
    L = kl.get_library()
    px = pix.to(torch.int32).contiguous()
    if not _on_gpu(px):
        raise RuntimeError("pix must live on the GPU")
    R = px.shape[0]
    plan = RenderPlan(scene, (0, 0, 1, R, 1), n_coarse, n_fine, fine=True, chunk_rays=chunk_rays)
    K, RT, b = _dev(cam_tar["K"], "cam_tar['K']").reshape(4, 4), _dev(cam_tar["RT"], "cam_tar['RT']").reshape(4, 4), _dev(bounds, "bounds").reshape(2, 3)
    a = plan.args
    a.K, a.RT, a.bounds = K.data_ptr(), RT.data_ptr(), b.data_ptr()
    a.znear, a.zfar = float(cam_tar["znear"]), float(cam_tar["zfar"])
    bits = lambda k: int(k) if isinstance(k, int) else int(sum(1 << i for i, x in enumerate(k.reshape(-1).tolist()) if x > 0.5))
    uc, uf = _dev(u_coarse, "u_coarse").reshape(R, n_coarse), _dev(u_fine, "u_fine").reshape(R, n_fine)
    nc = _dev(noise_coarse, "noise_coarse").reshape(-1) if noise_coarse is not None else None
    nf = _dev(noise_fine, "noise_fine").reshape(-1) if noise_fine is not None else None
    t = kl.TrainArgs()
    t.pix, t.u_coarse, t.u_fine = px.data_ptr(), uc.data_ptr(), uf.data_ptr()
    t.noise_coarse = nc.data_ptr() if nc is not None else None
    t.noise_fine = nf.data_ptr() if nf is not None else None
    t.keep_coarse, t.keep_fine, t.rand_noise_std = bits(keep_coarse), bits(keep_fine), float(rand_noise_std)
    state = None
    if keep_state:
        nb = L.kpn_render_rays_train_state_bytes(ctypes.byref(scene.desc), ctypes.byref(a))
        if nb == 0:
            raise kl.KpnError("bad render arguments: " + L.kpn_last_error().decode())
        state = torch.empty(nb, dtype=torch.uint8, device=px.device)
        L.check(L.kpn_render_rays_train_keep(ctypes.byref(scene.desc), _p(scene.ws), _p(weights.tensor), ctypes.byref(a),
                                             ctypes.byref(t), _p(state), nb, _stream()))
    else:
        L.check(L.kpn_render_rays_train(ctypes.byref(scene.desc), _p(scene.ws), _p(weights.tensor), ctypes.byref(a), ctypes.byref(t),
                                        _p(plan.ws), plan.nbytes, _stream()))
    # no host sync: every launch above is on torch's current stream, and the caching allocator hands a freed block to
    # later work of the SAME stream only, so the argument tensors may be released as soon as this returns
    out = {k: v.reshape(1, *v.shape[1:-2], R) if v.dim() == 4 else v.reshape(1, R) for k, v in plan.out.items()}
    return (out, state) if keep_state else out


def render_rays_train_backward(scene, weights, cam_tar, bounds, pix, u_coarse, u_fine, keep_coarse, keep_fine, grads,
                               noise_coarse=None, noise_fine=None, rand_noise_std=0.0, n_coarse=64, n_fine=64, chunk_rays=0,
                               state=None):
    """loss.backward() through render_rays_train (kpn_render_rays_train_backward): `grads` maps output names
    ('tex_fg', 'depth', 'alpha', 'tex_fg_fine', 'depth_fine', 'alpha_fine', 'sdf') to the gradients of those outputs,
    shaped like them ((1,3,R) / (1,R)); missing keys are zero.  Same other arguments as the forward call.
    `state`: the tensor render_rays_train(..., keep_state=True) returned for the SAME arguments (kpn_render_rays_train_
    backward_kept: the forward is not repeated).  Returns (d_plain, d_geo0, d_geo1, d_tex) as ops.query_backward."""
    L = kl.get_library()
    px = pix.to(torch.int32).contiguous()
    if not _on_gpu(px):
        raise RuntimeError("pix must live on the GPU")
    R, V = px.shape[0], scene.n_views
    K, RT, b = _dev(cam_tar["K"], "cam_tar['K']").reshape(4, 4), _dev(cam_tar["RT"], "cam_tar['RT']").reshape(4, 4), _dev(bounds, "bounds").reshape(2, 3)
    a = kl.RenderArgs()
    a.K, a.RT, a.bounds = K.data_ptr(), RT.data_ptr(), b.data_ptr()
    a.znear, a.zfar = float(cam_tar["znear"]), float(cam_tar["zfar"])
    a.x0, a.y0, a.step, a.nx, a.ny = 0, 0, 1, R, 1
    a.n_coarse, a.n_fine, a.fine, a.chunk_rays = int(n_coarse), int(n_fine), 1, int(chunk_rays)
    bits = lambda k: int(k) if isinstance(k, int) else int(sum(1 << i for i, x in enumerate(k.reshape(-1).tolist()) if x > 0.5))
    uc, uf = _dev(u_coarse, "u_coarse").reshape(R, n_coarse), _dev(u_fine, "u_fine").reshape(R, n_fine)
    nc = _dev(noise_coarse, "noise_coarse").reshape(-1) if noise_coarse is not None else None
    nf = _dev(noise_fine, "noise_fine").reshape(-1) if noise_fine is not None else None
    t = kl.TrainArgs()
    t.pix, t.u_coarse, t.u_fine = px.data_ptr(), uc.data_ptr(), uf.data_ptr()
    t.noise_coarse = nc.data_ptr() if nc is not None else None
    t.noise_fine = nf.data_ptr() if nf is not None else None
    t.keep_coarse, t.keep_fine, t.rand_noise_std = bits(keep_coarse), bits(keep_fine), float(rand_noise_std)
    g, keep_alive = kl.RenderGrads(), []
    for name in ("tex_fg", "depth", "alpha", "tex_fg_fine", "depth_fine", "alpha_fine", "sdf"):
        if grads.get(name) is not None:
            gt = _dev(grads[name], "grads['%s']" % name).reshape(-1)
            if gt.numel() != (3 * R if name.startswith("tex") else R):
                raise ValueError(f"grads['{name}'] has the wrong size")
            keep_alive.append(gt)
            setattr(g, "d_" + name, gt.data_ptr())
    d = scene.desc
    dv = px.device
    d_plain = torch.zeros(L.kpn_plain_weight_floats(), dtype=_f32, device=dv)
    d_g0 = torch.zeros(V, d.geo0_h, d.geo0_w, 64, dtype=_f32, device=dv)
    d_g1 = torch.zeros(V, d.geo1_h, d.geo1_w, 8, dtype=_f32, device=dv)
    d_tx = torch.zeros(V, d.tex_h, d.tex_w, 8, dtype=_f32, device=dv)
    nb = L.kpn_render_rays_train_backward_workspace_bytes(ctypes.byref(d), ctypes.byref(a))
    if nb == 0:
        raise kl.KpnError("bad render arguments: " + L.kpn_last_error().decode())
    ws = torch.empty(nb, dtype=torch.uint8, device=dv)
    if state is not None and state.numel() > 0:
        L.check(L.kpn_render_rays_train_backward_kept(ctypes.byref(d), _p(scene.ws), _p(weights.tensor), ctypes.byref(a), ctypes.byref(t),
                                                      ctypes.byref(g), _p(d_plain), _p(d_g0), _p(d_g1), _p(d_tx), _p(state),
                                                      state.numel(), _p(ws), nb, _stream()))
    else:
        L.check(L.kpn_render_rays_train_backward(ctypes.byref(d), _p(scene.ws), _p(weights.tensor), ctypes.byref(a), ctypes.byref(t),
                                                 ctypes.byref(g), _p(d_plain), _p(d_g0), _p(d_g1), _p(d_tx), _p(ws), nb, _stream()))
    # no host sync (stream-ordered reuse of freed blocks, see render_rays_train)
    return d_plain, d_g0.permute(0, 3, 1, 2), d_g1.permute(0, 3, 1, 2), d_tx.permute(0, 3, 1, 2)


def set_geo_rows_mode(mode):
    """Rows kernel of the field's first MLP (kpn_set_geo_rows_mode): 3 (default) = two fp16 pieces per operand on the fp16 MFMA,
    2 = three bf16 pieces on the bf16 MFMA (fp32's exponent range), both two tiles per wave and one wave per SIMD with fp32-class
    results; 0 = fp32 MFMA.  (1, the earlier one-tile split-bf16 kernel, is not part of the shipped library: DESIGN.md 9.2.)"""
    L = kl.get_library()
    L.check(L.kpn_set_geo_rows_mode(int(mode)))


def set_fuse_mode(mode):
    """Per-point kernel (kpn_set_fuse_mode): 1 (default) = weights as two fp16 pieces on the fp16 MFMA; 0 = fp32 MFMA."""
    L = kl.get_library()
    L.check(L.kpn_set_fuse_mode(int(mode)))


def get_fuse_mode():
    return int(kl.get_library().kpn_get_fuse_mode())


def set_density_first(mode):
    """Render passes: density of every point in the hull first, colour head for the points with relu(rad) > 0 only
    (kpn_set_density_first, include/kpnerf.h): 0 = never, 1 = always, 2 = auto (default); frames bit-identical in every mode."""
    kl.get_library().check(kl.get_library().kpn_set_density_first(int(mode)))


def get_density_first():
    return int(kl.get_library().kpn_get_density_first())


def set_range_guard(on):
    """The range guard of the two-fp16-piece kernels (include/kpnerf.h kpn_set_range_guard): on by default."""
    kl.get_library().check(kl.get_library().kpn_set_range_guard(int(bool(on))))


def range_guard_count():
    """Batches of rows the fp32-range kernels evaluated again since the library was loaded (synchronises the current stream):
    0 = every pass ran on the default (two-fp16-piece) kernels."""
    L = kl.get_library()
    n = ctypes.c_int64(0)
    L.check(L.kpn_range_guard_count(_stream(), ctypes.byref(n)))
    return int(n.value)


def packed_f16_range_check(packed):
    """Number of packed layers1 weights that fp16 cannot hold (rows mode 3 needs 0); synchronises the current stream."""
    import ctypes
    L = kl.get_library()
    beyond = ctypes.c_int32(-1)
    L.check(L.kpn_packed_f16_range_check(packed.data_ptr(), _stream(), ctypes.byref(beyond)))
    return int(beyond.value)


def get_geo_rows_mode():
    return int(kl.get_library().kpn_get_geo_rows_mode())


def frame_to_rgb8(img, bgr=False):
    """(3,H,W) or (1,3,H,W) fp32 -> (H,W,3) uint8 on the device: clamp to [0,1] (_arrange_nerf_images, reference
    src/model.py:427-430), x255 and truncate (`.astype(np.uint8)`, :496), optional B,G,R order for cv2.imwrite (:222)."""
    L = kl.get_library()
    x = _dev(img, "img")
    x = x.reshape(3, *x.shape[-2:])
    H, W = x.shape[-2:]
    out = torch.empty(H, W, 3, dtype=torch.uint8, device=x.device)
    L.check(L.kpn_frame_to_rgb8(_p(x), H, W, int(bool(bgr)), _p(out), _stream()))
    return out


def mse_psnr(pred, gt):
    """ZJUEvaluator.compute_score's mse and psnr (reference src/zju_evaluator.py:16-19,63-64) without leaving the
    device: returns a (2,) float64 tensor [mse, psnr]."""
    L = kl.get_library()
    a, b = _dev(pred, "pred"), _dev(gt, "gt")
    if a.shape != b.shape:
        raise ValueError("pred and gt must have the same shape")
    out = torch.empty(2, dtype=torch.float64, device=a.device)
    scratch = torch.empty(2048 * 8 + 8, dtype=torch.uint8, device=a.device)
    L.check(L.kpn_mse_psnr(_p(a), _p(b), a.numel(), _p(out), _p(scratch), _stream()))
    return out


def pix_l1_loss(src, tar, lam, want_grad=True):
    """pix_loss(src, tar, {"l1": lam})["l1"] of the reference (src/utils.py:164-168) and d loss / d src: returns
    (loss: 0-dim device tensor, d_src: tensor like src or None)."""
    L = kl.get_library()
    a, b = _dev(src, "src"), _dev(tar, "tar")
    if a.shape != b.shape:
        raise ValueError("src and tar must have the same shape")
    loss = torch.empty(1, dtype=_f32, device=a.device)
    d = torch.empty_like(a) if want_grad else None
    scratch = torch.empty(2048 * 8 + 8, dtype=torch.uint8, device=a.device)
    L.check(L.kpn_pix_l1_loss(_p(a), _p(b), a.numel(), float(lam), _p(loss), _p(d), _p(scratch), _stream()))
    return loss.reshape(()), d


def ssim(pred, gt, mask_at_box=None):
    """SSIM of ZJUEvaluator._compute_ssim (reference src/zju_evaluator.py:21-45): pred, gt (3,H,W) or (1,3,H,W) in [0,1];
    mask_at_box (H,W): the images are cropped to its bounding rectangle (cv2.boundingRect) first.  Returns a float."""
    L = kl.get_library()
    a, b = _dev(pred, "pred").reshape(3, *pred.shape[-2:]), _dev(gt, "gt").reshape(3, *gt.shape[-2:])
    H, W = a.shape[-2:]
    x0, y0, w, h = 0, 0, W, H
    if mask_at_box is not None:
        ys, xs = torch.nonzero(mask_at_box.reshape(H, W) != 0, as_tuple=True)
        if ys.numel() == 0:
            raise ValueError("empty mask")
        x0, y0 = int(xs.min()), int(ys.min())
        w, h = int(xs.max()) - x0 + 1, int(ys.max()) - y0 + 1
    nb = L.kpn_ssim_scratch_bytes(w, h)
    if nb == 0:
        raise ValueError("crop smaller than the 7x7 SSIM window")
    scratch = torch.empty(nb, dtype=torch.uint8, device=a.device)
    out = torch.empty(1, dtype=torch.float64, device=a.device)
    L.check(L.kpn_ssim(_p(a), _p(b), H, W, x0, y0, w, h, _p(out), _p(scratch), _stream()))
    return float(out.item())


def selftest_mfma():
    """Checks on the device that v_mfma_f32_32x32x2_f32 has the operand/result lane maps the kernels assume."""
    L = kl.get_library()
    scratch = torch.zeros(65536, dtype=_f32, device="cuda")
    err = ctypes.c_float(0.0)
    torch.cuda.synchronize()
    rc = L.kpn_selftest_mfma(_p(scratch), _stream(), ctypes.byref(err))
    if rc != 0:
        print("selftest_mfma:", L.kpn_last_error().decode())
    return rc, err.value
