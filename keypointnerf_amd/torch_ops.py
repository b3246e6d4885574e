"""PyTorch custom operators (``torch.ops.kpnerf.*``) over the gfx950 library.

BASELINE.json's north star asks for the kernels to be "exposed to Python through PyTorch-ROCm custom ops";
these are thin ``torch.library.custom_op`` registrations (device type "cuda" = HIP on ROCm) around
``keypointnerf_amd.ops`` with shape-only fake implementations, so the ops compose with torch.compile /
FakeTensor tracing and show up in profiler traces under their own names.  There is no CPU kernel registered:
calling them on CPU tensors raises NotImplementedError from the dispatcher.

    torch.ops.kpnerf.rgba2out(rgba, z)                       -> (color, depth, alpha, contrib, sdf)
    torch.ops.kpnerf.importance_sample(contrib, z, n, u?)    -> samples           (u=None: uniform linspace)
    torch.ops.kpnerf.ray_bbox_intersection(bounds, orig, d)  -> (near, far, hit)
    torch.ops.kpnerf.field_query(scene_ws, scene_dims, scene_scalars, weights, pts, view, mode) -> (out, valid)

    torch.ops.kpnerf.render_rays(scene_ws, scene_dims, scene_scalars, weights, K, RT, bounds, znear, zfar, grid,
                                 n_coarse, n_fine, fine) -> (tex_fg, depth, alpha, tex_fg_fine, depth_fine, alpha_fine, sdf)
    torch.ops.kpnerf.render_rays_train(plain, geo0, geo1, tex, img, KRT, extrin, kpt3d, fg_mask?, scene_scalars, K, RT,
                                       bounds, znear, zfar, pix, u_c, u_f, noise_c?, noise_f?, keep_c, keep_f, noise_std,
                                       n_coarse, n_fine) -> the same seven outputs, DIFFERENTIABLE

``scene_ws / scene_dims / scene_scalars`` come from ``ops.PreparedScene.as_op_args()``.

    torch.ops.kpnerf.pix_l1_loss(src, tar, lam) -> (loss, d loss / d src)   the L1 terms of the training loss, DIFFERENTIABLE

``rgba2out`` and ``render_rays_train`` carry ``register_autograd`` formulas whose backward is itself a registered op
(``kpnerf::rgba2out_backward``, ``kpnerf::render_rays_train_backward`` = kpn_render_rays_train_backward): gradients reach
the flat effective-parameter vector ``plain`` (and from there ``weight_g`` / ``weight_v`` / ``bias`` / ``ani_al`` through
``weights.plain_tensor_from_module``) and the three encoder feature maps.  This is the op the training drop-in calls.
"""
import ctypes
from typing import List, Optional, Tuple

import torch

from . import lib as kl
from . import ops

_lib = torch.library


@_lib.custom_op("kpnerf::rgba2out", mutates_args=(), device_types="cuda")
def rgba2out(rgba: torch.Tensor, z: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    return ops.rgba2out(rgba, z)


@rgba2out.register_fake
def _(rgba, z):
    B, R, S = z.shape
    f = lambda *s: rgba.new_empty(s)
    return f(B, R, 3), f(B, R), f(B, R), f(B, R, S), f(B, R)


@_lib.custom_op("kpnerf::rgba2out_backward", mutates_args=(), device_types="cuda")
def rgba2out_backward(rgba: torch.Tensor, z: torch.Tensor, d_color: Optional[torch.Tensor], d_depth: Optional[torch.Tensor],
                      d_alpha: Optional[torch.Tensor], d_sdf: Optional[torch.Tensor]) -> torch.Tensor:
    L = kl.get_library()
    q, zz = ops._dev(rgba, "rgba"), ops._dev(z, "z")
    B, R, S = zz.shape
    g = [None if x is None else ops._dev(x, "grad") for x in (d_color, d_depth, d_alpha, d_sdf)]
    d_rgba = torch.empty_like(q)
    L.check(L.kpn_rgba2out_backward(ops._p(q), ops._p(zz), B * R, S, ops._p(g[0]), ops._p(g[1]), ops._p(g[2]), ops._p(g[3]),
                                    ops._p(d_rgba), ops._stream()))
    return d_rgba


@rgba2out_backward.register_fake
def _(rgba, z, d_color, d_depth, d_alpha, d_sdf):
    return torch.empty_like(rgba)


def _rgba2out_setup(ctx, inputs, output):
    ctx.save_for_backward(*inputs)
    ctx.set_materialize_grads(False)


def _rgba2out_bwd(ctx, d_color, d_depth, d_alpha, d_contrib, d_sdf):
    # contrib and z carry no gradient in the reference (the sampler runs under no_grad, src/model.py:1038,1118)
    rgba, z = ctx.saved_tensors
    return torch.ops.kpnerf.rgba2out_backward(rgba, z, d_color, d_depth, d_alpha, d_sdf), None


rgba2out.register_autograd(_rgba2out_bwd, setup_context=_rgba2out_setup)


@_lib.custom_op("kpnerf::importance_sample", mutates_args=(), device_types="cuda")
def importance_sample(contrib: torch.Tensor, z: torch.Tensor, n: int, u: Optional[torch.Tensor] = None) -> torch.Tensor:
    return ops.importance_sample(contrib, z, n, uniform=u is None, u=u)


@importance_sample.register_fake
def _(contrib, z, n, u=None):
    return contrib.new_empty(contrib.shape[0], contrib.shape[1], n)


@_lib.custom_op("kpnerf::ray_bbox_intersection", mutates_args=(), device_types="cuda")
def ray_bbox_intersection(bounds: torch.Tensor, orig: torch.Tensor, direct: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    return ops.ray_bbox_intersection(bounds, orig, direct)


@ray_bbox_intersection.register_fake
def _(bounds, orig, direct):
    R = direct.shape[-2]
    return direct.new_empty(1, R, 1), direct.new_empty(1, R, 1), direct.new_empty(1, R, 1, dtype=torch.bool)


class _SceneView:
    """A PreparedScene rebuilt from op arguments (workspace tensor + plain ints/floats)."""

    def __init__(self, ws, dims, scalars):
        d = kl.SceneDesc()
        (d.n_views, d.src_h, d.src_w, d.geo0_h, d.geo0_w, d.geo1_h, d.geo1_w, d.tex_h, d.tex_w, d.disable_fg_mask) = dims
        d.znear, d.zfar, d.nml_scale, d.sigma = scalars
        # the raw NCHW inputs are only read by kpn_scene_prepare; the query reads the prepared workspace
        for k in ("KRT", "extrin", "kpt3d", "img", "fg_mask", "geo0", "geo1", "tex"):
            setattr(d, k, ws.data_ptr())
        self.desc, self.ws, self.n_views = d, ws, dims[0]


@_lib.custom_op("kpnerf::field_query", mutates_args=(), device_types="cuda")
def field_query(scene_ws: torch.Tensor, scene_dims: List[int], scene_scalars: List[float], weights: torch.Tensor,
                pts: torch.Tensor, view: torch.Tensor, mode: int) -> Tuple[torch.Tensor, torch.Tensor]:
    w = ops.PackedWeights.__new__(ops.PackedWeights)
    w.tensor = weights
    return ops.query(_SceneView(scene_ws, list(scene_dims), list(scene_scalars)), w, pts, view, mode=mode)


@field_query.register_fake
def _(scene_ws, scene_dims, scene_scalars, weights, pts, view, mode):
    N = pts.shape[-2]
    return pts.new_empty(1, N, 5), pts.new_empty(1, N, 1, dtype=torch.bool)


_OUT7 = Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]
_OUT8 = Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]
_OUT_KEYS = ("tex_fg", "depth", "alpha", "tex_fg_fine", "depth_fine", "alpha_fine", "sdf")


@_lib.custom_op("kpnerf::render_rays", mutates_args=(), device_types="cuda")
def render_rays(scene_ws: torch.Tensor, scene_dims: List[int], scene_scalars: List[float], weights: torch.Tensor,
                K: torch.Tensor, RT: torch.Tensor, bounds: torch.Tensor, znear: float, zfar: float, grid: List[int],
                n_coarse: int, n_fine: int, fine: bool) -> _OUT7:
    """Eval branch of batch_render_pifu_nerf (kpn_render_rays) for the pixel grid (x0, y0, step, nx, ny).  With
    fine=False the four fine outputs are empty tensors."""
    w = ops.PackedWeights.__new__(ops.PackedWeights)
    w.tensor = weights
    sv = _SceneView(scene_ws, list(scene_dims), list(scene_scalars))
    out = ops.render_rays(sv, w, {"K": K, "RT": RT, "znear": znear, "zfar": zfar}, bounds, grid=tuple(grid), n_coarse=n_coarse,
                          n_fine=n_fine, fine=fine)
    return tuple(out[k] if k in out else scene_ws.new_empty(0) for k in _OUT_KEYS)


@render_rays.register_fake
def _(scene_ws, scene_dims, scene_scalars, weights, K, RT, bounds, znear, zfar, grid, n_coarse, n_fine, fine):
    nx, ny = grid[3], grid[4]
    f = lambda *s: scene_ws.new_empty(s)
    e = lambda *s: f(*s) if fine else f(0)
    return f(1, 3, ny, nx), f(1, ny, nx), f(1, ny, nx), e(1, 3, ny, nx), e(1, ny, nx), e(1, ny, nx), e(1, ny, nx)


def _raw_scene(geo0, geo1, tex, img, KRT, extrin, kpt3d, fg_mask, scal):
    V, _, H, W = img.shape
    cam = {"KRT": KRT, "width": W, "height": H, "znear": scal[0], "zfar": scal[1], "nml_scale": scal[2]}
    return ops.PreparedScene(img, cam, [geo0, geo1], tex, {"kpt3d": kpt3d, "extrin": extrin}, fg_mask,
                             disable_fg_mask=fg_mask is None, sigma=scal[3])


# One training iteration calls the forward op and then the backward op with the SAME parameters and maps: the packed weights
# (weight-norm fold + operand order + fp16 / bf16 streams) and the prepared scene (NCHW -> NHWC of every map) of the forward call
# are kept for the backward call instead of being built twice.  Autograd hands the backward op NEW tensor objects over the same
# storage, so the key is (storage address, version counter, shape, device) per tensor — and the entry HOLDS the forward call's
# input tensors: while it lives their storage cannot be freed, so an equal address means the same storage, and an in-place change
# in between moves the version counter: a miss, never a stale hit.  The backward op drops the entry when it is done: nothing of
# an iteration stays resident after it.
# ONE entry for the process, behind a lock (round 5; an advisor finding): for CUDA tensors autograd runs the backward op on its
# device worker thread, not on the thread that ran the forward — a thread-local entry was never found by the backward (which then
# built the scene and the packed weights a second time, the very work the cache exists to avoid) and never cleared on the forward's
# thread.  Two threads training different models in one process take turns at the single entry: a miss, never a wrong hit.
import threading


class _IterCache:
    lock = threading.Lock()
    key = scene = w = pinned = None
    hits = misses = 0      # observable by the tests


def _tensor_key(t):
    if t is None:
        return None
    return (t.data_ptr(), t._version if not t.is_inference() else -1, tuple(t.shape), str(t.device), t.dtype)


def _iter_cache_clear():
    with _IterCache.lock:
        _IterCache.key = _IterCache.scene = _IterCache.w = _IterCache.pinned = None


def _scene_and_weights(plain, geo0, geo1, tex, img, KRT, extrin, kpt3d, fg_mask, scal):
    tensors = (plain, geo0, geo1, tex, img, KRT, extrin, kpt3d, fg_mask)
    key = tuple(_tensor_key(t) for t in tensors) + (tuple(scal),)
    with _IterCache.lock:
        if not any(t is not None and t.is_inference() for t in tensors) and key == _IterCache.key:
            _IterCache.hits += 1
            return _IterCache.scene, _IterCache.w
    scene = _raw_scene(geo0, geo1, tex, img, KRT, extrin, kpt3d, fg_mask, scal)
    w = ops.PackedWeights.from_plain(plain, device=geo0.device)
    with _IterCache.lock:
        _IterCache.misses += 1
        _IterCache.key, _IterCache.scene, _IterCache.w, _IterCache.pinned = key, scene, w, tensors
    return scene, w


@_lib.custom_op("kpnerf::render_rays_train", mutates_args=(), device_types="cuda")
def render_rays_train(plain: torch.Tensor, geo0: torch.Tensor, geo1: torch.Tensor, tex: torch.Tensor, img: torch.Tensor,
                      KRT: torch.Tensor, extrin: torch.Tensor, kpt3d: torch.Tensor, fg_mask: Optional[torch.Tensor],
                      scene_scalars: List[float], K: torch.Tensor, RT: torch.Tensor, bounds: torch.Tensor, znear: float,
                      zfar: float, pix: torch.Tensor, u_c: torch.Tensor, u_f: torch.Tensor, noise_c: Optional[torch.Tensor],
                      noise_f: Optional[torch.Tensor], keep_c: int, keep_f: int, noise_std: float, n_coarse: int,
                      n_fine: int, keep_state: bool = False) -> _OUT8:
    """The stochastic (`uniform=False`) branch of batch_render_pifu_nerf with every draw an input (kpn_render_rays_train).
    plain: flat effective parameters (weights.flatten_plain layout); geo0/geo1/tex: the encoders' NCHW maps; fg_mask None =
    disable_fg_mask; scene_scalars = [znear, zfar, nml_scale, sigma] of the SOURCE cameras / spatial encoder.
    Outputs: (1,3,R) / (1,R) tensors in the order of `pix`, then the pass state (uint8; empty unless keep_state): with it
    the backward op starts from the forward's rays, depths, field values, valid lists and rows instead of repeating the
    forward (kpn_render_rays_train_keep / kpn_render_rays_train_backward_kept)."""
    scene, w = _scene_and_weights(plain, geo0, geo1, tex, img, KRT, extrin, kpt3d, fg_mask, list(scene_scalars))
    res = ops.render_rays_train(scene, w, {"K": K, "RT": RT, "znear": znear, "zfar": zfar}, bounds, pix, u_c, u_f, keep_c, keep_f,
                                noise_coarse=noise_c, noise_fine=noise_f, rand_noise_std=noise_std, n_coarse=n_coarse,
                                n_fine=n_fine, keep_state=keep_state)
    out, state = res if keep_state else (res, torch.empty(0, dtype=torch.uint8, device=geo0.device))
    return tuple(out[k].clone() for k in _OUT_KEYS) + (state,)


@render_rays_train.register_fake
def _(plain, geo0, geo1, tex, img, KRT, extrin, kpt3d, fg_mask, scene_scalars, K, RT, bounds, znear, zfar, pix, u_c, u_f, noise_c,
      noise_f, keep_c, keep_f, noise_std, n_coarse, n_fine, keep_state=False):
    R = pix.shape[0]
    f = lambda *s: geo0.new_empty(s)
    return f(1, 3, R), f(1, R), f(1, R), f(1, 3, R), f(1, R), f(1, R), f(1, R), geo0.new_empty(0, dtype=torch.uint8)


@_lib.custom_op("kpnerf::render_rays_train_backward", mutates_args=(), device_types="cuda")
def render_rays_train_backward(plain: torch.Tensor, geo0: torch.Tensor, geo1: torch.Tensor, tex: torch.Tensor, img: torch.Tensor,
                               KRT: torch.Tensor, extrin: torch.Tensor, kpt3d: torch.Tensor, fg_mask: Optional[torch.Tensor],
                               scene_scalars: List[float], K: torch.Tensor, RT: torch.Tensor, bounds: torch.Tensor,
                               znear: float, zfar: float, pix: torch.Tensor, u_c: torch.Tensor, u_f: torch.Tensor,
                               noise_c: Optional[torch.Tensor], noise_f: Optional[torch.Tensor], keep_c: int, keep_f: int,
                               noise_std: float, n_coarse: int, n_fine: int, d_tex_fg: Optional[torch.Tensor],
                               d_depth: Optional[torch.Tensor], d_alpha: Optional[torch.Tensor],
                               d_tex_fg_fine: Optional[torch.Tensor], d_depth_fine: Optional[torch.Tensor],
                               d_alpha_fine: Optional[torch.Tensor], d_sdf: Optional[torch.Tensor],
                               state: Optional[torch.Tensor] = None
                               ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """loss.backward() through render_rays_train (kpn_render_rays_train_backward) -> (d_plain, d_geo0, d_geo1, d_tex),
    the map gradients NCHW like the maps."""
    scene, w = _scene_and_weights(plain, geo0, geo1, tex, img, KRT, extrin, kpt3d, fg_mask, list(scene_scalars))
    grads = dict(zip(_OUT_KEYS, (d_tex_fg, d_depth, d_alpha, d_tex_fg_fine, d_depth_fine, d_alpha_fine, d_sdf)))
    d_plain, d_g0, d_g1, d_tx = ops.render_rays_train_backward(
        scene, w, {"K": K, "RT": RT, "znear": znear, "zfar": zfar}, bounds, pix, u_c, u_f, keep_c, keep_f, grads,
        noise_coarse=noise_c, noise_fine=noise_f, rand_noise_std=noise_std, n_coarse=n_coarse, n_fine=n_fine, state=state)
    _iter_cache_clear()   # stream-ordered: the launches above hold nothing but device pointers the allocator keeps valid for them
    return d_plain, d_g0.contiguous(), d_g1.contiguous(), d_tx.contiguous()


@render_rays_train_backward.register_fake
def _(plain, geo0, geo1, tex, *rest):
    return torch.empty_like(plain), torch.empty_like(geo0), torch.empty_like(geo1), torch.empty_like(tex)


def _train_setup(ctx, inputs, output):
    # tensors go through save_for_backward, so that autograd's version-counter check raises if one of them (feature maps, the
    # flat parameters, the draws) or the kept pass state is modified in place between forward and backward — the state is only
    # valid for the values it was computed from; scalars stay on ctx
    args = list(inputs[:25])
    ctx.tensor_slots = [i for i, a in enumerate(args) if isinstance(a, torch.Tensor)]
    ctx.scalars = [None if isinstance(a, torch.Tensor) else a for a in args]
    ctx.save_for_backward(*[args[i] for i in ctx.tensor_slots], output[7])   # ... and the pass state (empty unless keep_state)
    ctx.n_inputs = len(inputs)
    ctx.set_materialize_grads(False)


def _train_bwd(ctx, *grads):
    saved = ctx.saved_tensors
    args = list(ctx.scalars)
    for i, t in zip(ctx.tensor_slots, saved[:-1]):
        args[i] = t
    state = saved[-1] if saved[-1].numel() > 0 else None
    d_plain, d_g0, d_g1, d_tx = torch.ops.kpnerf.render_rays_train_backward(
        *args, *[None if g is None else g.contiguous() for g in grads[:7]], state)
    return (d_plain, d_g0, d_g1, d_tx) + (None,) * (ctx.n_inputs - 4)


render_rays_train.register_autograd(_train_bwd, setup_context=_train_setup)


@_lib.custom_op("kpnerf::pix_l1_loss", mutates_args=(), device_types="cuda")
def pix_l1_loss(src: torch.Tensor, tar: torch.Tensor, lam: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """(lam * mean|src - tar|, d loss / d src) = the L1 term of the reference's pix_loss (src/utils.py:164-168) and the seed
    gradient autograd derives for it (kpn_pix_l1_loss).  Differentiable w.r.t. src."""
    loss, d = ops.pix_l1_loss(src, tar, lam, want_grad=True)
    return loss, d.reshape(src.shape)


@pix_l1_loss.register_fake
def _(src, tar, lam):
    return src.new_empty(()), torch.empty_like(src)


def _l1_setup(ctx, inputs, output):
    ctx.save_for_backward(output[1])


def _l1_bwd(ctx, d_loss, _d_grad):
    (g,) = ctx.saved_tensors
    return g * d_loss, None, None


pix_l1_loss.register_autograd(_l1_bwd, setup_context=_l1_setup)
