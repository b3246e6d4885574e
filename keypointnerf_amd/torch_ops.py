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

``scene_ws / scene_dims / scene_scalars`` come from ``ops.PreparedScene.as_op_args()``.
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
