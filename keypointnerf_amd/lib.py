"""ctypes binding of the C ABI in include/kpnerf.h — the stub a maintainer of the (pure-Python)
reference would add to call the gfx950 library.

The product library is ``keypointnerf_amd/_lib/libkpnerf_hip.so`` (built by ``build.py`` /
``__graft_entry__.build()``, hipcc --offload-arch=gfx950).  There is NO fallback: if it is missing
or cannot be loaded, ``get_library()`` raises.  ``KpnLibrary(path)`` can bind any library exporting
the same ABI; the CPU test-suite uses that to drive the host SIMT-emulator build of the very same
kernel sources (tests/simt) with numpy buffers — never through this module's default path.
"""
import ctypes
import os

# torch first: PyTorch-ROCm ships its own libamdhip64.so.7.  The library must bind to THE SAME HIP
# runtime instance as torch (device pointers and streams are shared), which the dynamic loader
# guarantees by SONAME once torch's copy is already in the process.  Loading libkpnerf_hip.so first
# would pull /opt/rocm's runtime in as a second instance ("no ROCm-capable device is detected").
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "_lib", "libkpnerf_hip.so")

c_f = ctypes.c_float
c_i32 = ctypes.c_int32
c_i64 = ctypes.c_int64
c_p = ctypes.c_void_p
c_sz = ctypes.c_size_t


class SceneDesc(ctypes.Structure):
    """struct kpn_scene_desc"""
    _fields_ = [(n, c_i32) for n in ("n_views", "src_h", "src_w", "geo0_h", "geo0_w", "geo1_h", "geo1_w", "tex_h",
                                     "tex_w", "disable_fg_mask")] + \
               [(n, c_f) for n in ("znear", "zfar", "nml_scale", "sigma")] + \
               [(n, c_p) for n in ("KRT", "extrin", "kpt3d", "img", "fg_mask", "geo0", "geo1", "tex")]


class RenderStages(ctypes.Structure):
    """struct kpn_render_stages"""
    _fields_ = [(n, c_p) for n in ("z_coarse", "rgba_coarse", "z_fine", "rgba_fine", "dirs", "cam_pos")]


class RenderArgs(ctypes.Structure):
    """struct kpn_render_args"""
    _fields_ = [(n, c_p) for n in ("K", "RT", "bounds")] + [("znear", c_f), ("zfar", c_f)] + \
               [(n, c_i32) for n in ("x0", "y0", "step", "nx", "ny", "n_coarse", "n_fine", "fine", "chunk_rays")] + \
               [(n, c_p) for n in ("tex_fg", "depth", "alpha", "tex_fg_fine", "depth_fine", "alpha_fine", "sdf")] + \
               [("step_y", c_i32), ("rows_kernel", c_i32), ("fuse_kernel", c_i32), ("stages", ctypes.POINTER(RenderStages))]


class TrainArgs(ctypes.Structure):
    """struct kpn_train_args"""
    _fields_ = [(n, c_p) for n in ("pix", "u_coarse", "noise_coarse", "noise_fine", "u_fine")] + \
               [("keep_coarse", ctypes.c_uint32), ("keep_fine", ctypes.c_uint32), ("rand_noise_std", c_f)]


class RenderGrads(ctypes.Structure):
    """struct kpn_render_grads"""
    _fields_ = [(n, c_p) for n in ("d_tex_fg", "d_depth", "d_alpha", "d_tex_fg_fine", "d_depth_fine", "d_alpha_fine", "d_sdf")]


# name -> (restype, argtypes); mirrors include/kpnerf.h one to one
_SIGNATURES = {
    "kpn_abi_version": (ctypes.c_int, []),
    "kpn_last_error": (ctypes.c_char_p, []),
    "kpn_is_device_build": (ctypes.c_int, []),
    "kpn_plain_weight_floats": (c_sz, []),
    "kpn_packed_weight_floats": (c_sz, []),
    "kpn_pack_weights": (ctypes.c_int, [c_p, c_p]),
    "kpn_pack_weights_device": (ctypes.c_int, [c_p, c_p, c_p]),
    "kpn_scene_workspace_bytes": (c_sz, [ctypes.POINTER(SceneDesc)]),
    "kpn_scene_prepare": (ctypes.c_int, [ctypes.POINTER(SceneDesc), c_p, c_p]),
    "kpn_ray_bbox_intersection": (ctypes.c_int, [c_p, c_p, c_p, c_i64, c_p, c_p, c_p, c_p]),
    "kpn_make_rays": (ctypes.c_int, [c_p, c_p, c_f, c_f, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_p, c_p, c_p, c_p, c_p]),
    "kpn_importance_sample": (ctypes.c_int, [c_p, c_p, c_p, c_i64, c_i32, c_i32, c_p, c_p]),
    "kpn_rgba2out": (ctypes.c_int, [c_p, c_p, c_i64, c_i32, c_p, c_p, c_p, c_p, c_p, c_p]),
    "kpn_rgba2out_backward": (ctypes.c_int, [c_p, c_p, c_i64, c_i32, c_p, c_p, c_p, c_p, c_p, c_p]),
    "kpn_geo_rows_backward_workspace_bytes": (c_sz, [c_i64, c_i32]),
    "kpn_geo_rows_backward": (ctypes.c_int, [ctypes.POINTER(SceneDesc), c_p, c_p, c_i64, c_p, ctypes.c_uint32, c_p, c_p, c_p,
                                             c_p, c_p, c_sz, c_p]),
    "kpn_query_backward_geometry_workspace_bytes": (c_sz, [c_i64, c_i32]),
    "kpn_query_backward_geometry": (ctypes.c_int, [ctypes.POINTER(SceneDesc), c_p, c_p, c_i64, c_p, c_i32, ctypes.c_uint32, c_p,
                                                   ctypes.c_float, c_p, c_p, c_p, c_p, c_p, c_sz, c_p]),
    "kpn_query_backward_workspace_bytes": (c_sz, [c_i64, c_i32]),
    "kpn_query_backward": (ctypes.c_int, [ctypes.POINTER(SceneDesc), c_p, c_p, c_i64, c_p, c_p, c_i32, ctypes.c_uint32, c_p,
                                          ctypes.c_float, c_p, c_p, c_p, c_p, c_p, c_p, c_sz, c_p]),
    "kpn_render_rays_train_backward_workspace_bytes": (c_sz, [ctypes.POINTER(SceneDesc), ctypes.POINTER(RenderArgs)]),
    "kpn_render_rays_train_backward": (ctypes.c_int, [ctypes.POINTER(SceneDesc), c_p, c_p, ctypes.POINTER(RenderArgs),
                                                      ctypes.POINTER(TrainArgs), ctypes.POINTER(RenderGrads), c_p, c_p, c_p, c_p,
                                                      c_p, c_sz, c_p]),
    "kpn_set_geo_rows_mode": (ctypes.c_int, [c_i32]),
    "kpn_get_geo_rows_mode": (ctypes.c_int, []),
    "kpn_packed_f16_range_check": (ctypes.c_int, [c_p, c_p, c_p]),
    "kpn_set_fuse_mode": (ctypes.c_int, [c_i32]),
    "kpn_get_fuse_mode": (ctypes.c_int, []),
    "kpn_set_density_first": (ctypes.c_int, [c_i32]),
    "kpn_get_density_first": (ctypes.c_int, []),
    "kpn_density_stats": (ctypes.c_int, [c_p, c_p, c_p, c_i32]),
    "kpn_density_first_passes": (ctypes.c_int, [c_p, c_p, c_i32]),
    "kpn_bwd_profile_enable": (ctypes.c_int, [c_i32]),
    "kpn_bwd_profile_collect": (ctypes.c_int, [c_p, c_p, c_p, c_p, c_p]),
    "kpn_set_range_guard": (ctypes.c_int, [c_i32]),
    "kpn_get_range_guard": (ctypes.c_int, []),
    "kpn_range_guard_count": (ctypes.c_int, [c_p, c_p]),
    "kpn_ssim_scratch_bytes": (c_sz, [c_i32, c_i32]),
    "kpn_ssim": (ctypes.c_int, [c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_p, c_p, c_p]),
    "kpn_query_workspace_bytes": (c_sz, [c_i64, c_i32]),
    "kpn_query": (ctypes.c_int, [ctypes.POINTER(SceneDesc), c_p, c_p, c_i64, c_p, c_p, c_i32, c_p, c_p, c_p, c_sz, c_p]),
    "kpn_render_workspace_bytes": (c_sz, [ctypes.POINTER(SceneDesc), ctypes.POINTER(RenderArgs)]),
    "kpn_render_rays": (ctypes.c_int, [ctypes.POINTER(SceneDesc), c_p, c_p, ctypes.POINTER(RenderArgs), c_p, c_sz, c_p]),
    "kpn_render_rays_train": (ctypes.c_int, [ctypes.POINTER(SceneDesc), c_p, c_p, ctypes.POINTER(RenderArgs),
                                             ctypes.POINTER(TrainArgs), c_p, c_sz, c_p]),
    "kpn_frame_to_rgb8": (ctypes.c_int, [c_p, c_i32, c_i32, c_i32, c_p, c_p]),
    "kpn_mse_psnr": (ctypes.c_int, [c_p, c_p, c_i64, c_p, c_p, c_p]),
    "kpn_flops_per_point": (ctypes.c_double, [c_i32]),
    "kpn_flops_per_row": (ctypes.c_double, []),
    "kpn_profile_enable": (ctypes.c_int, [c_i32]),
    "kpn_render_rays_train_state_bytes": (c_sz, [ctypes.POINTER(SceneDesc), ctypes.POINTER(RenderArgs)]),
    "kpn_render_rays_train_keep": (ctypes.c_int, [ctypes.POINTER(SceneDesc), c_p, c_p, ctypes.POINTER(RenderArgs),
                                                  ctypes.POINTER(TrainArgs), c_p, c_sz, c_p]),
    "kpn_render_rays_train_backward_kept": (ctypes.c_int, [ctypes.POINTER(SceneDesc), c_p, c_p, ctypes.POINTER(RenderArgs),
                                                           ctypes.POINTER(TrainArgs), ctypes.POINTER(RenderGrads), c_p, c_p, c_p, c_p,
                                                           c_p, c_sz, c_p, c_sz, c_p]),
    "kpn_pix_l1_loss": (ctypes.c_int, [c_p, c_p, ctypes.c_int64, ctypes.c_float, c_p, c_p, c_p, c_p]),
    "kpn_profile_collect": (ctypes.c_int, [c_p, c_p, c_p]),
    "kpn_profile_collect2": (ctypes.c_int, [c_p, c_p, c_p, c_p]),
    "kpn_profile_collect3": (ctypes.c_int, [c_p, c_p, c_p, c_p, c_p]),
    "kpn_row_scratch_cap_bytes": (ctypes.c_size_t, []),
    "kpn_set_row_scratch_cap_bytes": (ctypes.c_int, [ctypes.c_size_t]),
    "kpn_selftest_mfma": (ctypes.c_int, [c_p, c_p, c_p]),
}
ABI_VERSION = 3


class KpnError(RuntimeError):
    pass


class KpnLibrary:
    def __init__(self, path):
        if not os.path.isfile(path):
            raise KpnError(
                f"{path} not found: the HIP library has not been built. Run `python build.py` "
                f"(hipcc --offload-arch=gfx950). There is no CPU fallback for this path.")
        self.path = path
        self.cdll = ctypes.CDLL(path)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(self.cdll, name)  # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
            setattr(self, name, fn)
        v = self.kpn_abi_version()
        if v != ABI_VERSION:
            raise KpnError(f"{path}: ABI version {v}, binding expects {ABI_VERSION}")

    def check(self, code):
        if code != 0:
            raise KpnError(f"kpnerf error {code}: {self.kpn_last_error().decode()}")

    def exported_symbols(self):
        return sorted(_SIGNATURES)


_default = None


def get_library():
    """The gfx950 product library; raises KpnError if it is not built (no fallback)."""
    global _default
    if _default is None:
        lib = KpnLibrary(DEFAULT_LIB)
        if not lib.kpn_is_device_build():
            raise KpnError(f"{DEFAULT_LIB} is not a device build")
        _default = lib
    return _default
