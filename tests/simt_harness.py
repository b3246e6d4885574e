"""Drive the host SIMT-emulator build of the kernels (tests/simt) through the C ABI with numpy buffers.

TEST INFRASTRUCTURE: same kernel sources, same ABI, host memory.  Used by the CPU ("not gpu") tests to
check kernel logic against the oracle without a GPU.
"""
import ctypes
import os
import subprocess

import numpy as np

from keypointnerf_amd import lib as kl

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIMT_DIR = os.path.join(ROOT, "tests", "simt")
SIMT_SO = os.path.join(SIMT_DIR, "_build", "libkpnerf_simt.so")
CLANGXX = "/opt/rocm/lib/llvm/bin/clang++"
_CSRC = os.path.join(ROOT, "keypointnerf_amd", "csrc")
_SOURCES = [os.path.join(_CSRC, f) for f in sorted(os.listdir(_CSRC)) if f.endswith((".hip", ".h"))] + \
           [os.path.join(SIMT_DIR, f) for f in ("simt.h", "simt.cpp")] + [os.path.join(ROOT, "include", "kpnerf.h")]


def build_simt():
    if os.path.exists(SIMT_SO) and all(os.path.getmtime(SIMT_SO) >= os.path.getmtime(s) for s in _SOURCES):
        return SIMT_SO
    os.makedirs(os.path.dirname(SIMT_SO), exist_ok=True)
    cxx = CLANGXX if os.path.exists(CLANGXX) else "clang++"
    subprocess.check_call([cxx, "-std=c++17", "-O1", "-fPIC", "-shared", "-DKPN_SIMT_EMU", "-include",
                           os.path.join(SIMT_DIR, "simt.h"), "-x", "c++",
                           os.path.join(ROOT, "keypointnerf_amd", "csrc", "kpn_api.hip"),
                           os.path.join(SIMT_DIR, "simt.cpp"), "-o", SIMT_SO, "-lpthread"])
    return SIMT_SO


_lib = None


def simt_lib():
    global _lib
    if _lib is None:
        _lib = kl.KpnLibrary(build_simt())
        assert _lib.kpn_is_device_build() == 0
    return _lib


def f32(a):
    import torch
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().numpy()
    return np.ascontiguousarray(a, dtype=np.float32)


def ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


class HostScene:
    """kpn_scene_desc + prepared workspace, in host memory (numpy)."""

    def __init__(self, lib, scene, disable_fg_mask=False, sigma=0.1):
        cam = scene["cam"]
        self.lib = lib
        self.bufs = dict(KRT=f32(cam["KRT"]), extrin=f32(scene["sp_data"]["extrin"]),
                         kpt3d=f32(scene["sp_data"]["kpt3d"]).reshape(-1, 3), img=f32(scene["img"]),
                         geo0=f32(scene["feat_geo"][0]), geo1=f32(scene["feat_geo"][1]), tex=f32(scene["feat_tex"]))
        V, _, H, W = self.bufs["img"].shape
        m = scene["src_foreground_mask"]
        self.bufs["fg_mask"] = np.ascontiguousarray(np.asarray(m.cpu().numpy() if hasattr(m, "cpu") else m).reshape(V, H, W).astype(np.uint8))
        d = kl.SceneDesc()
        d.n_views, d.src_h, d.src_w = V, H, W
        d.geo0_h, d.geo0_w = self.bufs["geo0"].shape[-2:]
        d.geo1_h, d.geo1_w = self.bufs["geo1"].shape[-2:]
        d.tex_h, d.tex_w = self.bufs["tex"].shape[-2:]
        d.disable_fg_mask = int(disable_fg_mask)
        d.znear, d.zfar, d.nml_scale, d.sigma = float(cam["znear"]), float(cam["zfar"]), float(cam["nml_scale"]), sigma
        for k in ("KRT", "extrin", "kpt3d", "img", "fg_mask", "geo0", "geo1", "tex"):
            setattr(d, k, self.bufs[k].ctypes.data)
        self.desc = d
        self.V = V
        nbytes = lib.kpn_scene_workspace_bytes(ctypes.byref(d))
        assert nbytes > 0, lib.kpn_last_error()
        self.ws = np.zeros(nbytes // 4, np.float32)
        lib.check(lib.kpn_scene_prepare(ctypes.byref(d), ptr(self.ws), None))


def pack_weights(lib, state_dict):
    from keypointnerf_amd.weights import effective_weights, flatten_plain
    plain = flatten_plain(effective_weights(state_dict))
    assert plain.size == lib.kpn_plain_weight_floats()
    packed = np.zeros(lib.kpn_packed_weight_floats(), np.float32)
    lib.check(lib.kpn_pack_weights(ptr(plain), ptr(packed)))
    return packed


def query(lib, hs, packed, pts, view, mode=0):
    pts, view = f32(pts).reshape(-1, 3), f32(view).reshape(-1, 3)
    N = pts.shape[0]
    out = np.full((N, 5), np.nan, np.float32)
    valid = np.zeros(N, np.uint8)
    nb = lib.kpn_query_workspace_bytes(N, hs.V)
    ws = np.zeros(nb, np.uint8)
    lib.check(lib.kpn_query(ctypes.byref(hs.desc), ptr(hs.ws), ptr(packed), N, ptr(pts), ptr(view), mode, ptr(out),
                            ptr(valid), ptr(ws), nb, None))
    return out, valid.astype(bool)


def render(lib, hs, packed, cam_tar, bounds, grid, Sc, Sf, fine=True, chunk_rays=0, stages=False):
    """grid = (x0, y0, step, nx, ny[, step_y]).  stages: -> (outputs, kpn_render_stages arrays in ray order)."""
    x0, y0, step, nx, ny = grid[:5]
    K, RT, b = f32(cam_tar["K"]).reshape(4, 4), f32(cam_tar["RT"]).reshape(4, 4), f32(bounds).reshape(2, 3)
    R = nx * ny
    o = {"tex_fg": np.full((3, ny, nx), np.nan, np.float32), "depth": np.full((ny, nx), np.nan, np.float32),
         "alpha": np.full((ny, nx), np.nan, np.float32)}
    if fine:
        o.update({"tex_fg_fine": np.full((3, ny, nx), np.nan, np.float32), "depth_fine": np.full((ny, nx), np.nan, np.float32),
                  "alpha_fine": np.full((ny, nx), np.nan, np.float32), "sdf": np.full((ny, nx), np.nan, np.float32)})
    a = kl.RenderArgs()
    a.K, a.RT, a.bounds = K.ctypes.data, RT.ctypes.data, b.ctypes.data
    a.znear, a.zfar = float(cam_tar["znear"]), float(cam_tar["zfar"])
    a.x0, a.y0, a.step, a.nx, a.ny = x0, y0, step, nx, ny
    a.step_y = grid[5] if len(grid) > 5 else 0
    a.n_coarse, a.n_fine, a.fine, a.chunk_rays = Sc, Sf, int(fine), chunk_rays
    for k, v in o.items():
        setattr(a, k, v.ctypes.data)
    st = None
    if stages:
        st = {"z_coarse": np.full((R, Sc), np.nan, np.float32), "rgba_coarse": np.full((R, Sc, 5), np.nan, np.float32),
              "dirs": np.full((R, 3), np.nan, np.float32), "cam_pos": np.full(3, np.nan, np.float32)}
        if fine:
            st.update({"z_fine": np.full((R, Sc + Sf), np.nan, np.float32), "rgba_fine": np.full((R, Sc + Sf, 5), np.nan, np.float32)})
        cst = kl.RenderStages()
        for k, v in st.items():
            setattr(cst, k, v.ctypes.data)
        a.stages = ctypes.pointer(cst)
    nb = lib.kpn_render_workspace_bytes(ctypes.byref(hs.desc), ctypes.byref(a))
    assert nb > 0, lib.kpn_last_error()
    ws = np.zeros(nb, np.uint8)
    lib.check(lib.kpn_render_rays(ctypes.byref(hs.desc), ptr(hs.ws), ptr(packed), ctypes.byref(a), ptr(ws), nb, None))
    return (o, st) if stages else o


def render_train(lib, hs, packed, cam_tar, bounds, pix, Sc, Sf, u_c, noise_c, noise_f, u_f, keep_c, keep_f, noise_std, chunk_rays=0):
    """kpn_render_rays_train on host buffers; returns (C,R)-planar outputs reshaped to (R,...)."""
    K, RT, b = f32(cam_tar["K"]).reshape(4, 4), f32(cam_tar["RT"]).reshape(4, 4), f32(bounds).reshape(2, 3)
    pix = np.ascontiguousarray(pix, dtype=np.int32).reshape(-1, 2)
    R = pix.shape[0]
    o = {k: np.full((3, R), np.nan, np.float32) for k in ("tex_fg", "tex_fg_fine")}
    o.update({k: np.full(R, np.nan, np.float32) for k in ("depth", "alpha", "depth_fine", "alpha_fine", "sdf")})
    a = kl.RenderArgs()
    a.K, a.RT, a.bounds = K.ctypes.data, RT.ctypes.data, b.ctypes.data
    a.znear, a.zfar = float(cam_tar["znear"]), float(cam_tar["zfar"])
    a.x0, a.y0, a.step, a.nx, a.ny = 0, 0, 1, R, 1
    a.n_coarse, a.n_fine, a.fine, a.chunk_rays = Sc, Sf, 1, chunk_rays
    for k, v in o.items():
        setattr(a, k, v.ctypes.data)
    bufs = dict(u_c=f32(u_c).reshape(R, Sc), noise_c=f32(noise_c).reshape(-1), noise_f=f32(noise_f).reshape(-1), u_f=f32(u_f).reshape(R, Sf))
    t = kl.TrainArgs()
    t.pix, t.u_coarse, t.noise_coarse, t.noise_fine, t.u_fine = (pix.ctypes.data, bufs["u_c"].ctypes.data, bufs["noise_c"].ctypes.data,
                                                                 bufs["noise_f"].ctypes.data, bufs["u_f"].ctypes.data)
    t.keep_coarse, t.keep_fine, t.rand_noise_std = keep_c, keep_f, float(noise_std)
    nb = lib.kpn_render_workspace_bytes(ctypes.byref(hs.desc), ctypes.byref(a))
    ws = np.zeros(nb, np.uint8)
    lib.check(lib.kpn_render_rays_train(ctypes.byref(hs.desc), ptr(hs.ws), ptr(packed), ctypes.byref(a), ctypes.byref(t), ptr(ws), nb, None))
    return o


def geo_rows_backward(lib, hs, packed, pts, d_x, keep=0xFFFFFFFF):
    """kpn_geo_rows_backward on host buffers -> (d_plain flat, d_geo0 (V,64,h,w), d_geo1 (V,8,h,w)) (NCHW views)."""
    pts = f32(pts).reshape(-1, 3)
    N = pts.shape[0]
    d_x = f32(d_x).reshape(N, hs.V, 64)
    d = hs.desc
    d_plain = np.zeros(lib.kpn_plain_weight_floats(), np.float32)
    d_g0 = np.zeros((hs.V, d.geo0_h, d.geo0_w, 64), np.float32)
    d_g1 = np.zeros((hs.V, d.geo1_h, d.geo1_w, 8), np.float32)
    nb = lib.kpn_geo_rows_backward_workspace_bytes(N, hs.V)
    ws = np.zeros(nb, np.uint8)
    lib.check(lib.kpn_geo_rows_backward(ctypes.byref(d), ptr(hs.ws), ptr(packed), N, ptr(pts), keep, ptr(d_x), ptr(d_plain),
                                        ptr(d_g0), ptr(d_g1), ptr(ws), nb, None))
    return d_plain, d_g0.transpose(0, 3, 1, 2), d_g1.transpose(0, 3, 1, 2)


def query_backward_geometry(lib, hs, packed, pts, d_out, mode=0, keep=0xFFFFFFFF, noise=None, noise_std=0.0):
    """kpn_query_backward_geometry on host buffers -> (d_plain, d_geo0 NCHW, d_geo1 NCHW)."""
    pts = f32(pts).reshape(-1, 3)
    N = pts.shape[0]
    d_out = f32(d_out).reshape(N, 5)
    d = hs.desc
    d_plain = np.zeros(lib.kpn_plain_weight_floats(), np.float32)
    d_g0 = np.zeros((hs.V, d.geo0_h, d.geo0_w, 64), np.float32)
    d_g1 = np.zeros((hs.V, d.geo1_h, d.geo1_w, 8), np.float32)
    nz = None if noise is None else f32(noise).reshape(-1)
    nb = lib.kpn_query_backward_geometry_workspace_bytes(N, hs.V)
    ws = np.zeros(nb, np.uint8)
    lib.check(lib.kpn_query_backward_geometry(ctypes.byref(d), ptr(hs.ws), ptr(packed), N, ptr(pts), mode, keep, ptr(nz), noise_std,
                                              ptr(d_out), ptr(d_plain), ptr(d_g0), ptr(d_g1), ptr(ws), nb, None))
    return d_plain, d_g0.transpose(0, 3, 1, 2), d_g1.transpose(0, 3, 1, 2)


def query_backward(lib, hs, packed, pts, view, d_out, mode=0, keep=0xFFFFFFFF, noise=None, noise_std=0.0):
    """kpn_query_backward on host buffers -> (d_plain, d_geo0, d_geo1, d_tex) (maps as NCHW views)."""
    pts, view = f32(pts).reshape(-1, 3), f32(view).reshape(-1, 3)
    N = pts.shape[0]
    d_out = f32(d_out).reshape(N, 5)
    d = hs.desc
    d_plain = np.zeros(lib.kpn_plain_weight_floats(), np.float32)
    d_g0 = np.zeros((hs.V, d.geo0_h, d.geo0_w, 64), np.float32)
    d_g1 = np.zeros((hs.V, d.geo1_h, d.geo1_w, 8), np.float32)
    d_tx = np.zeros((hs.V, d.tex_h, d.tex_w, 8), np.float32)
    nz = None if noise is None else f32(noise).reshape(-1)
    nb = lib.kpn_query_backward_workspace_bytes(N, hs.V)
    ws = np.zeros(nb, np.uint8)
    lib.check(lib.kpn_query_backward(ctypes.byref(d), ptr(hs.ws), ptr(packed), N, ptr(pts), ptr(view), mode, keep, ptr(nz), noise_std,
                                     ptr(d_out), ptr(d_plain), ptr(d_g0), ptr(d_g1), ptr(d_tx), ptr(ws), nb, None))
    return d_plain, d_g0.transpose(0, 3, 1, 2), d_g1.transpose(0, 3, 1, 2), d_tx.transpose(0, 3, 1, 2)


def render_train_backward(lib, hs, packed, cam_tar, bounds, pix, Sc, Sf, u_c, noise_c, noise_f, u_f, keep_c, keep_f, noise_std, grads):
    """kpn_render_rays_train_backward on host buffers; grads: name -> array shaped like the output (any layout with
    the planar (C,R) order when flattened).  Returns (d_plain, d_geo0, d_geo1, d_tex) (maps as NCHW views)."""
    K, RT, b = f32(cam_tar["K"]).reshape(4, 4), f32(cam_tar["RT"]).reshape(4, 4), f32(bounds).reshape(2, 3)
    pix = np.ascontiguousarray(pix, dtype=np.int32).reshape(-1, 2)
    R = pix.shape[0]
    a = kl.RenderArgs()
    a.K, a.RT, a.bounds = K.ctypes.data, RT.ctypes.data, b.ctypes.data
    a.znear, a.zfar = float(cam_tar["znear"]), float(cam_tar["zfar"])
    a.x0, a.y0, a.step, a.nx, a.ny = 0, 0, 1, R, 1
    a.n_coarse, a.n_fine, a.fine, a.chunk_rays = Sc, Sf, 1, 0
    bufs = dict(u_c=f32(u_c).reshape(R, Sc), noise_c=f32(noise_c).reshape(-1), noise_f=f32(noise_f).reshape(-1), u_f=f32(u_f).reshape(R, Sf))
    t = kl.TrainArgs()
    t.pix, t.u_coarse, t.noise_coarse, t.noise_fine, t.u_fine = (pix.ctypes.data, bufs["u_c"].ctypes.data, bufs["noise_c"].ctypes.data,
                                                                 bufs["noise_f"].ctypes.data, bufs["u_f"].ctypes.data)
    t.keep_coarse, t.keep_fine, t.rand_noise_std = keep_c, keep_f, float(noise_std)
    g, keep_alive = kl.RenderGrads(), []
    for name, arr in grads.items():
        arr = f32(arr).reshape(-1)
        keep_alive.append(arr)
        setattr(g, "d_" + name, arr.ctypes.data)
    d = hs.desc
    d_plain = np.zeros(lib.kpn_plain_weight_floats(), np.float32)
    d_g0 = np.zeros((hs.V, d.geo0_h, d.geo0_w, 64), np.float32)
    d_g1 = np.zeros((hs.V, d.geo1_h, d.geo1_w, 8), np.float32)
    d_tx = np.zeros((hs.V, d.tex_h, d.tex_w, 8), np.float32)
    nb = lib.kpn_render_rays_train_backward_workspace_bytes(ctypes.byref(d), ctypes.byref(a))
    assert nb > 0, lib.kpn_last_error()
    ws = np.zeros(nb, np.uint8)
    lib.check(lib.kpn_render_rays_train_backward(ctypes.byref(d), ptr(hs.ws), ptr(packed), ctypes.byref(a), ctypes.byref(t), ctypes.byref(g),
                                                 ptr(d_plain), ptr(d_g0), ptr(d_g1), ptr(d_tx), ptr(ws), nb, None))
    return d_plain, d_g0.transpose(0, 3, 1, 2), d_g1.transpose(0, 3, 1, 2), d_tx.transpose(0, 3, 1, 2)


def render_train_keep_and_backward(lib, hs, packed, cam_tar, bounds, pix, Sc, Sf, u_c, noise_c, noise_f, u_f, keep_c, keep_f, noise_std,
                                   grads, chunk_rays=0):
    """kpn_render_rays_train_keep + kpn_render_rays_train_backward_kept on host buffers: returns (outputs, gradients)."""
    K, RT, b = f32(cam_tar["K"]).reshape(4, 4), f32(cam_tar["RT"]).reshape(4, 4), f32(bounds).reshape(2, 3)
    pix = np.ascontiguousarray(pix, dtype=np.int32).reshape(-1, 2)
    R = pix.shape[0]
    o = {k: np.full((3, R), np.nan, np.float32) for k in ("tex_fg", "tex_fg_fine")}
    o.update({k: np.full(R, np.nan, np.float32) for k in ("depth", "alpha", "depth_fine", "alpha_fine", "sdf")})
    a = kl.RenderArgs()
    a.K, a.RT, a.bounds = K.ctypes.data, RT.ctypes.data, b.ctypes.data
    a.znear, a.zfar = float(cam_tar["znear"]), float(cam_tar["zfar"])
    a.x0, a.y0, a.step, a.nx, a.ny = 0, 0, 1, R, 1
    a.n_coarse, a.n_fine, a.fine, a.chunk_rays = Sc, Sf, 1, chunk_rays
    for k, v in o.items():
        setattr(a, k, v.ctypes.data)
    bufs = dict(u_c=f32(u_c).reshape(R, Sc), noise_c=f32(noise_c).reshape(-1), noise_f=f32(noise_f).reshape(-1), u_f=f32(u_f).reshape(R, Sf))
    t = kl.TrainArgs()
    t.pix, t.u_coarse, t.noise_coarse, t.noise_fine, t.u_fine = (pix.ctypes.data, bufs["u_c"].ctypes.data, bufs["noise_c"].ctypes.data,
                                                                 bufs["noise_f"].ctypes.data, bufs["u_f"].ctypes.data)
    t.keep_coarse, t.keep_fine, t.rand_noise_std = keep_c, keep_f, float(noise_std)
    d = hs.desc
    ns = lib.kpn_render_rays_train_state_bytes(ctypes.byref(d), ctypes.byref(a))
    assert ns > 0, lib.kpn_last_error()
    state = np.zeros(ns, np.uint8)
    lib.check(lib.kpn_render_rays_train_keep(ctypes.byref(d), ptr(hs.ws), ptr(packed), ctypes.byref(a), ctypes.byref(t), ptr(state), ns, None))
    g, keep_alive = kl.RenderGrads(), []
    for name, arr in grads.items():
        arr = f32(arr).reshape(-1)
        keep_alive.append(arr)
        setattr(g, "d_" + name, arr.ctypes.data)
    d_plain = np.zeros(lib.kpn_plain_weight_floats(), np.float32)
    d_g0 = np.zeros((hs.V, d.geo0_h, d.geo0_w, 64), np.float32)
    d_g1 = np.zeros((hs.V, d.geo1_h, d.geo1_w, 8), np.float32)
    d_tx = np.zeros((hs.V, d.tex_h, d.tex_w, 8), np.float32)
    nb = lib.kpn_render_rays_train_backward_workspace_bytes(ctypes.byref(d), ctypes.byref(a))
    ws = np.zeros(nb, np.uint8)
    lib.check(lib.kpn_render_rays_train_backward_kept(ctypes.byref(d), ptr(hs.ws), ptr(packed), ctypes.byref(a), ctypes.byref(t),
                                                      ctypes.byref(g), ptr(d_plain), ptr(d_g0), ptr(d_g1), ptr(d_tx), ptr(state), ns,
                                                      ptr(ws), nb, None))
    return o, (d_plain, d_g0.transpose(0, 3, 1, 2), d_g1.transpose(0, 3, 1, 2), d_tx.transpose(0, 3, 1, 2))
