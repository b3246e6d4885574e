#!/usr/bin/env python
"""Per-phase cycles of k_geo_rows_bwd from a -DKPN_BWD_TIMING build (scripts/build_api_variant.sh <name> -DKPN_BWD_TIMING):
python scripts/bwd_timing.py exp_libs/<name>.so      (the training iteration of scripts/bench_train.py, one wave's tiles)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from keypointnerf_amd import lib as kl
kl._default = kl.KpnLibrary(sys.argv[1])
from keypointnerf_amd import ops
from keypointnerf_amd.synthetic import make_scene, random_hotpath_state_dict, to_device
dev = torch.device("cuda", 0)
sc = to_device(make_scene(n_views=3, src_hw=(512, 512), tar_hw=(512, 512), mask="ellipsoid", seed=1), dev)
w = ops.PackedWeights(random_hotpath_state_dict(seed=3), device=dev)
ps = ops.PreparedScene(sc["img"], sc["cam"], sc["feat_geo"], sc["feat_tex"], sc["sp_data"], sc["src_foreground_mask"])
patch, Sc, Sf = 32, 64, 64
R = patch * patch
yy, xx = torch.meshgrid(torch.arange(patch), torch.arange(patch), indexing="ij")
pix = torch.stack([xx.reshape(-1) + 256 - patch // 2, yy.reshape(-1) + 256 - patch // 2], -1).to(torch.int32).to(dev)
g = torch.Generator(device=dev).manual_seed(0)
u_c, u_f = torch.rand(R, Sc, device=dev, generator=g), torch.rand(R, Sf, device=dev, generator=g)
n_c, n_f = torch.randn(R * Sc, device=dev, generator=g), torch.randn(R * (Sc + Sf), device=dev, generator=g)
args = dict(noise_coarse=n_c, noise_fine=n_f, rand_noise_std=0.01, n_coarse=Sc, n_fine=Sf)
out = ops.render_rays_train(ps, w, sc["cam_tar"], sc["bounds"], pix, u_c, u_f, 0b111, 0b101, **args)
grads = {k: torch.randn_like(v) for k, v in out.items()}
bwd = lambda: ops.render_rays_train_backward(ps, w, sc["cam_tar"], sc["bounds"], pix, u_c, u_f, 0b111, 0b101, grads, **args)
dll = ctypes.CDLL(sys.argv[1])
buf = (ctypes.c_ulonglong * 16)()
bwd(); torch.cuda.synchronize(); dll.kpn_bwd_timing(buf)
bwd(); torch.cuda.synchronize(); dll.kpn_bwd_timing(buf)
c = list(buf)
names = ["point+projection", "layers1.0 encoding (+X0 dumps)", "layers1.0 geometry (+dump)", "layers1.1 (+X1)", "layers1.2 (+X2)", "X3",
         "layers1.3^T (+D3)", "layers1.2^T (+D2)", "layers1.1^T (+D1)", "layers1.0^T (+D0)", "scatter", "ticket"]
tot = sum(c)
print(os.path.basename(sys.argv[1]), "k_geo_rows_bwd, one wave, both passes of an iteration: cycles per phase (share)")
for n, x in zip(names, c):
    print(f"  {n:34s} {x:10d}  {100.0 * x / tot:5.1f} %")
print("  total", tot)
