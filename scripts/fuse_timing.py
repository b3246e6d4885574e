#!/usr/bin/env python
"""Per-phase cycles of k_fuse_color from a -DKPN_FUSE_TIMING build: python scripts/fuse_timing.py <lib.so>"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from keypointnerf_amd import lib as kl
kl._default = kl.KpnLibrary(sys.argv[1])
from keypointnerf_amd import ops
from keypointnerf_amd.synthetic import make_scene, random_hotpath_state_dict, to_device
dev = torch.device("cuda", 0)
sc = to_device(make_scene(n_views=3, src_hw=(256, 256), tar_hw=(64, 64), mask="dense", seed=1), dev)
w = ops.PackedWeights(random_hotpath_state_dict(seed=3), device=dev)
ps = ops.PreparedScene(sc["img"], sc["cam"], sc["feat_geo"], sc["feat_tex"], sc["sp_data"], sc["src_foreground_mask"])
N = 4_000_000
lo, hi = sc["bounds"].reshape(2, 3)[0], sc["bounds"].reshape(2, 3)[1]
P = (lo + (hi - lo) * (0.2 + 0.6 * torch.rand(N, 3, device=dev)))[None]
V = torch.nn.functional.normalize(torch.randn(N, 3, device=dev), dim=-1)[None]
dll = ctypes.CDLL(sys.argv[1])
buf = (ctypes.c_ulonglong * 8)()
ops.query(ps, w, P, V, mode=1); torch.cuda.synchronize(); dll.kpn_fuse_timing(buf)
ops.query(ps, w, P, V, mode=1); torch.cuda.synchronize(); dll.kpn_fuse_timing(buf)
c = list(buf)
tiles = N / 32 / 2048
names = ["ticket+list", "pooling (rows read twice)", "layers2", "compress", "x' of the views + statistics (2 passes)", "heads of the views", "store", "loop"]
print(os.path.basename(sys.argv[1]), "cycles per tile (one wave, ~%.0f tiles):" % tiles, {n: round(x / tiles) for n, x in zip(names, c)}, "total", round(sum(c) / tiles))
