#!/usr/bin/env python
"""Per-phase cycles of the pair-tile rows kernels from a -DKPN_H2_TIMING build (exp_libs/<lib>.so):
python scripts/h2_timing.py <lib.so> [rows mode: 3 (default) or 2]"""
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
MODE = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ops.set_geo_rows_mode(MODE)
dll = ctypes.CDLL(sys.argv[1])
buf = (ctypes.c_ulonglong * 8)()
ops.query(ps, w, P, V, mode=1); torch.cuda.synchronize(); dll.kpn_h2_timing(buf)
ops.query(ps, w, P, V, mode=1); torch.cuda.synchronize(); dll.kpn_h2_timing(buf)
c = list(buf)
items = N * 3 / 64 / 1024
names = ["prologue", "layers1.0", "layers1.1", "layers1.2", "layers1.3", "-", "epilogue+ticket", "-"]
tot = sum(c)
print(os.path.basename(sys.argv[1]), "mode", MODE, "cycles per work item (one wave, ~%.0f items):" % items, {n: round(x / items) for n, x in zip(names, c) if n != "-"}, "total", round(tot / items))
