#!/usr/bin/env python
"""sha256 of kpn_query's output on a fixed random point set (bit-identity checks between library builds):
python scripts/query_hash.py <lib.so> [mode]"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from keypointnerf_amd import lib as kl
kl._default = kl.KpnLibrary(sys.argv[1])
from keypointnerf_amd import ops
from keypointnerf_amd.synthetic import make_scene, random_hotpath_state_dict, to_device
dev = torch.device("cuda", 0)
if len(sys.argv) > 2:
    ops.set_geo_rows_mode(int(sys.argv[2]))
for views in (3, 4, 6):
    sc = to_device(make_scene(n_views=views, src_hw=(256, 256), tar_hw=(64, 64), mask="dense", seed=1), dev)
    w = ops.PackedWeights(random_hotpath_state_dict(seed=3), device=dev)
    ps = ops.PreparedScene(sc["img"], sc["cam"], sc["feat_geo"], sc["feat_tex"], sc["sp_data"], sc["src_foreground_mask"])
    N = 300_000
    lo, hi = sc["bounds"].reshape(2, 3)[0], sc["bounds"].reshape(2, 3)[1]
    gen = torch.Generator(device="cuda").manual_seed(4)
    P = (lo + (hi - lo) * (0.2 + 0.6 * torch.rand(N, 3, device=dev, generator=gen)))[None]
    V = torch.nn.functional.normalize(torch.randn(N, 3, device=dev, generator=gen), dim=-1)[None]
    out = ops.query(ps, w, P, V, mode=1)[0]
    print(os.path.basename(sys.argv[1]), "V =", views, hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16], float(out.abs().sum()))
