import sys; sys.path.insert(0,'/root/repo')
import torch
from keypointnerf_amd import ops
from keypointnerf_amd.synthetic import make_scene, random_hotpath_state_dict, to_device
import os
if os.environ.get('KPN_EXPERIMENT_LIB'):
    from keypointnerf_amd import lib as kl
    kl._default = kl.KpnLibrary(os.environ['KPN_EXPERIMENT_LIB'])
dev = torch.device("cuda", 0)
for mask in ("dense", "ellipsoid"):
    sc = to_device(make_scene(n_views=3, src_hw=(512, 512), tar_hw=(64, 64), mask=mask, seed=1), dev)
    w = ops.PackedWeights(random_hotpath_state_dict(seed=3), device=dev)
    ps = ops.PreparedScene(sc["img"], sc["cam"], sc["feat_geo"], sc["feat_tex"], sc["sp_data"], sc["src_foreground_mask"])
    N = 1_000_000
    lo, hi = sc["bounds"].reshape(2, 3)[0], sc["bounds"].reshape(2, 3)[1]
    P = (lo + (hi - lo) * torch.rand(N, 3, device=dev))[None]
    V = torch.nn.functional.normalize(torch.randn(N, 3, device=dev), dim=-1)[None]
    for mode in (0, 1):
        ops.set_geo_rows_mode(mode)
        ref = ops.query(ps, w, P, V, mode=1)[0].clone()
        bad = 0
        for i in range(60):
            o = ops.query(ps, w, P, V, mode=1)[0]
            bad += int((o != ref).any(-1).sum())
        print(mask, "mode", mode, "60 runs x 1M points: differing points total", bad, "valid", int((ref[..., 0] > 0).sum()))
ops.set_geo_rows_mode(0)
