import sys; sys.path.insert(0,'/root/repo')
import torch
from keypointnerf_amd import ops, lib as kl
from keypointnerf_amd.synthetic import make_scene, random_hotpath_state_dict, to_device
dev = torch.device("cuda", 0)
sc = to_device(make_scene(n_views=3, src_hw=(256, 256), tar_hw=(64, 64), mask="dense", seed=1), dev)
w = ops.PackedWeights(random_hotpath_state_dict(seed=3), device=dev)
ps = ops.PreparedScene(sc["img"], sc["cam"], sc["feat_geo"], sc["feat_tex"], sc["sp_data"], sc["src_foreground_mask"])
N = 400000
lo, hi = sc["bounds"].reshape(2, 3)[0], sc["bounds"].reshape(2, 3)[1]
P = lo + (hi - lo) * (0.2 + 0.6 * torch.rand(N, 3, device=dev))
V = torch.nn.functional.normalize(torch.randn(N, 3, device=dev), dim=-1)
import os
if os.environ.get('KPN_EXPERIMENT_LIB'):
    kl._default = kl.KpnLibrary(os.environ['KPN_EXPERIMENT_LIB'])
L = kl.get_library()
outs = {}
for mode in (0, 1):
    L.check(L.kpn_set_geo_rows_mode(mode))
    rs = [ops.query(ps, w, P[None], V[None], mode=1)[0].clone() for _ in range(4)]
    torch.cuda.synchronize()
    print("mode", mode, "max run-to-run diff", max(float((r - rs[0]).abs().max()) for r in rs[1:]), "n differing", int((rs[1] != rs[0]).any(-1).sum()))
    outs[mode] = rs[0]
d = (outs[1] - outs[0]).abs()
print("split vs fp32: max abs", float(d.max()), "mean abs", float(d.mean()), "sigma scale", float(outs[0][..., 0].abs().max()))
