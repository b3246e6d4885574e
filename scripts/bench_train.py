#!/usr/bin/env python
"""Field part of one training iteration (BASELINE.json configs[3]: 1024 rays/iter, 64 coarse + 64 fine samples, 3 source
views 512x512): kpn_render_rays_train (forward) + kpn_render_rays_train_backward, ms each and iterations/s.
Synthetic scene and weights; the image encoders, the loss and the optimizer are the reference's torch code (not timed)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keypointnerf_amd import ops  # noqa: E402
from keypointnerf_amd.synthetic import make_scene, random_hotpath_state_dict, to_device  # noqa: E402


def main():
    if os.environ.get("KPN_EXPERIMENT_LIB"):
        from keypointnerf_amd import lib as kl
        kl._default = kl.KpnLibrary(os.environ["KPN_EXPERIMENT_LIB"])
    patch = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    dev = torch.device("cuda", 0)
    sc = to_device(make_scene(n_views=3, src_hw=(512, 512), tar_hw=(512, 512), mask="ellipsoid", seed=1), dev)
    w = ops.PackedWeights(random_hotpath_state_dict(seed=3), device=dev)
    ps = ops.PreparedScene(sc["img"], sc["cam"], sc["feat_geo"], sc["feat_tex"], sc["sp_data"], sc["src_foreground_mask"])
    R, Sc, Sf = patch * patch, 64, 64
    yy, xx = torch.meshgrid(torch.arange(patch), torch.arange(patch), indexing="ij")
    pix = torch.stack([xx.reshape(-1) + 256 - patch // 2, yy.reshape(-1) + 256 - patch // 2], -1).to(torch.int32).to(dev)
    g = torch.Generator(device=dev).manual_seed(0)
    u_c, u_f = torch.rand(R, Sc, device=dev, generator=g), torch.rand(R, Sf, device=dev, generator=g)
    n_c, n_f = torch.randn(R * Sc, device=dev, generator=g), torch.randn(R * (Sc + Sf), device=dev, generator=g)
    args = dict(noise_coarse=n_c, noise_fine=n_f, rand_noise_std=0.01, n_coarse=Sc, n_fine=Sf)
    fwd = lambda: ops.render_rays_train(ps, w, sc["cam_tar"], sc["bounds"], pix, u_c, u_f, 0b111, 0b101, **args)
    out = fwd()
    grads = {k: torch.randn_like(v) for k, v in out.items()}
    bwd = lambda: ops.render_rays_train_backward(ps, w, sc["cam_tar"], sc["bounds"], pix, u_c, u_f, 0b111, 0b101, grads, **args)
    res = {}
    for name, fn in (("forward", fwd), ("backward", bwd)):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        K = 5
        for _ in range(K):
            fn()
        torch.cuda.synchronize()
        res[name] = (time.perf_counter() - t0) / K
    # re-packing the weights after an optimizer step (device-side gather; the training drop-in does this every iteration)
    from keypointnerf_amd.weights import effective_weights, flatten_plain
    plain = torch.from_numpy(flatten_plain(effective_weights(random_hotpath_state_dict(seed=3)))).to(dev)
    for _ in range(3):
        ops.PackedWeights.from_plain(plain)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        ops.PackedWeights.from_plain(plain)
    torch.cuda.synchronize()
    res["pack"] = (time.perf_counter() - t0) / 20
    tot = res["forward"] + res["backward"]
    print(f"train iteration, field part: {R} rays x ({Sc}+{Sc + Sf}) samples, V=3, alpha_fine mean {float(out['alpha_fine'].mean()):.2f}: "
          f"forward {res['forward']*1e3:.2f} ms, backward {res['backward']*1e3:.2f} ms, {1.0/tot:.1f} it/s, {R/tot/1e3:.1f} k rays/s; "
          f"weight re-pack {res['pack']*1e3:.3f} ms")


if __name__ == "__main__":
    main()
