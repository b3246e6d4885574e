#!/usr/bin/env python
"""The reference's train-mode call through the drop-in, end to end: net.batch_render_pifu_nerf(net, ...) in train
mode + a loss on its outputs + loss.backward(), 1024 rays (32x32 patch) x (64+128) samples, 3 source views 512x512.
The module is a parameter carrier with the reference's parameter names (the reference itself is not on the GPU box);
encoders, real loss and optimizer are outside this measurement."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keypointnerf_amd.dropin import install  # noqa: E402
from keypointnerf_amd.synthetic import make_scene, random_hotpath_state_dict, to_device  # noqa: E402


class Carrier(torch.nn.Module):
    def __init__(self, sd, scene):
        super().__init__()
        for k, v in sd.items():
            mod, parts = self, k.split(".")
            for p in parts[:-1]:
                if not hasattr(mod, p):
                    setattr(mod, p, torch.nn.Module())
                mod = getattr(mod, p)
            mod.register_parameter(parts[-1], torch.nn.Parameter(torch.as_tensor(v).clone()))
        self.disable_fg_mask = False
        self._scene = scene

    def attach_geo_feat(self, im, return_val=False):
        return self._scene["feat_geo"]

    def attach_tex_feat(self, im, return_val=False):
        return self._scene["feat_tex"]


def main():
    dev = torch.device("cuda", 0)
    s = to_device(make_scene(n_views=3, src_hw=(512, 512), tar_hw=(512, 512), mask="ellipsoid", seed=1), dev)
    net = install(Carrier(random_hotpath_state_dict(seed=3), s).to(dev))
    net.train()
    net.train_out_h = net.train_out_w = 32
    yy, xx = torch.meshgrid(torch.arange(512), torch.arange(512), indexing="ij")
    msk = (((yy - 256) ** 2 + (xx - 256) ** 2) < 60 ** 2)[None, None].to(dev)
    feat_geo = [f.clone().requires_grad_(True) for f in s["feat_geo"]]
    feat_tex = s["feat_tex"].clone().requires_grad_(True)
    opt = torch.optim.Adam(net.parameters(), lr=1e-5)
    tar = torch.rand(1, 3, 512, 512, device=dev)

    def step():
        opt.zero_grad(set_to_none=True)
        out = net.batch_render_pifu_nerf(net, s["img"], s["cam"], 3, s["cam_tar"], 5, 0, tar, feat_geo, feat_tex, dict(s["sp_data"]), None,
                                         fine=True, uniform=False, sample_per_ray_c=64, sample_per_ray_f=64, rand_noise_std=0.01,
                                         src_foreground_mask=s["src_foreground_mask"], bounds=s["bounds"], msk=msk)
        loss = (out["tex_fg_fine"] - out["tar_img"]).abs().mean() + (out["tex_fg"] - out["tar_img"]).abs().mean() + \
            ((out["alpha_fine"] - out["tar_alpha"][:, 0]) ** 2).mean()
        loss.backward()
        opt.step()
        return float(loss)

    np.random.seed(0)
    torch.manual_seed(0)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 10
    losses = [step() for _ in range(K)]
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print(f"drop-in training step (render fwd + L1/L2 loss + backward + Adam on the hot-path parameters), 1024 rays: "
          f"{dt*1e3:.2f} ms/step = {1/dt:.1f} it/s; loss {losses[0]:.4f} -> {losses[-1]:.4f}")


if __name__ == "__main__":
    main()
