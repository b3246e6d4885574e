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
from keypointnerf_amd.losses import compute_error  # noqa: E402
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
        self.sp_encoder = torch.nn.Module()    # what install() reads from the spatial encoder (configs/zju.json:39-45)
        self.sp_encoder.sp_type, self.sp_encoder.sp_level, self.sp_encoder.n_kpt, self.sp_encoder.scale = "rel_z_decay", 3, 24, 1.0
        self.sp_encoder.kwargs = {"sigma": 0.1}
        self._scene = scene

    def attach_geo_feat(self, im, return_val=False):
        return self._scene["feat_geo"]

    def attach_tex_feat(self, im, return_val=False):
        return self._scene["feat_tex"]


def main():
    dev = torch.device("cuda", 0)
    s = to_device(make_scene(n_views=3, src_hw=(512, 512), tar_hw=(512, 512), mask="ellipsoid", seed=1, tar_focal_at_512=800.0), dev)
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
        out = net.batch_render_pifu_nerf(net=net, img_in=s["img"], cam_in=s["cam"], n_views=3, cam_tar=s["cam_tar"], level=5, stride=0,
                                         tar_img=tar, bg_img=None, feat_geo=feat_geo, feat_tex=feat_tex, sp_data=dict(s["sp_data"]),
                                         camcenter=None, objcenter=None, msk=msk, src_foreground_mask=s["src_foreground_mask"],
                                         bounds=s["bounds"], fine=True, uniform=False, blur=3, sample_per_ray_c=64,
                                         sample_per_ray_f=64, rand_noise_std=0.01)
        # the reference's loss with the shipped lambdas (configs/zju.json:109-119; VGG needs downloaded weights: left out)
        out["tex_cal"], out["tex_cal_fine"] = out["tex_fg"], out["tex_fg_fine"]
        loss, _ = compute_error(out_nerf=out, vggloss=None, lambdas={"lambda_l1_c": 1.0, "lambda_l1": 10.0, "lambda_vgg": 0.5})
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
    print(f"drop-in training step (render fwd + the reference's L1 loss terms on the device + backward + Adam on the hot-path parameters), 1024 rays: "
          f"{dt*1e3:.2f} ms/step = {1/dt:.1f} it/s; loss {losses[0]:.4f} -> {losses[-1]:.4f}")


if __name__ == "__main__":
    main()
