#!/usr/bin/env python
"""Soak of the training backward: the weight gradients are formed without atomics (fixed-order reduce), so every repetition of the
backward FROM THE SAME KEPT FORWARD STATE (same valid lists = same row order; k_mask_compact hands out list slots with an atomic,
so a repeated forward orders the rows differently and the sums differ by rounding) must return them BIT-IDENTICAL — any sporadic
wrong value in the chain kernels (k_geo_rows_bwd, k_color_bwd, k_fuse_bwd) or in k_weight_grad shows up as a difference.
(d ani_al and the feature-map gradients are accumulated with float atomics and are compared with a tolerance.)
python scripts/soak_backward.py [repetitions]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keypointnerf_amd import ops  # noqa: E402
from keypointnerf_amd.synthetic import make_scene, random_hotpath_state_dict, to_device  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
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
out, state = ops.render_rays_train(ps, w, sc["cam_tar"], sc["bounds"], pix, u_c, u_f, 0b111, 0b101, keep_state=True, **args)
grads = {k: torch.randn_like(v) for k, v in out.items()}
bwd = lambda: ops.render_rays_train_backward(ps, w, sc["cam_tar"], sc["bounds"], pix, u_c, u_f, 0b111, 0b101, grads, state=state, **args)
ref = [t.clone() for t in bwd()]
nW = ref[0].numel() - 1                      # the last entry is d ani_al (atomics)
bad, worst_maps, t0 = 0, 0.0, time.time()
for i in range(reps):
    cur = bwd()
    if not torch.equal(cur[0][:nW], ref[0][:nW]):
        bad += 1
        d = (cur[0][:nW] - ref[0][:nW]).abs()
        print("repetition", i, "differs: max abs", float(d.max()), "entries", int((d != 0).sum()), flush=True)
    for a, b in zip(cur[1:], ref[1:]):
        worst_maps = max(worst_maps, float((a - b).abs().max() / (b.abs().max() + 1e-30)))
torch.cuda.synchronize()
print(json.dumps({"repetitions": reps, "rays_per_repetition": R, "field_evaluations_per_repetition": R * (Sc + Sc + Sf),
                  "repetitions_with_a_different_weight_gradient": bad, "weight_gradient_entries_compared": nW,
                  "feature_map_gradients_max_relative_run_to_run_difference (atomics)": worst_maps,
                  "seconds": time.time() - t0}))
