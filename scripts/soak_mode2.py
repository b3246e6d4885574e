#!/usr/bin/env python
"""Soak of k_geo_rows_h2 (kpn_set_geo_rows_mode(2): split-bf16, two tiles per wave, one wave per SIMD) on the MI355X:
N random points x V views evaluated REPEAT times; every repeat must be bit-identical to the first and, point by point,
within fp32-class distance of the fp32-MFMA kernel (mode 0).  Prints one JSON line.

    python scripts/soak_mode2.py [--points 400000] [--repeats 50] [--mode 2]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=400000)
    ap.add_argument("--repeats", type=int, default=50)
    ap.add_argument("--mode", type=int, default=2)
    ap.add_argument("--views", type=int, default=3)
    ap.add_argument("--lib", default="", help="an experimental build of the library (code-placement variants) instead of the product one")
    ap.add_argument("--mask", default="dense")
    args = ap.parse_args()
    from keypointnerf_amd import lib as kl
    if args.lib:
        kl._default = kl.KpnLibrary(args.lib)
    from keypointnerf_amd import ops
    from keypointnerf_amd.synthetic import make_scene, random_hotpath_state_dict, to_device
    dev = torch.device("cuda", 0)
    sc = to_device(make_scene(n_views=args.views, src_hw=(256, 256), tar_hw=(64, 64), mask=args.mask, seed=1), dev)
    w = ops.PackedWeights(random_hotpath_state_dict(seed=3), device=dev)
    ps = ops.PreparedScene(sc["img"], sc["cam"], sc["feat_geo"], sc["feat_tex"], sc["sp_data"], sc["src_foreground_mask"])
    lo, hi = sc["bounds"].reshape(2, 3)[0], sc["bounds"].reshape(2, 3)[1]
    gen = torch.Generator(device="cuda").manual_seed(4)
    P = (lo + (hi - lo) * (0.2 + 0.6 * torch.rand(args.points, 3, device=dev, generator=gen)))[None]
    V = torch.nn.functional.normalize(torch.randn(args.points, 3, device=dev, generator=gen), dim=-1)[None]
    ops.set_geo_rows_mode(0)
    ref, valid = ops.query(ps, w, P, V, mode=1)
    ref = ref.clone()
    n_valid = int(valid.sum())
    scale = ref.abs().amax(dim=(0, 1))
    ops.set_geo_rows_mode(args.mode)
    first = ops.query(ps, w, P, V, mode=1)[0].clone()
    off = int((((first - ref).abs() > 2e-5 * scale + 1e-6).any(-1)).sum())
    differing = 0
    worst = 0
    t0 = time.time()
    for _ in range(args.repeats):
        r = ops.query(ps, w, P, V, mode=1)[0]
        d = int((r != first).any(-1).sum())
        differing += d
        worst = max(worst, d)
    torch.cuda.synchronize()
    ops.set_geo_rows_mode(0)
    print(json.dumps({"lib": os.path.basename(args.lib) or "product", "mask": args.mask, "mode": args.mode, "points": args.points, "valid_points": n_valid, "views": args.views, "repeats": args.repeats,
                      "row_evaluations": n_valid * args.views * (args.repeats + 1), "points_off_vs_fp32_mfma": off,
                      "max_abs_vs_fp32_mfma": float((first - ref).abs().max()), "differing_points_total": differing,
                      "differing_points_worst_repeat": worst, "seconds": time.time() - t0}))


if __name__ == "__main__":
    main()
