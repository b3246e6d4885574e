#!/usr/bin/env python
"""Time kpn_geo_rows_backward at a training-batch shape (1024 rays x 192 samples, V=3): ms per call and the
MFMA rate of its kernels (recompute 62.4k + dX 48.6k + dW 70.1k MAC per row)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keypointnerf_amd import ops  # noqa: E402
from keypointnerf_amd.synthetic import make_scene, random_hotpath_state_dict, to_device  # noqa: E402


def main():
    if os.environ.get("KPN_EXPERIMENT_LIB"):  # timing experiments with ablated builds (never the product path)
        from keypointnerf_amd import lib as kl
        kl._default = kl.KpnLibrary(os.environ["KPN_EXPERIMENT_LIB"])
    rays = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    dev = torch.device("cuda", 0)
    sc = to_device(make_scene(n_views=3, src_hw=(512, 512), tar_hw=(64, 64), mask="dense", seed=1), dev)
    w = ops.PackedWeights(random_hotpath_state_dict(seed=3), device=dev)
    ps = ops.PreparedScene(sc["img"], sc["cam"], sc["feat_geo"], sc["feat_tex"], sc["sp_data"], sc["src_foreground_mask"])
    N = rays * 192
    lo, hi = sc["bounds"].reshape(2, 3)[0], sc["bounds"].reshape(2, 3)[1]
    P = lo + (hi - lo) * (0.25 + 0.5 * torch.rand(N, 3, device=dev))
    G = torch.randn(N, 3, 64, device=dev)
    _, valid = ops.query(ps, w, P[None], P[None])
    rows = int(valid.sum()) * 3
    for _ in range(2):
        ops.geo_rows_backward(ps, w, P, G)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 5
    for _ in range(K):
        ops.geo_rows_backward(ps, w, P, G)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    mac = 62400 + 48576 + 70080
    print(f"geo_rows_backward: {N} points, {rows} valid rows: {dt*1e3:.2f} ms/call, {rows/dt/1e6:.2f} M rows/s, "
          f"{2*mac*rows/dt/1e12:.1f} TFLOP/s (fp32 MFMA peak 157.3)")
    # the whole field evaluation: forward (kpn_query, mode 1) and its full reverse (kpn_query_backward)
    Vw = torch.nn.functional.normalize(torch.randn(N, 3, device=dev), dim=-1)
    Gq = torch.randn(N, 5, device=dev)
    for name, fn in (("query forward", lambda: ops.query(ps, w, P[None], Vw[None], mode=1)),
                     ("query_backward (full)", lambda: ops.query_backward(ps, w, P, Vw, Gq, mode=1)),
                     ("query_backward_geometry", lambda: ops.query_backward_geometry(ps, w, P, Gq, mode=1))):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K
        print(f"{name}: {dt*1e3:.2f} ms/call = {N/dt/1e6:.1f} M points/s")


if __name__ == "__main__":
    main()
