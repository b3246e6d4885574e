#!/usr/bin/env python
"""Times the UNMODIFIED reference (PyTorch, CPU) on the headline workload, in the build container.

north_star asks for "the reference CPU PyTorch path timed on the same box's host cores (core count stated)".
/root/reference cannot travel to the GPU box, so this is measured where the reference is mounted and committed as
profiles/reference_cpu_pytorch.json (bench.py carries it as cpu_baseline.reference_pytorch_recorded, labelled with the host it
was measured on).  Workload: tiles of BASELINE configs[1] — batch_render_pifu_nerf (src/model.py:942-1108), level 4
strided tile (64x64 = 4096 rays) of a 512x512 target, V=3 source views 512x512, Sc = Sf = 64, uniform=True,
fine=True, the bench scene (keypointnerf_amd.synthetic.make_scene seed 1, ellipsoid mask, tar_focal_at_512=800).

TEST / MEASUREMENT INFRASTRUCTURE: imports oracle.ref_shim; nothing under keypointnerf_amd/ imports this.
"""
import json
import os
import platform
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from keypointnerf_amd.synthetic import make_scene, perturb_reference_net  # noqa: E402
from oracle import ref_shim  # noqa: E402


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor()


def main():
    n_tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    torch.set_num_threads(os.cpu_count())
    net = ref_shim.build_reference_net(seed=0)
    perturb_reference_net(net, seed=7)
    scene = make_scene(n_views=3, src_hw=(512, 512), tar_hw=(512, 512), mask="ellipsoid", seed=1, tar_focal_at_512=800.0)
    cfg = dict(fine=True, uniform=True, sample_per_ray_c=64, sample_per_ray_f=64,
               src_foreground_mask=scene["src_foreground_mask"], bounds=scene["bounds"])
    times = []
    with torch.no_grad():
        for k in range(n_tiles + 1):                      # first tile = warm-up
            strd = torch.tensor([[float((3 * k) % 8), float((5 * k) % 8)]])
            t0 = time.perf_counter()
            out = net.batch_render_pifu_nerf(net, scene["img"], scene["cam"], 3, scene["cam_tar"], 4, strd, None,
                                             scene["feat_geo"], scene["feat_tex"], dict(scene["sp_data"]), None, **cfg)
            times.append(time.perf_counter() - t0)
            print(f"tile {k}: {times[-1]:.2f} s, alpha_fine mean {float(out['alpha_fine'].mean()):.4f}", flush=True)
    dt = sum(times[1:]) / n_tiles
    rays, evals = 4096, 4096 * (64 + 128)
    res = {"what": "reference facebookresearch/KeypointNeRF batch_render_pifu_nerf (unmodified, imported from /root/reference), "
                   "PyTorch CPU, fp32, one level-4 tile (4096 rays) of the 512x512 configs[1] frame, 64 coarse + 128 fine-pass "
                   "evaluations per ray",
           "seconds_per_tile": dt, "rays_per_sec": rays / dt, "field_evaluations_per_sec": evals / dt,
           "seconds_per_512x512_frame_extrapolated": dt * 64, "tiles_timed": n_tiles,
           "torch": torch.__version__, "threads": torch.get_num_threads(), "cores": os.cpu_count(), "cpu": cpu_model(),
           "host": "build container (no GPU); the reference sources are not available on the GPU box"}
    path = os.path.join(ROOT, "profiles", "reference_cpu_pytorch.json")
    json.dump(res, open(path, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
