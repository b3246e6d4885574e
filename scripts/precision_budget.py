#!/usr/bin/env python
"""Precision budget per layer and operand of the two-fp16-piece kernels (round 6, item 2 of the round-5 review; SURVEY section 7
"operand precision vs the 1e-4 bar"; reference src/utils.py:691-720 MLPUNetFusion, src/model.py:1267-1302 IBRRenderingHead).

Every fp32 operand of every Linear is carried as two fp16 pieces with three products (hh, hl, lh).  Which operands NEED their lo
piece?  This runs a PROBE build of the library (exp_libs/probe.so: -DKPN_PRECISION_PROBE, scripts/build_pair_variant.sh; the
shipped library has no such switch) in which a 64-bit mask replaces the lo piece of the B operand (activations) or of the A operand
(weights) of one layer by zero — exactly the arithmetic a kernel variant with that product dropped would do — and measures, per
(layer, operand), the error of the rendered rays against the C oracle on a fixed set of scenes: the reference fixtures at the
BASELINE sample counts (case_p configs[1] tile, case_q configs[4] chunk, case_s encoder maps, case_t trained weights) and N seeded
random scenes of the parity sweep (tests/test_gpu_fuzz.py).  A variant QUALIFIES only if its worst error stays <= 2.5e-5 (a 4x
margin under the bar) wherever the shipped arithmetic's does, and no ray beyond today's goes above the bar.
MEASUREMENT INFRASTRUCTURE (imports the oracle).  Usage: precision_budget.py [n_random_scenes] -> gpurun_out/precision_budget.{json,md}"""
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401
from keypointnerf_amd import lib as kl  # noqa: E402

PROBE = os.path.join(ROOT, "exp_libs", "probe.so")
kl._default = kl.KpnLibrary(PROBE)
from keypointnerf_amd import ops  # noqa: E402
from keypointnerf_amd.synthetic import make_scene, random_hotpath_state_dict, to_device  # noqa: E402
from oracle import oracle  # noqa: E402
from tests.golden_io import load_case, load_weights, pixel_list  # noqa: E402
from tests.test_gpu_fuzz import fuzz_scene  # noqa: E402

set_mask = kl._default.cdll.kpn_probe_set_mask
set_mask.argtypes, set_mask.restype = [ctypes.c_uint64], ctypes.c_int

ROWS = ["layers1.0", "layers1.1", "layers1.2", "layers1.3", "layers1.0 keypoint-encoding columns", "layers1.0 sampled-channel columns"]
POINT = ["layers2.0", "layers2.1", "layers2.2", "ibr_compress_gfeat", "ray_encoder.0", "ray_encoder.2", "base_layer.0 (mean|var columns)",
         "base_layer.0 (x columns)", "base_layer.2", "vis_layer1.0", "vis_layer1.2", "vis_layer2.0", "out_layer.0", "out_layer.2"]
VARIANTS = [("shipped (two pieces everywhere)", 0)]
for i, n in enumerate(ROWS):
    VARIANTS += [(f"rows kernel {n}: B one piece", 1 << (2 * i)), (f"rows kernel {n}: A one piece", 1 << (2 * i + 1))]
for i, n in enumerate(POINT):
    VARIANTS += [(f"per-point kernel {n}: B one piece", 1 << (2 * (8 + i))), (f"per-point kernel {n}: A one piece", 1 << (2 * (8 + i) + 1))]
ALL_B_ROWS = sum(1 << (2 * i) for i in range(4))
VARIANTS += [("rows kernel, every layer: B one piece", ALL_B_ROWS), ("rows kernel, every layer: A one piece", ALL_B_ROWS << 1)]


def scenes(n_random):
    """-> list of (name, scene_cpu, sd, pix, Sc, Sf, fine)"""
    out = []
    wref = load_weights()
    for case, fine, wfile in (("case_p_v3_headline_tile", True, None), ("case_q_v10_flat128_chunk", False, None),
                              ("case_s_v3_real_encoder_maps", True, None), ("case_t_v3_trained_tile", True, "weights_trained_seed0.npz")):
        path = os.path.join(ROOT, "tests", "golden", case + ".npz")
        if not os.path.exists(path):
            continue
        scene, cfg, g = load_case(case)
        sd = wref
        if wfile:
            z = np.load(os.path.join(ROOT, "tests", "golden", wfile))
            sd = {k: torch.from_numpy(z[k]) for k in z.files}
        pix, (h, w) = pixel_list(cfg, scene["cam_tar"])
        if pix.shape[0] > 1024:                  # every second row and column of the tile's lattice
            pix = np.ascontiguousarray(pix.reshape(h, w, 2)[::2, ::2].reshape(-1, 2))
        out.append((case, scene, sd, pix, cfg["Sc"], cfg["Sf"], fine))
    rng = np.random.default_rng(7)
    for i in range(n_random):
        c = fuzz_scene(rng)
        sd = random_hotpath_state_dict(seed=c["seed"], density_bias=c["bias"])
        scene = make_scene(n_views=c["V"], src_hw=c["src"], tar_hw=c["tar"], mask=c["mask"], seed=c["seed"] + 1, tar_angle=c["angle"], tar_focal_at_512=c["focal"])
        th, tw = c["tar"]
        yy, xx = np.meshgrid(np.arange(th), np.arange(tw), indexing="ij")
        out.append((f"random {i}", scene, sd, np.stack([xx.reshape(-1), yy.reshape(-1)], -1).astype(np.int32), c["Sc"], c["Sf"], c["fine"]))
    return out


def main():
    n_random = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    t0 = time.time()
    keys_of = lambda fine: ("tex_fg", "alpha") + (("tex_fg_fine", "alpha_fine") if fine else ())
    per = {name: {"max": 0.0, "errs": []} for name, _ in VARIANTS}
    base_err = []
    nrays = 0
    for sname, scene, sd, pix, Sc, Sf, fine in scenes(n_random):
        osc, wf = oracle.OracleScene(scene), oracle.flat_weights(sd)
        ref = oracle.render_rays(osc, wf, scene["cam_tar"], scene["bounds"], pix, Sc, Sf, fine=fine)
        s = to_device(scene, "cuda")
        ps = ops.PreparedScene(s["img"], s["cam"], s["feat_geo"], s["feat_tex"], s["sp_data"], s["src_foreground_mask"])
        w = ops.PackedWeights(sd)
        # the rays as 1 x R "grids" are not a grid: render the bounding lattice once per variant and pick the pixels
        x0, y0, x1, y1 = pix[:, 0].min(), pix[:, 1].min(), pix[:, 0].max(), pix[:, 1].max()
        dx = np.diff(np.unique(pix[:, 0])).min() if len(np.unique(pix[:, 0])) > 1 else 1
        dy = np.diff(np.unique(pix[:, 1])).min() if len(np.unique(pix[:, 1])) > 1 else 1
        assert dx == dy
        nx, ny = (x1 - x0) // dx + 1, (y1 - y0) // dy + 1
        ix, iy = (pix[:, 0] - x0) // dx, (pix[:, 1] - y0) // dy
        plan = ops.RenderPlan(ps, (int(x0), int(y0), int(dx), int(nx), int(ny)), Sc, Sf, fine=fine)
        nrays += pix.shape[0]
        for vname, mask in VARIANTS:
            assert set_mask(mask) == 0
            o = ops.render_rays(ps, w, s["cam_tar"], s["bounds"], plan=plan)
            e = np.zeros(pix.shape[0], np.float32)
            for k in keys_of(fine):
                a = o[k][0].cpu().numpy()
                got = a[:, iy, ix].T if a.ndim == 3 else a[iy, ix]
                d = np.abs(got - ref[k])
                e = np.maximum(e, d.max(-1) if d.ndim == 2 else d)
            per[vname]["errs"].append(e)
        set_mask(0)
        print(f"{sname}: {pix.shape[0]} rays, shipped max {per[VARIANTS[0][0]]['errs'][-1].max():.2e}, {time.time() - t0:.0f} s", flush=True)
    base = np.concatenate(per[VARIANTS[0][0]]["errs"])
    well = base <= 2.5e-5                        # the rays on which the shipped arithmetic itself has the 4x margin
    rows = []
    for vname, mask in VARIANTS:
        e = np.concatenate(per[vname]["errs"])
        rows.append({"variant": vname, "mask": hex(mask), "max_err": float(e.max()), "max_err_on_rays_where_shipped_has_4x_margin": float(e[well].max()),
                     "p999": float(np.quantile(e, 0.999)), "rays_above_2.5e-5": int((e > 2.5e-5).sum()), "rays_above_1e-4": int((e > 1e-4).sum()),
                     "new_rays_above_1e-4": int(((e > 1e-4) & ~(base > 1e-4)).sum()),
                     "qualifies": bool(e[well].max() <= 2.5e-5 and ((e > 1e-4) & ~(base > 1e-4)).sum() == 0)})
    res = {"rays": int(nrays), "random_scenes": n_random, "seconds": time.time() - t0, "rows": rows,
           "what": "per (layer, operand): worst |HIP - oracle| over RGB and alpha, coarse and fine, with that operand's lo fp16 piece replaced by zero"}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "precision_budget.json"), "w"), indent=1)
    md = ["| variant | max err | max err where shipped <= 2.5e-5 | p99.9 | rays > 2.5e-5 | rays > 1e-4 | new rays > 1e-4 | qualifies |", "|---|---|---|---|---|---|---|---|"]
    for r in rows:
        md.append(f"| {r['variant']} | {r['max_err']:.2e} | {r['max_err_on_rays_where_shipped_has_4x_margin']:.2e} | {r['p999']:.2e} | {r['rays_above_2.5e-5']} | "
                  f"{r['rays_above_1e-4']} | {r['new_rays_above_1e-4']} | {'YES' if r['qualifies'] else 'no'} |")
    open(os.path.join(ROOT, "gpurun_out", "precision_budget.md"), "w").write(f"{nrays} rays, {n_random} random scenes + the reference fixtures\n\n" + "\n".join(md) + "\n")
    print("\n".join(md))


if __name__ == "__main__":
    main()
