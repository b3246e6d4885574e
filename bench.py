#!/usr/bin/env python
"""bench.py — rays/s of the ray-march path on MI355X (contract: see the task statement / DESIGN.md §6).

A "step" is one pass of the hot path over one batch of synthetic input = ONE full novel-view frame of
BASELINE.json configs[1]: 512x512 target, 3 source views (512x512), 64 coarse + 64 fine samples per
ray (the reference evaluates the field 192 times per ray; here 128: the fine pass takes the coarse samples' values from the
coarse pass, bit-identical outputs), synthetic seeded scene + random-init hot-path weights.  Inside a
step: scene preparation (NCHW -> channels-last), ray set-up, coarse field pass, compositing, importance
resampling, fine field pass, compositing, planar image write.  Inputs are resident in HBM before the
timed region.  With --gpus N every rank renders its own target camera per step (frames of a render job
shard over ranks, weak scaling) and the finished RGB images are gathered to rank 0 over RCCL; `--gpus N` run
directly (no launcher) starts the N ranks itself.

Prints ONE JSON line (rank 0).  Extra objects: "roofline" for the dominant kernel k_geo_rows (fp32
MFMA bound; duration measured live with HIP events on the launch stream, see kpn_profile_*), and
"cpu_baseline": the CPU oracle ("port") timed on this host's cores on a bounded sample of the same
workload (a strided sub-lattice of the same frame sized for ~15 s of CPU work).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
# split-operand rows kernels: N 16-bit MFMAs (2.5 PFLOP/s dense peak, bf16 and fp16 alike) per fp32 product term set ->
# fp32-equivalent roof = 2500 / N: mode 2 (three bf16 pieces) N = 6, mode 3 (two fp16 pieces, the default) N = 3 (hh hl lh; round 4
# until its last build: 4, roof 625 — the products' count went down, the algorithmic FLOP did not)
SPLIT_PRODUCTS = {2: 6, 3: 3}
ROWS_KERNEL_BASE = {0: "k_geo_rows", 2: "k_geo_rows_h2", 3: "k_geo_rows_f2"}


class _RowsKernel(dict):
    """name of the rows kernel a render pass launches: the pair-tile kernels pool over the views themselves in the render passes
    (suffix p, POOL layout of the scratch) unless KPN_NO_POOL=1"""
    def __getitem__(self, mode):
        pool = mode >= 2 and not (os.environ.get("KPN_NO_POOL") and int(os.environ["KPN_NO_POOL"]))
        return ROWS_KERNEL_BASE[mode] + ("p" if pool else "")


ROWS_KERNEL = _RowsKernel()
ROWS_DTYPE = {0: "f32",
              2: "f32 (dominant kernel: every fp32 operand as three bf16 pieces, six bf16-MFMA products per term set, fp32 accumulation: "
                 "all terms above 2^-24 relative kept; everything else fp32)",
              3: "f32 (dominant kernel: every fp32 operand as two fp16 pieces (22 bits, the residual formed exactly), three fp16-MFMA "
                 "products per term set (hh hl lh: all terms above 2^-24 relative), fp32 accumulation; everything else fp32)"}


def rows_peak_tflops(mode):
    return FP32_MFMA_PEAK_TFLOPS if mode == 0 else 2500.0 / SPLIT_PRODUCTS[mode]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--views", type=int, default=3)
    ap.add_argument("--samples", type=int, default=64, help="coarse = fine samples per ray")
    ap.add_argument("--mask", default="ellipsoid", choices=["ellipsoid", "dense"])
    ap.add_argument("--tar-focal", type=float, default=800.0,
                    help="target camera focal length in px at 512 (800 = the subject framed like the reference's orbit, "
                         "31 %% of the field evaluations valid; 600 = round 1's scene, 17 %%)")
    ap.add_argument("--density-bias", type=float, default=0.0,
                    help="added to the bias of the density output (0 = every valid point has density > 0, as with the seeded "
                         "random weights; -20 = about half of the visual hull is empty, as with a trained density)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary (dense mask / round-1 scene / training) results")
    ap.add_argument("--chunk-rays", type=int, default=0)
    ap.add_argument("--no-fine", action="store_true", help="flat sampling: coarse pass only (BASELINE configs[4] style)")
    ap.add_argument("--geo-rows-mode", type=int, default=3, choices=[0, 2, 3],
                    help="rows kernel of the field's first MLP: 3 = two fp16 pieces per operand, three products on the fp16 MFMA, two "
                         "tiles per wave, one wave per SIMD (the library's default: fp32-class results); 2 = three bf16 pieces, six "
                         "products (fp32's exponent range); 0 = fp32 MFMA")
    ap.add_argument("--no-coarse-reuse", action="store_true",
                    help="evaluate the field at all Sc+Sf merged samples in the fine pass, as the reference does (default: the "
                         "coarse samples' values are taken from the coarse pass: bit-identical outputs, Sc+Sf instead of "
                         "2*Sc+Sf evaluations per ray)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (production); gloo + all ranks on cuda:0 only to exercise the N>1 flow on a 1-GPU box")
    ap.add_argument("--cpu-sample-rays", type=int, default=0, help="0 = size the CPU sample for ~15 s")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1: weak = every rank renders its own frame of the orbit per step (frames sharded, the production job); "
                         "strong = ONE frame per step, its rows dealt round-robin to the N ranks (SURVEY 8(e))")
    ap.add_argument("--force-dist", action="store_true", help="N = 1: run the N > 1 code path anyway (process group of one rank, "
                    "FrameGatherer, barriers, the MAX all-reduce of the time) — what one GPU allows of the RCCL branch; "
                    "needs MASTER_ADDR / MASTER_PORT (RANK / WORLD_SIZE default to 0 / 1)")
    ap.add_argument("--sync-gather", action="store_true", help="N > 1: gather each frame before the next one starts (default: the "
                    "gather of frame i overlaps the rendering of frame i + 1)")
    ap.add_argument("--no-configs4", action="store_true", help="skip secondary.configs4_full (4096^2 rays, 10 views, 128 flat samples: ~15 s + set-up)")
    return ap.parse_args()


def frame_parity(frame, oracle, osc, wflat, cam_tar, bounds, pix, Sc, Sf, fine, ref, render_one=None):
    """The HIP frame against the oracle's rays at the pixels `pix` (x, y): worst |difference| per output and the verdict of the
    parity gate (tests/parity_gate.py: <= 1e-4 on every ray unless the oracle's own conditioning probe explains the ray; the
    probe is run on the rays above the bar only).  `frame`: {key: (3,H,W) / (H,W) numpy arrays}.  The checker, not the product.
    Round 6: a ray the gate widens counts as explained only if the CONDITIONAL re-check passes for it (parity_gate.recheck_widened:
    each stage of the oracle on the kernels' own inputs of that stage, strict stage bars; render_one() = the product's one-pixel
    render with kpn_render_stages): `conditional_ok`, and per widened ray the stage excesses in `widened_rays`."""
    import numpy as np
    from tests import parity_gate
    keys = [k for k in ("tex_fg", "alpha", "tex_fg_fine", "alpha_fine") if k in ref and k in frame]
    got = {k: (frame[k][:, pix[:, 1], pix[:, 0]].T if frame[k].ndim == 3 else frame[k][pix[:, 1], pix[:, 0]]) for k in keys}
    err = {k: np.abs(got[k] - ref[k]) for k in keys}
    ray_err = {k: (e.max(-1) if e.ndim == 2 else e) for k, e in err.items()}
    above = np.zeros(pix.shape[0], bool)
    for k in keys:
        above |= ~(ray_err[k] <= parity_gate.RGBA_TOL)
    res = {"rays": int(pix.shape[0]), "max_abs_rgb": float(max(np.nanmax(ray_err[k]) for k in keys if k.startswith("tex"))),
           "max_abs_alpha": float(max(np.nanmax(ray_err[k]) for k in keys if k.startswith("alpha"))),
           "rays_above_1e-4": int(above.sum()), "widened": 0, "unexplained": 0, "finite": bool(all(np.isfinite(got[k]).all() for k in keys)),
           "what": "HIP frame vs the C oracle (pinned to the reference's outputs) on these rays; gate = tests/parity_gate.py"}
    if above.any():
        idx = np.nonzero(above)[0]
        sub = {k: got[k][idx] for k in keys}
        subref = {k: ref[k][idx] for k in keys}
        env = parity_gate.oracle_envelope(oracle, osc, wflat, cam_tar, bounds, pix[idx], Sc, Sf, fine=fine)
        try:
            rep = parity_gate.check_rays(sub, subref, env, keys=keys, max_widened_fraction=1.0, what="bench frame")
            res["widened"] = len(rep["widened"])
            if render_one is not None and rep["widened"]:
                parity_gate.recheck_widened(rep, render_one(), oracle, osc, wflat, cam_tar, bounds, pix[idx], Sc, Sf, fine=fine)   # render_one: a factory
                res["conditional_ok"] = bool(rep["conditional_ok"])
                res["widened_rays"] = [{"pixel": [int(v) for v in pix[idx][r["ray"]]], "err": r["err"], "oracle_envelope": r["oracle_envelope"],
                                        "conditional_ok": r["conditional_ok"], "stage_excess_over_bar": r["conditional"]["stages"],
                                        "bin_flips": r["conditional"]["bin_flips"]} for r in rep["widened"]]
            elif rep["widened"]:
                res["conditional_ok"] = None    # not re-checked (no product render available)
        except AssertionError as e:
            res["unexplained"] = int(above.sum())
            res["error"] = str(e)[:400]
    res["ok"] = bool(res["finite"] and res["unexplained"] == 0 and res["widened"] <= max(1, pix.shape[0] // 500))
    return res


def cpu_baseline(args, scene_cpu, sd, target_s=15.0, fine=True, frame=None, render_one=None):
    """Oracle (C restatement, OpenMP over points) on a strided sub-lattice of the SAME frame, sized from a
    short probe so that the timed run is about `target_s` seconds of CPU work.  With `frame` (the HIP frame of the timed
    region, numpy) the oracle's rays are also COMPARED with it: -> (baseline, parity)."""
    import numpy as np
    from oracle import oracle
    osc = oracle.OracleScene(scene_cpu)
    wflat = oracle.flat_weights(sd)

    def lattice(n):
        step = max(1, args.res // n)
        ys, xs = np.meshgrid(np.arange(n) * step, np.arange(n) * step, indexing="ij")
        return np.stack([xs.reshape(-1), ys.reshape(-1)], -1).astype(np.int32), step

    last = {}

    def run(pix):
        t0 = time.perf_counter()
        last["ref"] = oracle.render_rays(osc, wflat, scene_cpu["cam_tar"], scene_cpu["bounds"], pix, args.samples, args.samples, fine=fine)
        return time.perf_counter() - t0

    pix, _ = lattice(32)
    run(pix[:64])                      # warm-up (library load, page-in)
    probe = run(pix)                   # 1024 rays spread over the frame
    n = int(min(args.res, max(32, (1024 * target_s / max(probe, 1e-3)) ** 0.5))) if args.cpu_sample_rays <= 0 else int(args.cpu_sample_rays ** 0.5)
    n = max(32, n // 32 * 32)
    pix, step = lattice(n)
    dt = run(pix)
    base = {"value": pix.shape[0] / dt, "unit": "rays/s", "cores": os.cpu_count(), "kind": "port",
            "sample": f"{pix.shape[0]} rays ({n}x{n} lattice, step {step}) of the same frame, {args.samples}+{args.samples} "
                      f"samples/ray, {dt:.1f} s of wall time, C oracle with OpenMP over points on {os.cpu_count()} hardware threads"}
    parity = None
    if frame is not None:
        parity = frame_parity(frame, oracle, osc, wflat, scene_cpu["cam_tar"], scene_cpu["bounds"], pix, args.samples, args.samples, fine, last["ref"],
                              render_one=render_one)
        parity["sample"] = f"the {pix.shape[0]} rays of the cpu_baseline lattice ({n}x{n}, step {step}) of the timed {args.res}x{args.res} frame"
    return base, parity


def time_frames(L, ops, torch, scene, w, res, samples, fine, steps, warmup=1, with_kernel=False):
    """ms per frame + valid (point, view) rows per frame of one more workload (secondary results)."""
    ps = ops.PreparedScene(scene["img"], scene["cam"], scene["feat_geo"], scene["feat_tex"], scene["sp_data"],
                           scene["src_foreground_mask"])
    plan = ops.RenderPlan(ps, (0, 0, 1, res, res), samples, samples, fine=fine)
    for _ in range(warmup):
        ops.render_rays(ps, w, scene["cam_tar"], scene["bounds"], plan=plan)
        torch.cuda.synchronize()     # (untimed; lets density first's auto mode see this workload's statistics before the timed frames)
    _a, _b = ctypes.c_int64(0), ctypes.c_int64(0)
    L.check(L.kpn_density_first_passes(ctypes.byref(_a), ctypes.byref(_b), 1))   # count the timed frames' passes only
    L.check(L.kpn_profile_enable(1))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        L.check(L.kpn_scene_prepare(ctypes.byref(ps.desc), ctypes.c_void_p(ps.ws.data_ptr()),
                                    ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
        ops.render_rays(ps, w, scene["cam_tar"], scene["bounds"], plan=plan)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    ms, launches, rows = ctypes.c_double(0), ctypes.c_int64(0), ctypes.c_int64(0)
    L.check(L.kpn_profile_collect(ctypes.byref(ms), ctypes.byref(launches), ctypes.byref(rows)))
    L.check(L.kpn_profile_enable(0))
    del plan, ps
    if with_kernel:
        return dt * 1e3, rows.value / steps, ms.value / max(1, launches.value), rows.value * L.kpn_flops_per_row() / max(1e-9, ms.value * 1e-3) / 1e12
    return dt * 1e3, rows.value / steps


def time_configs4(L, ops, torch, dev, sd, mode, with_parity=True):
    """BASELINE configs[4] at FULL size — 4096 x 4096 rays, 10 source views of 4096^2, 128 flat samples per ray (no fine pass),
    dense mask, random weights: the roofline-measurement config.  One timed step (about 1.3e10 rows), the rows
    kernel's launch times taken inside the library like the headline's."""
    from keypointnerf_amd.synthetic import make_scene, to_device
    res, views, samples = 4096, 10, 128
    t_setup = time.perf_counter()
    scene = to_device(make_scene(n_views=views, src_hw=(res, res), tar_hw=(res, res), mask="dense", seed=1, tar_focal_at_512=800.0), dev)
    w = ops.PackedWeights(sd, device=dev)
    ps = ops.PreparedScene(scene["img"], scene["cam"], scene["feat_geo"], scene["feat_tex"], scene["sp_data"], scene["src_foreground_mask"])
    plan = ops.RenderPlan(ps, (0, 0, 1, res, res), samples, samples, fine=False)
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t_setup
    L.check(L.kpn_profile_enable(1))
    t0 = time.perf_counter()
    out = ops.render_rays(ps, w, scene["cam_tar"], scene["bounds"], plan=plan)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms, launches, rows, surplus = ctypes.c_double(0), ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0)
    L.check(L.kpn_profile_collect2(ctypes.byref(ms), ctypes.byref(launches), ctypes.byref(rows), ctypes.byref(surplus)))
    L.check(L.kpn_profile_enable(0))
    achieved = rows.value * L.kpn_flops_per_row() / max(1e-9, ms.value * 1e-3) / 1e12
    peak = rows_peak_tflops(mode)
    valid = rows.value / (views * res * res * samples)
    parity = None
    if with_parity:   # 9,216 rays of this very frame against the oracle (a 96 x 96 lattice over the 4096^2 target; 1,024 until round 4)
        try:
            import numpy as np
            from oracle import oracle
            ys, xs = np.meshgrid(np.arange(96) * 42 + 53, np.arange(96) * 42 + 53, indexing="ij")
            pix = np.stack([xs.reshape(-1), ys.reshape(-1)], -1).astype(np.int32)
            scene_cpu = to_device(scene, "cpu")
            osc, wflat = oracle.OracleScene(scene_cpu), oracle.flat_weights(sd)
            ref = oracle.render_rays(osc, wflat, scene_cpu["cam_tar"], scene_cpu["bounds"], pix, samples, samples, fine=False)
            px, py = torch.from_numpy(pix[:, 0]).long().to(dev), torch.from_numpy(pix[:, 1]).long().to(dev)
            frame = {"tex_fg": out["tex_fg"][0][:, py, px].cpu().numpy(), "alpha": out["alpha"][0][py, px].cpu().numpy()}
            keys = ("tex_fg", "alpha")
            got = {"tex_fg": frame["tex_fg"].T, "alpha": frame["alpha"]}
            from tests import parity_gate
            try:
                rep = parity_gate.check_rays(got, ref, parity_gate.oracle_envelope(oracle, osc, wflat, scene_cpu["cam_tar"], scene_cpu["bounds"], pix, samples, samples, fine=False),
                                             keys=keys, max_widened_fraction=0.01, what="configs[4] subset")
                # round 6: widened rays count only if every stage agrees given the kernels' own inputs (strict stage bars)
                parity_gate.recheck_widened(rep, parity_gate.product_render_one(ops, ps, w, scene["cam_tar"], scene["bounds"], samples, samples, fine=False),
                                            oracle, osc, wflat, scene_cpu["cam_tar"], scene_cpu["bounds"], pix, samples, samples, fine=False)
                parity = {"rays": int(pix.shape[0]), "max_abs_rgb": rep["max_err"]["tex_fg"], "max_abs_alpha": rep["max_err"]["alpha"],
                          "rays_above_1e-4": rep["above_bar"], "widened": len(rep["widened"]), "unexplained": 0, "ok": True,
                          "conditional_ok": rep.get("conditional_ok"),
                          "widened_rays": [{"pixel": [int(v) for v in pix[r["ray"]]], "err": r["err"], "stage_excess_over_bar": r["conditional"]["stages"]} for r in rep["widened"]]}
            except AssertionError as e:
                parity = {"rays": int(pix.shape[0]), "ok": False, "error": str(e)[:400]}
            parity["sample"] = "9,216 rays (96 x 96 lattice, step 42) of this 4096 x 4096 frame vs the C oracle, V = 10, 128 flat samples"
            del scene_cpu, osc
        except Exception as e:  # noqa: BLE001
            parity = {"ok": False, "error": f"{type(e).__name__}: {e}"}
    r = {"workload": "configs[4]: 4096x4096 rays, 10 source views 4096x4096, 128 flat samples/ray, dense mask, random weights; one step, no warm-up",
         "ms_per_step": dt * 1e3, "rays_per_sec": res * res / dt, "sampled_points_per_sec": res * res * samples / dt,
         "fully_evaluated_points_per_sec": res * res * samples * valid / dt, "valid_fraction_of_field_evaluations": valid,
         "valid_rows": rows.value, "render_workspace_bytes": plan.nbytes, "mean_alpha": float(out["alpha"].mean()), "setup_s": t_setup, "parity": parity,
         "roofline": {"kernel": ROWS_KERNEL[mode], "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                      "launches": launches.value, "avg_launch_ms": ms.value / max(1, launches.value), "kernel_time_share": ms.value * 1e-3 / dt}}
    del plan, ps, scene, out
    torch.cuda.empty_cache()
    return r


def time_training(ops, torch, dev, sd, steps=10, patch=32):
    """BASELINE configs[3], field part: train-branch forward + backward in HIP for patch x patch rays x (64 + 128) samples, V=3
    (kpn_render_rays_train + kpn_render_rays_train_backward), inputs resident, random draws prepared outside the timed
    region.  patch = 32: 1024 rays (BASELINE configs[3]); 64: the reference's shipped patch (configs/zju.json:36-37).
    Returns ms per forward and per backward."""
    from keypointnerf_amd.synthetic import make_scene, to_device
    scene = to_device(make_scene(n_views=3, src_hw=(512, 512), tar_hw=(512, 512), mask="ellipsoid", seed=1,
                                 tar_focal_at_512=800.0), dev)
    ps = ops.PreparedScene(scene["img"], scene["cam"], scene["feat_geo"], scene["feat_tex"], scene["sp_data"],
                           scene["src_foreground_mask"])
    w = ops.PackedWeights(sd, device=dev)
    R, Sc, Sf = patch * patch, 64, 64
    g = torch.Generator(device=dev).manual_seed(0)
    yy, xx = torch.meshgrid(torch.arange(patch, device=dev), torch.arange(patch, device=dev), indexing="ij")
    pix = torch.stack([xx.reshape(-1) + 256 - patch // 2, yy.reshape(-1) + 256 - patch // 2], -1).to(torch.int32)
    u_c, u_f = torch.rand(R, Sc, device=dev, generator=g), torch.rand(R, Sf, device=dev, generator=g)
    n_c, n_f = torch.randn(R * Sc, device=dev, generator=g), torch.randn(R * (Sc + Sf), device=dev, generator=g)
    grads = {"tex_fg": torch.randn(1, 3, R, device=dev, generator=g) / R, "tex_fg_fine": torch.randn(1, 3, R, device=dev, generator=g) / R}
    kw = dict(noise_coarse=n_c, noise_fine=n_f, rand_noise_std=0.01, n_coarse=Sc, n_fine=Sf)
    kept = {}

    def fwd():
        kept["state"] = ops.render_rays_train(ps, w, scene["cam_tar"], scene["bounds"], pix, u_c, u_f, 0b111, 0b101, keep_state=True, **kw)[1]
    bwd = lambda: ops.render_rays_train_backward(ps, w, scene["cam_tar"], scene["bounds"], pix, u_c, u_f, 0b111, 0b101, grads,
                                                 state=kept["state"], **kw)
    bwd_classic = lambda: ops.render_rays_train_backward(ps, w, scene["cam_tar"], scene["bounds"], pix, u_c, u_f, 0b111, 0b101, grads, **kw)
    out = {}
    from keypointnerf_amd import lib as kl
    L = kl.get_library()
    for name, fn in (("forward_ms", fwd), ("backward_ms", bwd), ("backward_repeating_the_forward_ms", bwd_classic)):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        out[name] = (time.perf_counter() - t0) / steps * 1e3
        if name == "backward_ms":       # the same iterations once more with the library's event brackets on (not inside the timed loop)
            L.check(L.kpn_bwd_profile_enable(1))
            for _ in range(steps):
                fn()
            torch.cuda.synchronize()
            out["backward_roofline"] = backward_roofline(L, steps)
            L.check(L.kpn_bwd_profile_enable(0))
    out["iterations_per_sec"] = 1e3 / (out["forward_ms"] + out["backward_ms"])
    out["rays_per_sec"] = R * out["iterations_per_sec"]
    out["workload"] = (f"configs[3] field part: {R} rays x (64 coarse + 128 fine-pass) evaluations, V=3, view dropout + density noise, "
                       "fwd (kpn_render_rays_train_keep) + bwd (kpn_render_rays_train_backward_kept) in HIP")
    return out


def backward_roofline(L, steps):
    """Per kernel group of the training backward (HIP events inside the library, kpn_bwd_profile_*): time per iteration, achieved
    TFLOP/s against the roof of the arithmetic it runs on, and the bytes of layer inputs / output gradients it dumps or reads back
    against HBM.  FLOP models, 2 x MAC (layer shapes: SURVEY Appendix A; reference src/utils.py:691-720, src/model.py:1267-1302):
      k_geo_rows_bwd   per (point, view) row: recompute layers1.0-1.2 (29,696 + 16,384 + 16,320) + dX = W^T dY of layers1.3, 1.2, 1.1
                       and the 64 sampled-channel columns of layers1.0 (7,680 + 16,320 + 16,384 + 8,192) = 110,976 MAC; three bf16
                       pieces per operand, six products: roof 2500 / 6 = 416.7 fp32-equivalent TFLOP/s
      k_weight_grad    dW = dY^T X of every layer: 70,080 (layers1) + 13,256 (colour head) MAC per row + 15,488 (layers2 + compress)
                       per point; three bf16 pieces, six products: roof 416.7
      k_color_bwd      per kept row: the head's recompute + its reverse (2 x 13,256 MAC), + compress per point (3,072): fp32 MFMA, roof 157.3
      k_fuse_bwd       per point: layers2 recompute + reverse (2 x 12,416) + the compress reverse (3,072): fp32 MFMA, roof 157.3
    Dumps: k_geo_rows_bwd writes 4,288 B per row (X0..X3, D0..D3), k_color_bwd 2,920 B per kept row; k_weight_grad reads both once."""
    ms, n = (ctypes.c_double * 5)(), (ctypes.c_int64 * 5)()
    rows, kept, pts = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0)
    L.check(L.kpn_bwd_profile_collect(ms, n, ctypes.byref(rows), ctypes.byref(kept), ctypes.byref(pts)))
    rows, kept, pts = rows.value / steps, kept.value / steps, pts.value / steps
    flop = {"k_geo_rows_bwd": (3, 2.0 * 110976 * rows, 2500.0 / 6, 4288.0 * rows, "written"),
            "k_weight_grad": (4, 2.0 * ((70080 + 13256) * rows + 15488 * pts), 2500.0 / 6, 4288.0 * rows + 2920.0 * kept, "read"),
            "k_color_bwd": (1, 2.0 * (2 * 13256 * kept + 3072 * pts), FP32_MFMA_PEAK_TFLOPS, 2920.0 * kept, "written"),
            "k_fuse_bwd": (2, 2.0 * ((2 * 12416 + 3072) * pts), FP32_MFMA_PEAK_TFLOPS, None, None)}
    res = {"rows_per_iteration": rows, "kept_rows_per_iteration": kept, "valid_points_per_iteration": pts, "kernels": {}}
    for k, (i, f, peak, dump, how) in flop.items():
        t = ms[i] / steps
        e = {"ms_per_iteration": t, "achieved": f / max(t * 1e-3, 1e-12) / 1e12, "peak": peak, "unit": "TFLOP/s",
             "frac": f / max(t * 1e-3, 1e-12) / 1e12 / peak, "bound": "mfma"}
        if dump:
            e["dump_bytes_" + how] = dump
            e["dump_GBps"] = dump / max(t * 1e-3, 1e-12) / 1e9
            e["dump_frac_of_hbm_8TBps"] = e["dump_GBps"] / 8000.0
        res["kernels"][k] = e
    res["sum_of_kernels_ms"] = sum(ms[i] for i in range(5)) / steps
    return res


def launch_ranks(args):
    """`python bench.py --gpus N` outside a launcher: start the N ranks ourselves (one process per GPU) under
    torch.distributed.run on 127.0.0.1 and hand its exit status back.  Under an external launcher (the driver's
    `python -m torch.distributed.run ... bench.py --gpus N`) WORLD_SIZE is already set and this is skipped."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: required for RCCL between processes on this driver
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} rank(s); refusing to report a "
                 f"number for a different GPU count")
    multi = world > 1 or args.force_dist          # the distributed code path (world 1 with --force-dist: a process group of one rank)
    if multi:
        import torch.distributed as dist
        if args.force_dist and world == 1:
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        if args.dist_backend == "nccl":
            if torch.cuda.device_count() < world:
                sys.exit(f"bench.py: --gpus {world} needs {world} visible GPUs (found {torch.cuda.device_count()}); "
                         f"--dist-backend gloo runs all ranks on cuda:0 for flow tests only")
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        assert dist.get_world_size() == world
    else:
        dist = None
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())

    from keypointnerf_amd import lib as kl
    from keypointnerf_amd import ops
    from keypointnerf_amd.parallel import FrameGatherer, orbit_target_camera, rows_of_rank
    from keypointnerf_amd.synthetic import make_scene, random_hotpath_state_dict, to_device

    if args.no_coarse_reuse:
        os.environ["KPN_NO_COARSE_REUSE"] = "1"
    else:
        os.environ.pop("KPN_NO_COARSE_REUSE", None)
    L = kl.get_library()
    L.check(L.kpn_set_geo_rows_mode(args.geo_rows_mode))
    sd = random_hotpath_state_dict(seed=3, density_bias=args.density_bias)
    res = args.res
    scene_cpu = make_scene(n_views=args.views, src_hw=(res, res), tar_hw=(res, res), mask=args.mask, seed=1,
                           tar_focal_at_512=args.tar_focal)
    scene = to_device(scene_cpu, dev)
    w = ops.PackedWeights(sd, device=dev)
    ps = ops.PreparedScene(scene["img"], scene["cam"], scene["feat_geo"], scene["feat_tex"], scene["sp_data"],
                           scene["src_foreground_mask"])
    fine = not args.no_fine
    ref_evals_per_ray = args.samples * (3 if fine else 1)          # the reference: Sc coarse + (Sc + Sf) fine
    evals_per_ray = args.samples * (2 if fine and not args.no_coarse_reuse else (3 if fine else 1))   # performed here
    strong = world > 1 and args.scaling == "strong"
    # strong: row y of the frame -> rank y mod world (interleaved: every rank gets the same share of the subject)
    y0, step_y, nrows = rows_of_rank(res, rank, world) if strong else (0, 0, res)
    if strong and res % world:
        sys.exit("bench.py: --scaling strong needs the frame height to be a multiple of the rank count (equal bands gather into one frame)")
    plan = ops.RenderPlan(ps, (0, y0, 1, res, nrows, step_y), args.samples, args.samples, fine=fine, chunk_rays=args.chunk_rays)
    gdev = dev if args.dist_backend == "nccl" else torch.device("cpu")
    gatherer = FrameGatherer(world, rank, (3, nrows, res), device=gdev) if multi else None

    def step(i):
        # weak: frame i of the job, rank r renders target camera (i*world + r) of the orbit; strong: camera i, band r of its rows
        cam_tar = orbit_target_camera(scene["cam_tar"], i if strong else i * world + rank) if multi else scene["cam_tar"]
        L.check(L.kpn_scene_prepare(ctypes.byref(ps.desc), ctypes.c_void_p(ps.ws.data_ptr()),
                                    ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
        out = ops.render_rays(ps, w, cam_tar, scene["bounds"], plan=plan)
        if multi:  # the job's only exchange: finished RGB frames (3 MB each) / bands gathered to rank 0, asynchronously
            k = gatherer.submit(out["tex_fg_fine" if fine else "tex_fg"][0])
            if args.sync_gather and gatherer.work[k] is not None:
                gatherer.work[k].wait()
                if args.dist_backend == "nccl":
                    torch.cuda.current_stream().synchronize()
        return out

    for i in range(args.warmup):
        step(i)
    L.check(L.kpn_profile_enable(1))
    dens = [ctypes.c_int64(0), ctypes.c_int64(0)]   # points in the hull whose density was evaluated / of which live (relu(rad) > 0)
    dpasses = [ctypes.c_int64(0), ctypes.c_int64(0)]   # timed render passes that ran density first / on the fused per-point kernel
    L.check(L.kpn_density_first_passes(ctypes.byref(dpasses[0]), ctypes.byref(dpasses[1]), 1))
    L.check(L.kpn_density_stats(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), ctypes.byref(dens[0]), ctypes.byref(dens[1]), 1))
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step(args.warmup + i)
    if multi:
        gatherer.finish()                     # the last frames' gathers are inside the timed region
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # the frame of the last timed step, for the comparison with the oracle further down (world 1: rank 0's own camera)
    frame_np = {k: v[0].cpu().numpy() for k, v in out.items() if k in ("tex_fg", "alpha", "tex_fg_fine", "alpha_fine")} if (world == 1 and rank == 0 and not multi) else None
    gather_ms = None
    if multi:                             # the exchange alone (after the timed region): one synchronous round, averaged
        img = out["tex_fg_fine" if fine else "tex_fg"][0]
        torch.cuda.synchronize(); dist.barrier()
        tg = time.perf_counter()
        for _ in range(5):
            wk = gatherer.work[gatherer.submit(img)]
            if wk is not None:
                wk.wait()
            torch.cuda.synchronize()
        dist.barrier()
        gather_ms = (time.perf_counter() - tg) / 5 * 1e3
        gatherer.finish()
    ms, launches, rows, surplus, clock_ghz = ctypes.c_double(0), ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_double(0)
    # launch times (HIP events) and the shader clock the chip sustained under the rows kernel (s_memtime stamps of its first
    # workgroup against the same launches' event time: include/kpnerf.h kpn_profile_collect3)
    L.check(L.kpn_profile_collect3(ctypes.byref(ms), ctypes.byref(launches), ctypes.byref(rows), ctypes.byref(surplus), ctypes.byref(clock_ghz)))
    L.check(L.kpn_profile_enable(0))
    L.check(L.kpn_density_stats(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), ctypes.byref(dens[0]), ctypes.byref(dens[1]), 1))
    L.check(L.kpn_density_first_passes(ctypes.byref(dpasses[0]), ctypes.byref(dpasses[1]), 1))
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
    if multi:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    rays_per_step = res * res
    value = (1 if strong else world) * rays_per_step * args.steps / dt
    rccl = None
    if multi and args.dist_backend == "nccl":
        try:
            rccl = "RCCL " + ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception as e:  # noqa: BLE001
            rccl = f"unknown ({e})"
    if rank == 0:
        flops_row = L.kpn_flops_per_row()
        achieved = (rows.value * flops_row) / (ms.value * 1e-3) / 1e12 if ms.value > 0 else 0.0
        # HBM bytes per launch of k_geo_rows: PMC-measured bytes per row (profiles/geo_rows_traffic.json, from
        # separate rocprofv3 --pmc passes, FETCH_SIZE doubled per the gfx950 note) x rows per launch of THIS run
        traffic = None
        tj = os.path.join(ROOT, "profiles", "geo_rows_traffic.json")
        if os.path.exists(tj) and launches.value > 0:
            tr = json.load(open(tj))
            entry = tr.get(ROWS_KERNEL[args.geo_rows_mode]) if args.geo_rows_mode != 0 else tr
            if entry:
                traffic = entry["hbm_bytes_per_row"] * rows.value / launches.value
        alpha_mean = float(out["alpha_fine" if fine else "alpha"].mean())
        rows_per_step = rows.value / max(1, args.steps)
        valid_frac = rows_per_step / (args.views * rays_per_step * evals_per_ray)
        torch.cuda.synchronize()
        peak_alloc = torch.cuda.max_memory_allocated(dev)
        peak = rows_peak_tflops(args.geo_rows_mode)
        line = {
            "metric": f"rendered rays/sec ({args.samples} coarse" + (f" + {args.samples} fine samples/ray)" if fine else " samples/ray, flat)"),
            "value": value, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            # of the points inside the visual hull (all of which go through layers1 / layers2): the fraction with relu(rad) == 0, whose
            # colour head the density-first passes skip (exact: such a sample's compositing weight is 0; include/kpnerf.h)
            "sigma_zero_fraction": (1.0 - dens[1].value / dens[0].value) if dens[0].value > 0 else None,
            "density_first": {"mode": {0: "never", 1: "always", 2: "auto"}[L.kpn_get_density_first()],
                              "passes_density_first": dpasses[0].value, "passes_fused_kernel": dpasses[1].value,
                              "note": "per-point part of a render pass as density pass + colour of the live points only, or the fused kernel; "
                                      "auto = from the dead fraction earlier passes measured on the device (>= 20 %: density first); bit-identical frames"},
            "dtype": ROWS_DTYPE[args.geo_rows_mode],
            "data": "synthetic",
            "config": {"workload": f"{'configs[1]' if (fine and args.views == 3 and args.samples == 64 and res == 512) else 'custom'}: "
                                   f"{res}x{res} novel view, {args.views} source views {res}x{res}, "
                                   f"{args.samples} coarse" + (f" + {args.samples} fine" if fine else "") + f" samples/ray, {args.mask} fg mask, "
                                   f"seeded synthetic scene, random-init hot-path weights",
                       "rays_per_step": rays_per_step, "field_evals_per_ray": evals_per_ray,
                       "reference_field_evals_per_ray": ref_evals_per_ray,
                       "note": ("fine pass re-uses the coarse samples' field values (same points, bit-identical outputs; "
                                "tests/test_gpu_parity.py::test_fine_pass_reuses_coarse_values_bit_exactly); "
                                "--no-coarse-reuse evaluates them again like the reference") if evals_per_ray != ref_evals_per_ray else "",
                       "sampled_points_per_sec": value * evals_per_ray,
                       "valid_rows_per_step": rows_per_step,
                       "valid_fraction_of_field_evaluations": valid_frac,
                       "fully_evaluated_points_per_sec": value * evals_per_ray * valid_frac,
                       "render_workspace_bytes": plan.nbytes, "scene_workspace_bytes": ps.ws.numel() * 4,
                       "peak_device_bytes_allocated": peak_alloc,
                       # batches of rows the fp32-range kernels evaluated again (range guard of the two-fp16-piece kernels): 0 = the
                       # whole run stayed on the default kernels
                       "range_guard_batches_redone": ops.range_guard_count(), "range_guard": bool(L.kpn_get_range_guard()),
                       "mean_alpha_fine": alpha_mean, "parallelism": (f"one frame, rows dealt round-robin to {world} ranks (row y -> rank y mod {world}), de-interleaved on rank 0" if strong else f"frames sharded over {world} rank(s)") + ", one process per GPU"
                                      + (f", {args.dist_backend} gather of finished frames to rank 0" if multi else ""),
                       "dist_world_size": (dist.get_world_size() if multi else 1),
                       "dist_backend": (args.dist_backend if multi else None),
                       "rccl_ranks": (dist.get_world_size() if (multi and args.dist_backend == "nccl") else None),
                       "dist_library": rccl,
                       "gather": (None if not multi else ("synchronous per frame" if (args.sync_gather or not gatherer.asynchronous) else
                                  "asynchronous: frame i travels while frame i+1 renders (two staging buffers)")),
                       "gather_ms_per_round_alone": gather_ms},
            "roofline": {"kernel": ROWS_KERNEL[args.geo_rows_mode], "bound": "mfma", "achieved": achieved,
                         "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic,
                         "achieved_over_fp32_mfma_peak": achieved / FP32_MFMA_PEAK_TFLOPS,
                         # written per row: mode 0 the 64-vector + the gather record (320 B); pair kernels without pooling the 64-vector
                         # (256 B); with pooling inside the kernel (POOL) 128 floats per POINT = 512 / V bytes per row
                         "algorithmic_bytes_per_launch": (320.0 if args.geo_rows_mode == 0 else (512.0 / args.views if ROWS_KERNEL[args.geo_rows_mode].endswith("p") else 256.0)) * rows.value / max(1, launches.value),
                         "launches": launches.value, "avg_launch_ms": ms.value / max(1, launches.value),
                         "surplus_launches": surplus.value,
                         "traffic_source": "profiles/geo_rows_traffic.json: PMC (FETCH_SIZE, WRITE_SIZE) bytes per row from separate rocprofv3 --pmc passes of this kernel, times this run's rows per launch" if traffic is not None else None,
                         "note": "achieved = algorithmic fp32 FLOP (140,160 per row) / kernel time; in the split-operand modes the matrix pipe executes N 16-bit products per fp32 product term set (mode 3: N = 3, mode 2: N = 6), so the roof is the dense 16-bit peak / N; the row scratch between the rows kernel and k_fuse_color is capped ("
                                 + f"{L.kpn_row_scratch_cap_bytes() / 2**30:.1f} GiB) and reused by batches of a pass; the worst-case "
                                 "number of batches is launched and the surplus ones return at once (a few us each): "
                                 "`launches` / `avg_launch_ms` cover the launches that processed rows, a rocprofv3 average "
                                 "over ALL k_geo_rows launches is lower by the factor launches / (launches + surplus_launches)",
                         "algorithmic_flop_per_row": flops_row,
                         # what the chip clocked at under this load, and the fraction against the roof at THAT clock (the peaks of
                         # MI355X_MICROARCH.md are quoted at 2.4 GHz)
                         "effective_clock_ghz": clock_ghz.value or None,
                         "frac_at_sustained_clock": (achieved / (peak * clock_ghz.value / 2.4)) if clock_ghz.value > 0 else None,
                         "clock_note": "shader cycles (s_memtime) of the rows kernel's first workgroup / the launches' HIP-event time; the peaks are quoted at 2.4 GHz, the chip clocks to its power budget",
                         "kernel_time_share": (ms.value * 1e-3) / dt},
        }
        if world == 1 and not args.no_secondary:
            sec = {}
            # the other mask kind on the same cameras, and round 1's lighter framing (tar focal 600: 17 % valid)
            other = "dense" if args.mask == "ellipsoid" else "ellipsoid"
            for name, kw in ((f"{other}_mask", dict(mask=other, tar_focal_at_512=args.tar_focal)),
                             ("round1_scene_ellipsoid_focal600", dict(mask="ellipsoid", tar_focal_at_512=600.0))):
                sc2 = to_device(make_scene(n_views=args.views, src_hw=(res, res), tar_hw=(res, res), seed=1, **kw), dev)
                ms2, rows2, kms, ktf = time_frames(L, ops, torch, sc2, w, res, args.samples, fine, steps=max(2, min(args.steps, 5)), with_kernel=True)
                sec[name] = {"ms_per_frame": ms2, "rays_per_sec": rays_per_step / (ms2 * 1e-3),
                             "valid_fraction_of_field_evaluations": rows2 / (args.views * rays_per_step * evals_per_ray),
                             "roofline": {"kernel": ROWS_KERNEL[args.geo_rows_mode], "bound": "mfma", "achieved": ktf, "peak": peak,
                                          "unit": "TFLOP/s", "frac": ktf / peak, "avg_launch_ms": kms}}
                del sc2
            if args.density_bias == 0.0:
                # a density that is exactly 0 in about half of the visual hull (trained models: the free space between the
                # hull and the surface): tiles of the valid list without a single live point skip the colour head
                for db in (-20.0, -30.0):
                    w2 = ops.PackedWeights(random_hotpath_state_dict(seed=3, density_bias=db), device=dev)
                    cs = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
                    nst = max(2, min(args.steps, 5))
                    L.check(L.kpn_density_stats(cs, ctypes.byref(dens[0]), ctypes.byref(dens[1]), 1))
                    ms2, rows2 = time_frames(L, ops, torch, scene, w2, res, args.samples, fine, steps=nst, warmup=4)   # auto (the default)
                    L.check(L.kpn_density_stats(cs, ctypes.byref(dens[0]), ctypes.byref(dens[1]), 1))
                    L.check(L.kpn_density_first_passes(ctypes.byref(dpasses[0]), ctypes.byref(dpasses[1]), 1))
                    L.check(L.kpn_set_density_first(1))
                    ms5, _ = time_frames(L, ops, torch, scene, w2, res, args.samples, fine, steps=nst)
                    L.check(L.kpn_set_density_first(0))    # the fused per-point kernel: short path per 32-point tile only
                    ms4, _ = time_frames(L, ops, torch, scene, w2, res, args.samples, fine, steps=nst)
                    os.environ["KPN_NO_ZERO_SKIP"] = "1"
                    ms3, _ = time_frames(L, ops, torch, scene, w2, res, args.samples, fine, steps=nst)
                    os.environ.pop("KPN_NO_ZERO_SKIP")
                    L.check(L.kpn_set_density_first(2))
                    sec[f"partly_empty_hull_density_bias_{int(db)}"] = {
                        "ms_per_frame": ms2, "sigma_zero_fraction": (1.0 - dens[1].value / dens[0].value) if dens[0].value > 0 else None,
                        "passes_density_first_of_the_timed_frames_in_auto_mode": [dpasses[0].value, dpasses[0].value + dpasses[1].value],
                        "ms_per_frame_density_first_always": ms5,
                        "ms_per_frame_fused_per_point_kernel_tile_short_path": ms4,
                        "ms_per_frame_without_the_zero_density_short_path": ms3,
                        "rays_per_sec": rays_per_step / (ms2 * 1e-3)}
            # the same frame with the other rows kernels: three bf16 pieces (mode 2, roof 2500 / 6) and fp32 MFMA (mode 0, roof 157.3)
            for m, key in ((2, "bf16x3_rows_kernel_mode2"), (0, "fp32_mfma_rows_kernel_mode0"), (3, "fp16x2_rows_kernel_mode3")):
                if m == args.geo_rows_mode:
                    continue
                L.check(L.kpn_set_geo_rows_mode(m))
                ms2, rows2, kms, ktf = time_frames(L, ops, torch, scene, w, res, args.samples, fine, steps=max(2, min(args.steps, 5)), with_kernel=True)
                L.check(L.kpn_set_geo_rows_mode(args.geo_rows_mode))
                sec[key] = {"ms_per_frame": ms2, "rays_per_sec": rays_per_step / (ms2 * 1e-3),
                            "roofline": {"kernel": ROWS_KERNEL[m], "bound": "mfma", "achieved": ktf, "peak": rows_peak_tflops(m),
                                         "unit": "TFLOP/s", "frac": ktf / rows_peak_tflops(m), "avg_launch_ms": kms}}
            if args.views == 3:
                sec["training_step_configs3"] = time_training(ops, torch, dev, sd)
                sec["training_step_4096"] = time_training(ops, torch, dev, sd, steps=5, patch=64)   # the reference's shipped 64 x 64 patch
            if not args.no_configs4 and res == 512 and fine:       # next to the headline only
                del plan, ps, scene
                torch.cuda.empty_cache()
                try:
                    sec["configs4_full"] = time_configs4(L, ops, torch, dev, sd, args.geo_rows_mode)
                except Exception as e:  # noqa: BLE001  (e.g. a box with less host memory): say so, keep the line
                    sec["configs4_full"] = {"error": f"{type(e).__name__}: {e}"}
            line["secondary"] = sec
            # the dominant kernel's fraction of its roof on the three workloads the headline is quoted beside (same kernel, same
            # accounting: algorithmic FLOP of the rows it processed / its launch time inside the library)
            side = {"configs[1] ellipsoid mask (headline)": {"frac": line["roofline"]["frac"], "achieved": achieved, "avg_launch_ms": line["roofline"]["avg_launch_ms"],
                                                              "ms_per_frame": line["ms_per_step"], "rays_per_sec": value}}
            dm = sec.get("dense_mask")
            if dm:
                side["configs[1] dense mask (80 % of the field evaluations valid)"] = dict(dm["roofline"], ms_per_frame=dm["ms_per_frame"], rays_per_sec=dm["rays_per_sec"])
            c4 = sec.get("configs4_full")
            if c4 and "roofline" in c4:
                side["configs[4] at full size (4096^2 rays, V = 10, 128 flat samples)"] = dict(c4["roofline"], ms_per_frame=c4["ms_per_step"], rays_per_sec=c4["rays_per_sec"],
                                                                                            parity_ok=(c4.get("parity") or {}).get("ok"))
            line["roofline"]["workloads"] = {k: {kk: vv for kk, vv in v.items() if kk in ("frac", "achieved", "peak", "avg_launch_ms", "ms_per_frame", "rays_per_sec", "parity_ok")}
                                             for k, v in side.items()}
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N=1 only
            def make_render_one():   # the product's one-pixel render with its per-sample stages, for the rays the gate widens
                from tests import parity_gate
                sc1 = to_device(scene_cpu, dev)
                ps1 = ops.PreparedScene(sc1["img"], sc1["cam"], sc1["feat_geo"], sc1["feat_tex"], sc1["sp_data"], sc1["src_foreground_mask"])
                return parity_gate.product_render_one(ops, ps1, w, sc1["cam_tar"], sc1["bounds"], args.samples, args.samples, fine)
            port, parity = cpu_baseline(args, scene_cpu, sd, fine=fine, frame=frame_np, render_one=make_render_one)
            # the timed frame itself, compared with the oracle on the rays the CPU baseline renders anyway
            line["parity"] = parity
            # `cpu_baseline` = the C port of the reference's arithmetic (the oracle), timed LIVE on this box's host cores on a bounded
            # sample of the timed frame.  The UNMODIFIED reference (PyTorch CPU, src/model.py) cannot travel to the GPU box; its rate on a
            # configs[1] tile, timed where /root/reference is mounted (the 8-core build container: scripts/time_reference_cpu.py), rides
            # along as a recorded figure, labelled as such — until round 4 it was `value`, which is not "this box's host cores".
            line["cpu_baseline"] = dict(port)
            rj = os.path.join(ROOT, "profiles", "reference_cpu_pytorch.json")
            if os.path.exists(rj):
                ref = json.load(open(rj))
                line["cpu_baseline"]["reference_pytorch_recorded"] = {
                    "value": ref.get("rays_per_sec", ref.get("value")), "unit": "rays/s", "cores": ref.get("cores"), "kind": "reference",
                    "sample": ref.get("sample", ref.get("what")),
                    "recorded": "profiles/reference_cpu_pytorch.json (build container, 8 cores; the reference is absent on the GPU box)", "detail": ref}
            ej = os.path.join(ROOT, "profiles", "r04_eager_pytorch_on_mi355x.json")
            if os.path.exists(ej):   # eager PyTorch restatement of the path on an MI355X (scripts/bench_torch_eager.py), recorded
                e = json.load(open(ej))
                line["eager_pytorch_same_gpu"] = {"rays_per_sec": max(v["rays_per_sec"] for k, v in e.items() if "ray" in k and isinstance(v, dict)),
                                                  "recorded_by": "scripts/bench_torch_eager.py (profiles/r04_eager_pytorch_on_mi355x.json)",
                                                  "what": e["what"]}
        print(json.dumps(line), flush=True)
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
