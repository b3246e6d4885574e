"""Trained-weights fixture (round 6): the UNMODIFIED reference (imported from /root/reference through oracle/ref_shim.py), its
hot-path modules trained by its own batch_render_pifu_nerf train branch + compute_error (src/model.py:845-895, src/utils.py:97-171:
L1 coarse + 10 x L1 fine as configs/zju.json:109-112 weighs them; the VGG term is stubbed like everywhere here — no pretrained
weights offline) with Adam at the shipped learning rate (configs/zju.json:18: 5e-4) on a small synthetic multi-view data set that
MEANS something: a textured ellipsoid seen by three source cameras and a target camera per scene, the feature maps produced by the
reference's own image encoders (at their init, frozen) from those images, the target image as ground truth.  Then the
configs[1]-tile fixture of oracle/make_golden.py::run_headline_case with THOSE weights:

    tests/golden/weights_trained_seed0.npz        the hot-path state dict after training
    tests/golden/case_t_v3_trained_tile.npz       one level-4 tile of a 512 x 512 target, V = 3, 64 + 64 samples, encoder maps

Why: every other fixture uses the reference's seeded init (biases / weight_g perturbed); a trained density is sharp (opaque
surface, empty space: relu(rad) == 0 in most of the hull) and trained weight_g / biases have drifted — the operand statistics the
two-fp16-piece kernels must carry.  Run in the build container only:   python -m oracle.make_trained_golden [steps]
TEST INFRASTRUCTURE ONLY."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from keypointnerf_amd.synthetic import make_scene, perturb_reference_net  # noqa: E402
from oracle import make_golden, ref_shim  # noqa: E402

SEMI = np.array([0.42, 0.95, 0.36])     # the ellipsoid of make_scene's "ellipsoid" masks


def surface_colour(p):
    """smooth texture on the ellipsoid's surface, as a function of the world point (the same point seen from any camera)"""
    x, y, z = p[..., 0], p[..., 1], p[..., 2]
    return np.stack([0.5 + 0.4 * np.sin(7.0 * x + 3.0 * y), 0.5 + 0.4 * np.cos(5.0 * y - 4.0 * z), 0.5 + 0.4 * np.sin(6.0 * z + 2.0 * x * y)], -1)


def ellipsoid_image(K, E, H, W):
    """(3,H,W) colour (black background) and (H,W) hit mask of the textured ellipsoid seen by the camera K (4,4), E = [R|t] (4,4)"""
    K, E = np.asarray(K, np.float64), np.asarray(E, np.float64)
    R, t = E[:3, :3], E[:3, 3]
    o = -R.T @ t
    ys, xs = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    d = np.stack([xs, ys, np.ones_like(xs)], -1) @ np.linalg.inv(K[:3, :3]).T @ R
    os_, ds_ = o / SEMI, d / SEMI
    a, b, c = (ds_ * ds_).sum(-1), 2.0 * (ds_ * os_).sum(-1), (os_ * os_).sum() - 1.0
    disc = b * b - 4 * a * c
    hit = disc >= 0.0
    tt = (-b - np.sqrt(np.where(hit, disc, 0.0))) / (2 * a)
    p = o[None, None] + d * tt[..., None]
    n = p / SEMI ** 2
    n /= np.linalg.norm(n, axis=-1, keepdims=True) + 1e-12
    shade = 0.65 + 0.35 * np.clip(-(n * (d / np.linalg.norm(d, axis=-1, keepdims=True))).sum(-1), 0, 1)
    img = surface_colour(p) * shade[..., None] * hit[..., None]
    return torch.tensor(img.transpose(2, 0, 1), dtype=torch.float32), torch.tensor(hit)


def training_scene(net, seed, src_hw=(128, 128), tar_hw=(64, 64), tar_angle=None, tar_focal=800.0):
    """make_scene's cameras / keypoints / masks; source images, target image = the textured ellipsoid; feature maps = the reference's
    own encoders on the source images"""
    scene = make_scene(n_views=3, src_hw=src_hw, tar_hw=tar_hw, mask="ellipsoid", seed=seed, tar_angle=tar_angle, tar_focal_at_512=tar_focal)
    H, W = src_hw
    imgs = [ellipsoid_image(scene["cam"]["K"][v].numpy(), scene["cam"]["extrin"][v].numpy(), H, W)[0] for v in range(3)]
    scene["img"] = torch.stack(imgs, 0)
    with torch.no_grad():
        fg = type(net).attach_geo_feat(net, scene["img"], True)
        ft = type(net).attach_tex_feat(net, scene["img"], True)
    scene["feat_geo"], scene["feat_tex"] = [x.detach().clone() for x in fg], ft.detach().clone()
    tar, hit = ellipsoid_image(scene["cam_tar"]["K"][0].numpy(), scene["cam_tar"]["RT"][0].numpy(), *tar_hw)
    scene["tar_img"], scene["tar_msk"] = tar[None], hit[None, None]
    return scene


def train(net, steps, patch=32, Sc=16, Sf=16, log_every=25):
    from src import utils as rutils                      # the reference's own compute_error
    cfgj = ref_shim.load_config()
    lambdas = dict(find_key(cfgj, "lambdas"))
    lambdas["lambda_vgg"] = 0.0
    lr = float(find_key(cfgj, "lr"))
    scenes = [training_scene(net, seed=60 + i, tar_angle=[25.0, 80.0, 150.0, 215.0, 290.0, 335.0][i]) for i in range(6)]
    hot = [p for n, p in net.named_parameters() if n.startswith(make_golden.HOT_PREFIXES)]
    opt = torch.optim.Adam(hot, lr=lr)
    net.train()
    net.train_out_h = net.train_out_w = patch
    np.random.seed(0)
    torch.manual_seed(0)
    hist, t0 = [], time.time()
    for it in range(steps):
        s = scenes[it % len(scenes)]
        cfg = dict(fine=True, uniform=False, sample_per_ray_c=Sc, sample_per_ray_f=Sf, rand_noise_std=0.01,
                   src_foreground_mask=s["src_foreground_mask"], bounds=s["bounds"], msk=s["tar_msk"])
        out = net.batch_render_pifu_nerf(net, s["img"], s["cam"], 3, s["cam_tar"], 5, 0, s["tar_img"], s["feat_geo"], s["feat_tex"],
                                         dict(s["sp_data"]), None, **cfg)
        out["tex_cal"], out["tex_cal_fine"] = out["tex_fg"], out["tex_fg_fine"]          # src/model.py:887-891
        loss, err = rutils.compute_error(out_nerf=out, vggloss=None, lambdas=lambdas)
        opt.zero_grad()
        loss.backward()
        opt.step()
        hist.append(float(loss))
        if it % log_every == 0 or it == steps - 1:
            print(f"step {it:4d}: loss {float(loss):.4f}  (mean of last {min(len(hist), log_every)}: {np.mean(hist[-log_every:]):.4f})  "
                  f"alpha_fine {float(out['alpha_fine'].mean()):.3f}  {time.time() - t0:.0f} s", flush=True)
    net.eval()
    return hist, scenes


def find_key(d, key):
    if isinstance(d, dict):
        if key in d:
            return d[key]
        for v in d.values():
            r = find_key(v, key)
            if r is not None:
                return r
    return None


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 320
    net = ref_shim.build_reference_net(seed=0)
    perturb_reference_net(net, seed=7)                  # the starting point of every other fixture
    w0 = {k: v.detach().clone() for k, v in net.state_dict().items() if k.startswith(make_golden.HOT_PREFIXES)}
    hist, scenes = train(net, steps)
    sd = {k: make_golden._np(v) for k, v in net.state_dict().items() if k.startswith(make_golden.HOT_PREFIXES)}
    np.savez_compressed(os.path.join(make_golden.GOLDEN_DIR, "weights_trained_seed0.npz"), **sd)
    drift = {k: float((torch.from_numpy(sd[k]) - w0[k]).abs().max()) for k in sd}
    print("largest parameter drift:", sorted(drift.items(), key=lambda kv: -kv[1])[:5])
    # the fixture: a configs[1] tile (level 4 of a 512 x 512 target, 64 + 64 samples) of a scene the training did not see, its maps
    # from the encoders; run_headline_case renders it with the reference and records outputs + stage samples
    held_out = training_scene(net, seed=77, src_hw=(128, 128), tar_hw=(512, 512), tar_angle=118.0)
    orig = make_golden.make_scene
    make_golden.make_scene = lambda **kw: {k: v for k, v in held_out.items() if k not in ("tar_img", "tar_msk")}
    try:
        n_eval, dt = make_golden.run_headline_case(net, "case_t_v3_trained_tile", 3, (128, 128), "ellipsoid", 4, (3, 5), 64, 64, seed=77)
    finally:
        make_golden.make_scene = orig
    json.dump({"steps": steps, "lr": 5e-4, "loss_first_25": float(np.mean(hist[:25])), "loss_last_25": float(np.mean(hist[-25:])),
               "largest_drift": dict(sorted(drift.items(), key=lambda kv: -kv[1])[:8]),
               "what": "oracle/make_trained_golden.py: the unmodified reference's train branch + compute_error (L1 terms), Adam lr 5e-4, "
                       "32 x 32 patches, 16 + 16 samples, six textured-ellipsoid scenes, encoders frozen at the reference's init"},
              open(os.path.join(os.path.dirname(make_golden.GOLDEN_DIR), "..", "profiles", "r06_trained_fixture_training_log.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
