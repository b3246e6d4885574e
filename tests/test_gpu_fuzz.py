"""Randomised parity sweep as a gate (-m gpu): HIP render (through the C ABI) against the C oracle on seeded random scenes —
view counts, resolutions, sample counts, masks, target cameras, density biases, ragged chunks.  Every ray must be within 1e-4
(RGB and alpha, coarse and fine) unless the oracle's own conditioning probe explains it (tests/parity_gate.py) AND, since round 6, the conditional re-check passes for it: rays that a
hard validity threshold of a resampled point or the 1e10 last interval makes ill-conditioned in the reference itself.
No tolerated-outlier count: an unexplained ray fails the test.  (scripts/fuzz_parity.py runs the same sweep at any size.)"""
import numpy as np
import pytest
import torch

from tests import parity_gate

pytestmark = pytest.mark.gpu


def fuzz_scene(rng):
    V = int(rng.choice([1, 2, 3, 3, 3, 4, 6, 10]))
    sh, sw = int(rng.choice([48, 64, 96, 128])), int(rng.choice([48, 64, 96, 128]))
    th, tw = int(rng.integers(8, 40)), int(rng.integers(8, 40))
    Sc, Sf = int(rng.choice([8, 16, 32, 64, 96])), int(rng.choice([4, 16, 32, 64]))
    mask = str(rng.choice(["ellipsoid", "dense"]))
    focal = float(rng.choice([600.0, 800.0, 1000.0]))
    angle = float(rng.uniform(0, 360)) if rng.random() < 0.5 else None
    fine = bool(rng.random() < 0.8)
    seed = int(rng.integers(1, 10 ** 6))
    bias = float(rng.choice([0.0, 0.0, -15.0, -25.0]))
    chunk = int(rng.choice([0, 0, 100, 333]))
    return dict(V=V, src=(sh, sw), tar=(th, tw), Sc=Sc, Sf=Sf, mask=mask, focal=focal, angle=angle, fine=fine, seed=seed, bias=bias, chunk=chunk)


def run_scene(ops, cfg):
    from keypointnerf_amd.synthetic import make_scene, random_hotpath_state_dict, to_device
    from oracle import oracle
    sd = random_hotpath_state_dict(seed=cfg["seed"], density_bias=cfg["bias"])
    scene = make_scene(n_views=cfg["V"], src_hw=cfg["src"], tar_hw=cfg["tar"], mask=cfg["mask"], seed=cfg["seed"] + 1,
                       tar_angle=cfg["angle"], tar_focal_at_512=cfg["focal"])
    s = to_device(scene, "cuda")
    ps = ops.PreparedScene(s["img"], s["cam"], s["feat_geo"], s["feat_tex"], s["sp_data"], s["src_foreground_mask"])
    th, tw = cfg["tar"]
    out = ops.render_rays(ps, ops.PackedWeights(sd), s["cam_tar"], s["bounds"], grid=(0, 0, 1, tw, th), n_coarse=cfg["Sc"], n_fine=cfg["Sf"],
                          fine=cfg["fine"], chunk_rays=cfg["chunk"])
    yy, xx = np.meshgrid(np.arange(th), np.arange(tw), indexing="ij")
    pix = np.stack([xx.reshape(-1), yy.reshape(-1)], -1).astype(np.int32)
    osc, wf = oracle.OracleScene(scene), oracle.flat_weights(sd)
    ref = oracle.render_rays(osc, wf, scene["cam_tar"], scene["bounds"], pix, cfg["Sc"], cfg["Sf"], fine=cfg["fine"])
    keys = ("tex_fg", "alpha") + (("tex_fg_fine", "alpha_fine") if cfg["fine"] else ())
    got = {k: (out[k][0].permute(1, 2, 0).reshape(-1, 3) if k.startswith("tex") else out[k].reshape(-1)).cpu().numpy() for k in keys}
    rep = parity_gate.check_rays(
        got, ref, parity_gate.oracle_envelope(oracle, osc, wf, scene["cam_tar"], scene["bounds"], pix, cfg["Sc"], cfg["Sf"], fine=cfg["fine"]),
        keys=keys, max_widened_fraction=0.01, what=str(cfg))
    # round 6: a widened ray is accepted only if every stage of the oracle, run on the kernels' own inputs of that stage, agrees with
    # the kernels at the strict stage bars (parity_gate.conditional_check); raises otherwise
    w = ops.PackedWeights(sd)
    return parity_gate.recheck_widened(rep, parity_gate.product_render_one(ops, ps, w, s["cam_tar"], s["bounds"], cfg["Sc"], cfg["Sf"], cfg["fine"]),
                                       oracle, osc, wf, scene["cam_tar"], scene["bounds"], pix, cfg["Sc"], cfg["Sf"], fine=cfg["fine"])


# A PINNED regression set (an advisor finding of round 5: "so gate drift shows up"): six scenes of the 200-scene sweep (seed 2024,
# profiles/r06_z_fuzz_parity_200_scenes_default.json) that hold rays the reference itself is ill-conditioned on — scene index ->
# rays above the bar with the shipped kernels.  Every one of them must be widened by the FROZEN gate and pass the conditional re-check
# (run_scene raises otherwise); a kernel or gate change that moves the counts shows here.
PINNED_SEED, PINNED = 2024, {7: 1, 13: 1, 26: 1, 67: 2, 68: 3, 167: 4}


def test_pinned_ill_conditioned_scenes():
    from keypointnerf_amd import ops
    assert ops.get_geo_rows_mode() == 3 and ops.get_fuse_mode() == 1
    rng = np.random.default_rng(PINNED_SEED)
    got = {}
    for i in range(max(PINNED) + 1):
        cfg = fuzz_scene(rng)                      # (the sequence of the sweep: every scene draws from the stream)
        if i in PINNED:
            rep = run_scene(ops, cfg)
            assert rep["above_bar"] == len(rep["widened"]) and rep.get("conditional_ok", True), (i, rep["above_bar"], len(rep["widened"]))
            got[i] = rep["above_bar"]
    assert sum(abs(got[i] - PINNED[i]) for i in PINNED) <= 2, (got, PINNED)


@pytest.mark.parametrize("rows_mode", [3, 0])
def test_random_scenes_against_the_oracle(rows_mode):
    from keypointnerf_amd import ops
    rng = np.random.default_rng(7)
    default_mode = ops.get_geo_rows_mode()
    ops.set_geo_rows_mode(rows_mode)
    try:
        rays = widened = above = 0
        for _ in range(24):
            rep = run_scene(ops, fuzz_scene(rng))
            rays += rep["rays"]
            widened += len(rep["widened"])
            above += rep["above_bar"]
        # the ill-conditioned rays are rare: a handful per 1e4 (profiles/*fuzz*)
        assert widened <= max(3, rays // 1000), (widened, rays)
    finally:
        ops.set_geo_rows_mode(default_mode)
