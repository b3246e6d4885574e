"""The parity gate itself under test (round 6: the gate is frozen — tests/parity_gate.py — and these are the tests that must fail
before one of its constants may move).

Positive: the oracle's own stages, and the kernel sources on the emulator, pass the CONDITIONAL check (every stage of the oracle on
the other side's inputs of that stage) on every ray of a small scene.  Negative: check_rays and conditional_check REJECT
  * the reference + 3e-4 on the 0.1 % of rays with the largest envelopes (an error 3x the bar on the rays the widened bar is most
    generous to),
  * a renderer that blends the wrong source image into a view (two views' images swapped),
  * a renderer whose every resampled depth sits one bin further along the ray.
All on the CPU (C oracle + the wave64 emulator)."""
import numpy as np
import pytest

from oracle import oracle
from tests import parity_gate
from tests import simt_harness as sh
from keypointnerf_amd.synthetic import make_scene, random_hotpath_state_dict

SC, SF, N = 16, 16, 32          # 1024 rays


@pytest.fixture(scope="module")
def world():
    scene = make_scene(n_views=3, src_hw=(64, 64), tar_hw=(N, N), mask="ellipsoid", seed=5, tar_focal_at_512=800.0)
    sd = random_hotpath_state_dict(seed=3)
    osc, wflat = oracle.OracleScene(scene), oracle.flat_weights(sd)
    ys, xs = np.meshgrid(np.arange(N), np.arange(N), indexing="ij")
    pix = np.stack([xs.reshape(-1), ys.reshape(-1)], -1).astype(np.int32)
    ref = oracle.render_rays(osc, wflat, scene["cam_tar"], scene["bounds"], pix, SC, SF, fine=True, stages=True)
    env_fn = parity_gate.oracle_envelope(oracle, osc, wflat, scene["cam_tar"], scene["bounds"], pix, SC, SF, fine=True, trials=8)
    return dict(scene=scene, sd=sd, osc=osc, wflat=wflat, pix=pix, ref=ref, env_fn=env_fn)


def _stages(o):
    return {"z_coarse": o["z_c"], "rgba_coarse": o["rgba_c"], "z_fine": o["z_f"], "rgba_fine": o["rgba_f"]}


def _cond(w, out, stages, pix=None):
    pix = w["pix"] if pix is None else pix
    return parity_gate.conditional_check(oracle, w["osc"], w["wflat"], w["scene"]["cam_tar"], w["scene"]["bounds"], pix, out, stages, SC, SF)


def test_the_oracle_passes_its_own_conditional_check(world):
    res = _cond(world, world["ref"], _stages(world["ref"]))
    assert all(r["ok"] for r in res), [r for r in res if not r["ok"]][:2]
    assert max(max(r["stages"].values()) for r in res) <= 0.0
    assert float(world["ref"]["alpha_fine"].max()) > 0.2       # the scene is not empty


def test_the_kernels_on_the_emulator_pass_the_conditional_check_on_every_ray(world):
    """the product's kernels (emulated wave64), through the C ABI with kpn_render_stages: end to end within the bar AND every stage
    within its strict bar given the kernels' own inputs"""
    lib = sh.simt_lib()
    n = 12
    hs = sh.HostScene(lib, world["scene"])
    packed = sh.pack_weights(lib, world["sd"])
    out, st = sh.render(lib, hs, packed, world["scene"]["cam_tar"], world["scene"]["bounds"], (10, 10, 1, n, n), SC, SF, stages=True)
    ys, xs = np.meshgrid(np.arange(n) + 10, np.arange(n) + 10, indexing="ij")
    pix = np.stack([xs.reshape(-1), ys.reshape(-1)], -1).astype(np.int32)
    got = {k: (out[k].reshape(3, -1).T if out[k].ndim == 3 else out[k].reshape(-1)) for k in ("tex_fg", "alpha", "tex_fg_fine", "alpha_fine")}
    sel = pix[:, 1] * N + pix[:, 0]
    for k in got:
        assert np.abs(got[k] - world["ref"][k][sel]).max() <= parity_gate.RGBA_TOL, k
    res = _cond(world, got, st, pix)
    assert all(r["ok"] for r in res), [r for r in res if not r["ok"]][:2]


def test_gate_rejects_three_times_the_bar_on_the_most_ill_conditioned_rays(world):
    """ref + 3e-4 on 0.1 % of the rays, chosen where the envelope is LARGEST (where the widened bar is most generous): rejected,
    unless the oracle itself moves that much there — and then the conditional check still rejects it (its compositor stage sees
    an output 3e-4 away from what the field values composite to)."""
    w = world
    env = w["env_fn"](np.arange(N * N))
    worst = np.argsort(-np.maximum(env["alpha_fine"], env["tex_fg_fine"]))[:max(1, N * N // 1000)]
    out = {k: w["ref"][k].copy() for k in ("tex_fg", "alpha", "tex_fg_fine", "alpha_fine")}
    out["alpha_fine"][worst] += 3e-4
    out["tex_fg_fine"][worst] += 3e-4
    rejected = False
    try:
        rep = parity_gate.check_rays(out, w["ref"], w["env_fn"], what="negative: +3e-4")
    except AssertionError:
        rejected = True
    if not rejected:   # the envelope licensed it: "explained" must then mean re-checked
        assert len(rep["widened"]) == len(worst)
        res = _cond(w, {k: v[worst] for k, v in out.items()}, {k: v[worst] for k, v in _stages(w["ref"]).items()}, w["pix"][worst])
        assert not any(r["ok"] for r in res)
        assert all(r["stages"]["composite_fine"] > 0 for r in res)
    # the same error on well-conditioned rays never reaches the conditional check
    best = np.argsort(np.maximum(env["alpha_fine"], env["tex_fg_fine"]))[:2]
    out = {k: w["ref"][k].copy() for k in ("tex_fg", "alpha", "tex_fg_fine", "alpha_fine")}
    out["alpha_fine"][best] += 3e-4
    with pytest.raises(AssertionError):
        parity_gate.check_rays(out, w["ref"], w["env_fn"], what="negative: +3e-4, well-conditioned rays")


def test_gate_rejects_a_swapped_view_pair_in_the_blend(world):
    """a renderer that samples view 1's image where view 0's belongs (and vice versa)"""
    w = world
    wrong_scene = dict(w["scene"])
    img = w["scene"]["img"].clone()
    img[[0, 1]] = img[[1, 0]]
    wrong_scene["img"] = img
    o = oracle.render_rays(oracle.OracleScene(wrong_scene), w["wflat"], w["scene"]["cam_tar"], w["scene"]["bounds"], w["pix"], SC, SF, stages=True)
    assert np.abs(o["tex_fg_fine"] - w["ref"]["tex_fg_fine"]).max() > 1e-3
    with pytest.raises(AssertionError):
        parity_gate.check_rays(o, w["ref"], w["env_fn"], what="negative: swapped views")
    hit = np.nonzero(np.abs(o["tex_fg_fine"] - w["ref"]["tex_fg_fine"]).max(-1) > 1e-3)[0][:8]
    res = _cond(w, {k: o[k][hit] for k in ("tex_fg", "alpha", "tex_fg_fine", "alpha_fine")}, {k: v[hit] for k, v in _stages(o).items()}, w["pix"][hit])
    assert not any(r["ok"] for r in res) and all(r["stages"]["field_fine"] > 0 for r in res)


def test_gate_rejects_a_one_bin_shift_of_every_resample(world):
    """every importance sample one coarse bin further along the ray; the field and the compositor are the oracle's own on those
    depths, so ONLY the resampling stage is wrong"""
    w = world
    sel = np.nonzero(w["ref"]["alpha_fine"] > 0.05)[0][:64]
    pix = w["pix"][sel]
    dirs, cam_pos, near, far = oracle.make_rays(w["scene"]["cam_tar"], w["scene"]["bounds"], pix)
    zc, rc = w["ref"]["z_c"][sel], w["ref"]["rgba_c"][sel]
    _, _, _, contrib, _ = oracle.rgba2out(rc, zc)
    zmid = 0.5 * (zc[:, 1:] + zc[:, :-1])
    znew = oracle.importance_sample(np.ascontiguousarray(contrib[:, 1:SC - 1]), zmid, SF)
    znew = znew + (zc[:, 1:2] - zc[:, 0:1])                                   # one bin further
    zf = np.sort(np.concatenate([zc, znew], -1), -1).astype(np.float32)
    pts = (cam_pos[None, None] + dirs[:, None] * zf[..., None]).astype(np.float32)
    rf = oracle.query(w["osc"], w["wflat"], pts.reshape(-1, 3), np.repeat(dirs[:, None], SC + SF, 1).reshape(-1, 3), apply_eval_func=True)[0].reshape(len(sel), SC + SF, 5)
    color, _, alpha, _, _ = oracle.rgba2out(rf, zf)
    out = {"tex_fg": w["ref"]["tex_fg"][sel], "alpha": w["ref"]["alpha"][sel], "tex_fg_fine": color, "alpha_fine": alpha}
    ref = {k: w["ref"][k][sel] for k in out}
    env_fn = parity_gate.oracle_envelope(oracle, w["osc"], w["wflat"], w["scene"]["cam_tar"], w["scene"]["bounds"], pix, SC, SF, fine=True, trials=8)
    assert max(np.abs(out["alpha_fine"] - ref["alpha_fine"]).max(), np.abs(out["tex_fg_fine"] - ref["tex_fg_fine"]).max()) > 2e-4
    with pytest.raises(AssertionError):
        parity_gate.check_rays(out, ref, env_fn, what="negative: one-bin shift", max_widened_fraction=1.0)
    res = _cond(w, out, {"z_coarse": zc, "rgba_coarse": rc, "z_fine": zf, "rgba_fine": rf}, pix)
    assert not any(r["ok"] for r in res)
    assert all(r["stages"]["resample"] > 0 and r["stages"]["field_fine"] <= 0 and r["stages"]["composite_fine"] <= 0 for r in res)


def test_gate_constants_are_frozen():
    """round 6: no constant of the gate moves without a failing negative test beside it"""
    import inspect
    sig = inspect.signature(parity_gate.check_rays)
    assert (parity_gate.RGBA_TOL, parity_gate.ENVELOPE_TRIALS) == (1e-4, 64)
    assert {k: sig.parameters[k].default for k in ("tol", "cap", "cap_envelopes", "max_widened_fraction")} == \
        {"tol": 1e-4, "cap": 0.1, "cap_envelopes": 20.0, "max_widened_fraction": 2e-3}
    assert (parity_gate.FIELD_TOL, parity_gate.COMPOSITE_TOL, parity_gate.DEPTH_TOL, parity_gate.HEAD_TERMS, parity_gate.RAY_TOL) == (2e-5, 5e-6, 5e-6, 4.0, 1e-6)
