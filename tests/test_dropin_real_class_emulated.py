"""The UNMODIFIED reference class, live, with and without install() — in the build container, where /root/reference is.

The build container has the reference but no GPU; the GPU box has a GPU but no reference.  To run the real `KeypointNeRF`
through the drop-in anyway, these tests execute the product's kernel SOURCES on the wave64 host emulator (tests/simt, the same
C ABI) behind the unchanged `keypointnerf_amd.ops` / `dropin` / `torch_ops` code: the test swaps the library handle, the
"is this tensor on the GPU" predicate and the stream getter, and gives the `torch.ops.kpnerf.*` operators a CPU kernel for the
duration of the test (a scoped library fragment).  Nothing in the product knows about this: without these patches every call
below raises, which `tests/test_dropin_cpu.py` asserts.

What this pins that the StandInNet tests (GPU) and the recorded goldens cannot: the live call sites, the live random-number
stream (the reference's draws and the drop-in's must consume `torch` / `numpy` generators in the same order with the same
shapes: same seed => same dropout, jitter, noise, patch centre), the weight-norm fold and packing from the live parameters,
the encoders' hand-off, and gradients flowing back into the encoders' parameters through autograd."""
import numpy as np
import pytest
import torch

from oracle import ref_shim

pytestmark = pytest.mark.skipif(not ref_shim.reference_available(), reason="needs /root/reference (build container)")

_CPU_KERNELS = ("rgba2out", "rgba2out_backward", "importance_sample", "ray_bbox_intersection", "field_query", "render_rays",
                "render_rays_train", "render_rays_train_backward", "pix_l1_loss")


@pytest.fixture
def emulated(monkeypatch):
    from keypointnerf_amd import lib as kl, ops, torch_ops
    from tests.simt_harness import simt_lib
    L = simt_lib()
    monkeypatch.setattr(kl, "get_library", lambda: L)
    monkeypatch.setattr(ops, "_on_gpu", lambda t: True)
    monkeypatch.setattr(ops, "_stream", lambda: None)
    with torch.library._scoped_library("kpnerf", "FRAGMENT") as frag:
        for name in _CPU_KERNELS:
            frag.impl(name, getattr(torch_ops, name)._init_fn, "CPU")
        yield L


def _net_and_scene(tar=16, src=128):
    from keypointnerf_amd.synthetic import make_scene, perturb_reference_net
    net = ref_shim.build_reference_net(seed=0)
    perturb_reference_net(net, seed=7)
    s = make_scene(n_views=3, src_hw=(src, src), tar_hw=(tar, tar), mask="ellipsoid", seed=5, tar_focal_at_512=800.0)
    return net, s


def _close(got, ref, name, tol=1e-4, flips=0):
    d = (got - ref).abs().reshape(-1) if got.dim() < 3 else (got - ref).abs().reshape(-1, got.shape[-2] * got.shape[-1]).max(0)[0]
    assert int((d > tol).sum()) <= flips, (name, float(d.max()))


@torch.no_grad()
def test_render_pifu_nerf_of_the_live_class(emulated):
    """render_full_nerf_image's call (src/model.py:454-472), encoders included, reference vs installed."""
    from keypointnerf_amd.dropin import install, uninstall
    net, s = _net_and_scene()
    kw = dict(net=net, img_in=s["img"], cam_in=s["cam"], cam_tar=s["cam_tar"], tar_img=None, sp_data=dict(s["sp_data"]),
              objcenter=torch.zeros(1, 3), fine=True, uniform=True, objrad=250., blur=3, level=1, sample_per_ray_c=12,
              sample_per_ray_f=8, src_foreground_mask=s["src_foreground_mask"], bounds=s["bounds"], mask_at_box=torch.ones(1, 16, 16))
    ref = net.render_pifu_nerf(**kw)
    install(net)
    got = net.render_pifu_nerf(**kw)
    again = net.render_pifu_nerf(**kw)                               # second camera of an orbit: encoders not re-run, same frame
    uninstall(net)
    assert float(ref["alpha_fine"].max()) > 0.5 and float(ref["alpha_fine"].min()) < 0.05    # a subject and a background
    for k in ("tex_fg", "alpha", "depth", "tex_fg_fine", "alpha_fine", "depth_fine"):
        assert got[k].shape == ref[k].shape and got[k].device.type == "cpu", k
        _close(got[k], ref[k], k)
        assert torch.equal(got[k], again[k]), k


def test_encoders_run_once_per_source_set_and_again_when_their_state_changes(emulated):
    """render_novel_views attaches the source set once (attach_im_feat, src/model.py:479) and render_pifu_nerf re-runs both
    encoders per camera (:913-914).  With the drop-in the attached maps are taken as long as the images AND the encoders' state
    (parameters and buffers, by version counter) are the ones they were computed with — also under torch.inference_mode, where
    the images carry no version counter (Lightning validate / test) — and the encoders run again the moment a weight changes."""
    from keypointnerf_amd.dropin import install, uninstall
    net, s = _net_and_scene()
    calls = {"geo": 0, "tex": 0}
    net.geo_encoder.register_forward_hook(lambda *a: calls.__setitem__("geo", calls["geo"] + 1))
    net.tex_encoder.register_forward_hook(lambda *a: calls.__setitem__("tex", calls["tex"] + 1))
    kw = dict(net=net, img_in=s["img"], cam_in=s["cam"], cam_tar=s["cam_tar"], tar_img=None, sp_data=dict(s["sp_data"]),
              objcenter=torch.zeros(1, 3), fine=True, uniform=True, objrad=250., blur=3, level=1, sample_per_ray_c=8,
              sample_per_ray_f=8, src_foreground_mask=s["src_foreground_mask"], bounds=s["bounds"], mask_at_box=torch.ones(1, 16, 16))
    install(net)
    try:
        with torch.inference_mode():
            kw_inf = dict(kw, img_in=s["img"].clone())               # an inference tensor: no version counter
            net.attach_im_feat(kw_inf["img_in"])
            assert calls == {"geo": 1, "tex": 1}
            a = net.render_pifu_nerf(**kw_inf)
            b = net.render_pifu_nerf(**kw_inf)
            assert calls == {"geo": 1, "tex": 1}, calls              # two cameras: no further encoder run
            for k in a:
                assert torch.equal(a[k], b[k]), k
        with torch.no_grad():
            next(net.geo_encoder.parameters()).mul_(1.5)             # a new checkpoint / an optimizer step: in-place, same tensor
        with torch.inference_mode():
            c = net.render_pifu_nerf(**kw_inf)
            assert calls["geo"] == 2 and calls["tex"] == 2, calls    # stale maps are not served
            assert not torch.equal(a["tex_fg_fine"], c["tex_fg_fine"])
            net.render_pifu_nerf(**kw_inf)                           # ... and the fresh ones are kept for the next camera
            assert calls["geo"] == 2, calls
            other = dict(kw_inf, img_in=(kw_inf["img_in"] * 0.5).clone())
            net.render_pifu_nerf(**other)                            # a different source set
            assert calls["geo"] == 3, calls
    finally:
        uninstall(net)


@torch.no_grad()
def test_attached_maps_are_not_served_after_the_module_replaced_them(emulated):
    """The reference overwrites net.feat_geo / net.feat_tex on EVERY attach_*_feat call, also with return_val=True
    (src/model.py:664,678), while net.im only changes without return_val.  Sequence: attach_im_feat(A); render(img_in=B) — the
    encoders run with return_val=True, so the module now holds F(B) although net.im is still A; render(img_in=A) must NOT take the
    attached shortcut (it would return F(B) for A).  Also: a bare attach_tex_feat(B) after attach_im_feat(A) must not pair
    F_geo(A) with F_tex(B)."""
    from keypointnerf_amd.dropin import install, uninstall
    net, s = _net_and_scene()
    A = s["img"]
    B = (s["img"] * 0.5 + 0.1).clone()
    kw = dict(net=net, cam_in=s["cam"], cam_tar=s["cam_tar"], tar_img=None, sp_data=dict(s["sp_data"]),
              objcenter=torch.zeros(1, 3), fine=True, uniform=True, objrad=250., blur=3, level=1, sample_per_ray_c=8,
              sample_per_ray_f=8, src_foreground_mask=s["src_foreground_mask"], bounds=s["bounds"], mask_at_box=torch.ones(1, 16, 16))
    install(net)
    try:
        net.attach_im_feat(A)
        want_a = net.render_pifu_nerf(img_in=A, **kw)                # the attached maps of A
        got_b = net.render_pifu_nerf(img_in=B, **kw)                 # replaces net.feat_geo / feat_tex by F(B)
        again_a = net.render_pifu_nerf(img_in=A, **kw)
        assert not torch.equal(want_a["tex_fg_fine"], got_b["tex_fg_fine"])
        for k in want_a:
            assert torch.equal(want_a[k], again_a[k]), k             # A is rendered from F(A), not from the module's F(B)
        net.attach_im_feat(A)
        net.attach_tex_feat(B)                                       # bare: feat_tex is now F_tex(B), net.im still A
        mixed = net.render_pifu_nerf(img_in=A, **kw)
        for k in want_a:
            assert torch.equal(want_a[k], mixed[k]), k
    finally:
        uninstall(net)


def test_compute_error_matches_the_reference_for_its_other_lambdas(emulated):
    """losses.compute_error against the reference's compute_error (src/utils.py:97-171) beyond the shipped lambdas: l2 / lp /
    mask loss switched on, an ssim weight and `*top*` keys present (both without effect in the reference), weights of 0 —
    same keys, same values, nothing refused."""
    from keypointnerf_amd import losses
    ref_shim.load_reference()
    import src.utils as rutils
    g = torch.Generator().manual_seed(3)
    out = {"tex_cal": torch.rand(1, 3, 8, 8, generator=g), "tex_cal_fine": torch.rand(1, 3, 8, 8, generator=g),
           "alpha": torch.rand(1, 1, 8, 8, generator=g), "alpha_fine": torch.rand(1, 1, 8, 8, generator=g),
           "tar_img": torch.rand(1, 3, 8, 8, generator=g), "tar_alpha": (torch.rand(1, 1, 8, 8, generator=g) > 0.5).float()}
    for lambdas in ({"lambda_l1": 10.0, "lambda_l1_c": 1.0, "lambda_vgg": 0.5},
                    {"lambda_l1": 2.0, "lambda_l1_c": 0.0, "lambda_l2": 3.0, "lambda_lp": 0.5, "lambda_ssim": 1.0, "lambda_mloss": 4.0,
                     "lambda_l1top30": 1.0, "lambda_l2top10": 0.0}):
        want_loss, want = rutils.compute_error(out, None, lambdas)
        got_loss, got = losses.compute_error(out, None, lambdas)
        assert set(got) == set(want), (set(got), set(want))
        for k in want:
            assert abs(float(got[k]) - float(want[k])) <= 1e-6 * max(1.0, abs(float(want[k]))), k
        assert abs(float(got_loss) - float(want_loss)) <= 1e-6 * max(1.0, abs(float(want_loss)))


def test_validation_call_of_the_live_class_consumes_the_same_random_stream(emulated):
    """KeypointNeRF.forward in eval mode (validation_step, src/model.py:509-526 -> :866-884): uniform=False, jittered
    depths and a CPU-drawn importance u.  Same seed => the reference and the drop-in must draw the same numbers."""
    from keypointnerf_amd.dropin import install, uninstall
    net, s = _net_and_scene()
    feat_geo = net.attach_geo_feat(s["img"], return_val=True)
    feat_tex = net.attach_tex_feat(s["img"], return_val=True)
    kw = dict(net=net, img_in=s["img"], cam_in=s["cam"], n_views=3, cam_tar=s["cam_tar"], level=2, stride=torch.tensor([[1, 0]]),
              tar_img=torch.rand(1, 3, 16, 16), bg_img=None, feat_geo=feat_geo, feat_tex=feat_tex, sp_data=dict(s["sp_data"]),
              camcenter=None, objcenter=None, msk=torch.ones(1, 1, 16, 16), src_foreground_mask=s["src_foreground_mask"],
              bounds=s["bounds"], fine=True, uniform=False, blur=3, rand_noise_std=0.01, sample_per_ray_c=12, sample_per_ray_f=6)  # the reference needs R(Sc+Sf) % Sf == 0 (:826)
    with torch.no_grad():
        torch.manual_seed(11)
        ref = net.batch_render_pifu_nerf(**kw)
        after_ref = torch.rand(3)
        install(net)
        torch.manual_seed(11)
        got = net.batch_render_pifu_nerf(**kw)
        after_got = torch.rand(3)
        uninstall(net)
    assert torch.equal(after_ref, after_got)                          # the generator is left in the same state
    assert set(got) == set(ref)
    for k in ref:
        if ref[k] is None:
            assert got[k] is None
            continue
        assert got[k].shape == ref[k].shape, k
        _close(got[k], ref[k], k)


def test_training_step_of_the_live_class_forward_and_gradients(emulated):
    """Train mode (src/model.py:866-884 with net.training): random patch centre (numpy), per-view dropout, density noise,
    stratified depths, importance draws — and loss.backward() into the field's parameters AND, through the feature maps, into
    both encoders."""
    from keypointnerf_amd.dropin import install, uninstall
    net, s = _net_and_scene()
    net.train()
    net.train_out_h = net.train_out_w = 6
    tar = torch.rand(1, 3, 16, 16)
    msk = torch.zeros(1, 1, 16, 16)
    msk[..., 4:12, 4:12] = 1

    def step():
        torch.manual_seed(21)
        np.random.seed(5)
        net.zero_grad(set_to_none=True)
        feat_geo = net.attach_geo_feat(s["img"], return_val=True)
        feat_tex = net.attach_tex_feat(s["img"], return_val=True)
        out = net.batch_render_pifu_nerf(
            net=net, img_in=s["img"], cam_in=s["cam"], n_views=3, cam_tar=s["cam_tar"], level=1, stride=0, tar_img=tar, bg_img=None,
            feat_geo=feat_geo, feat_tex=feat_tex, sp_data=dict(s["sp_data"]), camcenter=None, objcenter=None, msk=msk,
            src_foreground_mask=s["src_foreground_mask"], bounds=s["bounds"], fine=True, uniform=False, blur=3, rand_noise_std=0.01,
            sample_per_ray_c=8, sample_per_ray_f=8)
        loss = (out["tex_fg"] - out["tar_img"]).abs().mean() + 10.0 * (out["tex_fg_fine"] - out["tar_img"]).abs().mean() \
            + 0.1 * out["alpha_fine"].mean() + 0.01 * out["depth"].mean()
        loss.backward()
        grads = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
        return {k: v.detach().clone() for k, v in out.items() if v is not None}, float(loss), grads

    ref_out, ref_loss, ref_g = step()
    install(net)
    got_out, got_loss, got_g = step()
    uninstall(net)
    assert set(got_out) == set(ref_out)
    for k in ref_out:
        _close(got_out[k], ref_out[k], k)
    assert torch.equal(got_out["tar_img"], ref_out["tar_img"])        # same patch centre
    assert float(ref_out["alpha_fine"].max()) > 0.9 and float(ref_out["alpha_fine"].min()) < 0.1 and max(
        float(g.abs().max()) for g in ref_g.values()) > 1e-3          # the patch sees the subject; the gradients are not noise
    assert abs(got_loss - ref_loss) < 1e-5
    assert set(got_g) == set(ref_g)
    hot = [n for n in ref_g if n.startswith(("mlp_geo.", "mlp_tex.", "ibr_compress_gfeat."))]
    enc = [n for n in ref_g if n.startswith(("geo_encoder.", "tex_encoder."))]
    assert len(hot) >= 40 and len(enc) >= 20                         # the field and both encoders receive gradients
    # Absolute floor: a bias in front of a normalisation layer has a gradient that is zero in exact arithmetic — the reference's own
    # value for it is rounding noise (7e-8 against gradients of 1e-3 and more in the same encoder), and so is the difference.
    gmax = max(float(g.abs().max()) for g in ref_g.values())
    for n in ref_g:
        scale = float(ref_g[n].abs().max())
        assert float((got_g[n] - ref_g[n]).abs().max()) <= 2e-4 * scale + max(2e-7, 1e-6 * gmax), (n, scale)
