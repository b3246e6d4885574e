"""CPU checks of install(): attribute rebinding on the REAL reference class (build container only) and
forwarding of the training path to the reference's own methods."""
import pytest
import torch

from oracle import ref_shim

pytestmark = pytest.mark.skipif(not ref_shim.reference_available(), reason="needs /root/reference (build container)")


def test_install_rebinds_only_the_seam_and_keeps_the_checkpoint_format():
    from keypointnerf_amd.dropin import _SEAMS, install, uninstall
    net = ref_shim.build_reference_net(seed=0)
    keys_before = list(net.state_dict().keys())
    install(net)
    for k in _SEAMS:
        assert k in net.__dict__, k
    assert list(net.state_dict().keys()) == keys_before          # parameter names untouched
    # eval-mode calls demand GPU tensors: there is no CPU fallback
    with pytest.raises(RuntimeError):
        net.rgba2out(torch.zeros(1, 2, 4, 5), torch.zeros(1, 2, 4))
    uninstall(net)
    for k in _SEAMS:
        assert k not in net.__dict__
    c = net.rgba2out(torch.rand(1, 2, 4, 5), torch.rand(1, 2, 4).sort(-1)[0])  # the reference's own staticmethod again
    assert c[0].shape == (1, 2, 3)


def test_reference_keyword_call_sites_bind_to_the_installed_seams():
    """The reference calls both render seams with `net=` BY KEYWORD (src/model.py:454-472 render_full_nerf_image,
    :866-884 KeypointNeRF.forward).  After install() the very same keyword sets must bind (a `net_` parameter name used
    to swallow `net=` into **config -> TypeError) and reach this library, which refuses CPU tensors."""
    import inspect
    from keypointnerf_amd.dropin import install
    from keypointnerf_amd.synthetic import make_scene
    net = ref_shim.build_reference_net(seed=0)
    ref_sig = {k: inspect.signature(getattr(type(net), k)) for k in ("render_pifu_nerf", "batch_render_pifu_nerf")}
    install(net)
    for k, sig in ref_sig.items():                                   # parameter names and order as the reference's
        ours = inspect.signature(getattr(net, k))
        assert list(ours.parameters) == list(sig.parameters), k
        for name, prm in sig.parameters.items():
            assert ours.parameters[name].default == prm.default or prm.default is inspect._empty or isinstance(prm.default, dict), (k, name)
    s = make_scene(n_views=3, src_hw=(64, 64), tar_hw=(64, 64), mask="ellipsoid", seed=5)
    feats = [t for t in s["feat_geo"]], s["feat_tex"]
    net.attach_geo_feat = lambda im, return_val=False: feats[0]      # skip the 28 M-parameter encoders on the CPU
    net.attach_tex_feat = lambda im, return_val=False: feats[1]
    with pytest.raises(RuntimeError, match="GPU"):                   # src/model.py:454-472
        net.render_pifu_nerf(net=net, img_in=s["img"], cam_in=s["cam"], cam_tar=s["cam_tar"], tar_img=None, sp_data=s["sp_data"],
                             objcenter=torch.zeros(1, 3), fine=True, uniform=True, objrad=250., blur=3, level=1,
                             sample_per_ray_c=8, sample_per_ray_f=8, src_foreground_mask=s["src_foreground_mask"],
                             bounds=s["bounds"], mask_at_box=torch.ones(1, 64, 64))
    # src/model.py:866-884 (eval mode = validation_step): served by torch.ops.kpnerf.render_rays_train, which has no CPU kernel
    with pytest.raises((RuntimeError, NotImplementedError), match="kpnerf::render_rays_train.*CPU"):
        net.batch_render_pifu_nerf(net=net, img_in=s["img"], cam_in=s["cam"], n_views=3, cam_tar=s["cam_tar"], level=5,
                                   stride=torch.zeros(1, 2, dtype=torch.long), tar_img=None, bg_img=None, feat_geo=feats[0],
                                   feat_tex=feats[1], sp_data=s["sp_data"], camcenter=None, objcenter=None,
                                   msk=torch.ones(1, 1, 64, 64), src_foreground_mask=s["src_foreground_mask"],
                                   bounds=s["bounds"], **net.kwargs["dr_kwargs"])


def test_install_refuses_unsupported_spatial_encoders():
    from keypointnerf_amd.dropin import encoder_sigma, install
    net = ref_shim.build_reference_net(seed=0)
    assert encoder_sigma(net) == 0.1                                 # configs/zju.json:43
    net.sp_encoder.kwargs.pop("sigma")
    assert encoder_sigma(net) == 150.0                               # the reference's own fallback, src/spatial.py:112
    net.sp_encoder.sp_type = "rel_z"                                 # same feature width, different arithmetic
    with pytest.raises(NotImplementedError):
        install(net)


def test_install_loss_rebinds_the_global_forward_resolves():
    """KeypointNeRF.forward calls the module-global compute_error (src/model.py:894): install_loss swaps exactly that name."""
    from keypointnerf_amd import losses
    rmodel = ref_shim.load_reference()
    ref = losses.install_loss(rmodel)
    try:
        assert rmodel.compute_error is losses.compute_error
        assert rmodel.KeypointNeRF.forward.__globals__["compute_error"] is losses.compute_error
    finally:
        rmodel.compute_error = ref
    import inspect
    assert list(inspect.signature(losses.compute_error).parameters) == list(inspect.signature(ref).parameters)
