"""Short soak of the two field kernels as a gate (-m gpu): the same 400,000 points evaluated again and again must come out
bit-identical, and within fp32 class of the fp32-MFMA kernels.  The default kernels carry fp32 through 16-bit matrix operands in
hand-placed (rows kernel) and compiler-scheduled (per-point kernel, two waves per SIMD) instruction streams; a build of the
per-point kernel whose product order differed from the shipped one was wrong on 62 % of these points and different from run to
run on the MI355X while the emulator agreed with the oracle — this is the test that build fails at once.
scripts/soak_mode2.py runs the same loop for 10^10 row evaluations."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mask", ["dense", "ellipsoid"])
def test_field_kernels_are_deterministic_and_fp32_class(mask):
    from keypointnerf_amd import ops
    from keypointnerf_amd.synthetic import make_scene, random_hotpath_state_dict, to_device
    dev = torch.device("cuda", 0)
    sc = to_device(make_scene(n_views=3, src_hw=(256, 256), tar_hw=(64, 64), mask=mask, seed=1), dev)
    w = ops.PackedWeights(random_hotpath_state_dict(seed=3), device=dev)
    ps = ops.PreparedScene(sc["img"], sc["cam"], sc["feat_geo"], sc["feat_tex"], sc["sp_data"], sc["src_foreground_mask"])
    lo, hi = sc["bounds"].reshape(2, 3)[0], sc["bounds"].reshape(2, 3)[1]
    gen = torch.Generator(device="cuda").manual_seed(4)
    n = 400000
    P = (lo + (hi - lo) * (0.2 + 0.6 * torch.rand(n, 3, device=dev, generator=gen)))[None]
    V = torch.nn.functional.normalize(torch.randn(n, 3, device=dev, generator=gen), dim=-1)[None]
    rows_default, fuse_default = ops.get_geo_rows_mode(), ops.get_fuse_mode()
    try:
        ops.set_geo_rows_mode(0); ops.set_fuse_mode(0)
        ref32, valid = ops.query(ps, w, P, V, mode=1)             # fp32 MFMA in both field kernels
        ref32 = ref32.clone()
        assert int(valid.sum()) > n // 4
        ops.set_fuse_mode(fuse_default)
        ref = ops.query(ps, w, P, V, mode=1)[0].clone()           # fp32-MFMA rows kernel, default per-point kernel
        scale = ref.abs().amax(dim=(0, 1))
        ops.set_geo_rows_mode(rows_default)
        first = ops.query(ps, w, P, V, mode=1)[0].clone()
        assert torch.isfinite(first).all()
        off = int((((first - ref).abs() > 2e-5 * scale + 1e-6).any(-1)).sum())                  # the bar of scripts/soak_mode2.py
        assert off == 0, f"{off} of {n} points beyond fp32-class distance of the fp32-MFMA rows kernel"
        off = int((((first - ref32).abs() > 2e-4 * scale + 1e-5).any(-1)).sum())
        assert off == 0, f"{off} of {n} points far from the fp32 kernels' results"
        differing = 0
        for _ in range(300):
            differing += int((ops.query(ps, w, P, V, mode=1)[0] != first).any(-1).sum())
        assert differing == 0, f"{differing} point results differed between runs"
    finally:
        ops.set_geo_rows_mode(rows_default); ops.set_fuse_mode(fuse_default)


@pytest.mark.parametrize("bias", [-20.0, -30.0])
def test_density_first_frames_are_deterministic(bias):
    """Round 6: the density-first passes build their live list with atomics (its order differs from run to run) and pass B reads the
    scratch through it: 150 renders of a 128 x 128 frame whose hull is 25 % / 82 % empty, density first forced on, every output of
    every render bit-identical to the first — and to the fused kernel's frame."""
    from keypointnerf_amd import lib as kl
    from keypointnerf_amd import ops
    from keypointnerf_amd.synthetic import make_scene, random_hotpath_state_dict, to_device
    L = kl.get_library()
    dev = torch.device("cuda", 0)
    sc = to_device(make_scene(n_views=3, src_hw=(256, 256), tar_hw=(128, 128), mask="ellipsoid", seed=1, tar_focal_at_512=800.0), dev)
    w = ops.PackedWeights(random_hotpath_state_dict(seed=3, density_bias=bias), device=dev)
    ps = ops.PreparedScene(sc["img"], sc["cam"], sc["feat_geo"], sc["feat_tex"], sc["sp_data"], sc["src_foreground_mask"])
    plan = ops.RenderPlan(ps, (0, 0, 1, 128, 128), 64, 64, fine=True)
    try:
        L.check(L.kpn_set_density_first(0))
        fused = {k: v.clone() for k, v in ops.render_rays(ps, w, sc["cam_tar"], sc["bounds"], plan=plan).items()}
        L.check(L.kpn_set_density_first(1))
        differing = 0
        for _ in range(150):
            out = ops.render_rays(ps, w, sc["cam_tar"], sc["bounds"], plan=plan)
            differing += sum(int((out[k] != fused[k]).sum()) for k in fused)
        assert differing == 0, f"{differing} output values differed"
        assert float(fused["alpha_fine"].max()) > 0.05
    finally:
        L.check(L.kpn_set_density_first(2))
