"""torch.ops.kpnerf.* registrations (keypointnerf_amd/torch_ops.py): schemas + shape inference on CPU, numerics on GPU."""
import numpy as np
import pytest
import torch


def test_ops_are_registered_with_fake_kernels():
    import keypointnerf_amd.torch_ops  # noqa: F401
    for name in ("rgba2out", "rgba2out_backward", "importance_sample", "ray_bbox_intersection", "field_query", "render_rays",
                 "render_rays_train", "render_rays_train_backward"):
        assert hasattr(torch.ops.kpnerf, name)
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        rgba, z = torch.empty(1, 7, 16, 5, device="cuda"), torch.empty(1, 7, 16, device="cuda")
        color, depth, alpha, contrib, sdf = torch.ops.kpnerf.rgba2out(rgba, z)
        assert color.shape == (1, 7, 3) and contrib.shape == (1, 7, 16) and sdf.shape == (1, 7)
        s = torch.ops.kpnerf.importance_sample(torch.empty(1, 7, 14, device="cuda"), torch.empty(1, 7, 15, device="cuda"), 9)
        assert s.shape == (1, 7, 9)
        e = lambda *sh, **k: torch.empty(*sh, device="cuda", **k)
        R, V = 10, 3
        outs = torch.ops.kpnerf.render_rays_train(
            e(100), e(V, 64, 8, 8), e(V, 8, 32, 32), e(V, 8, 16, 16), e(V, 3, 64, 64), e(V, 4, 4), e(V, 4, 4), e(1, 24, 3), None,
            [2.0, 5.0, 100.0, 0.1], e(1, 4, 4), e(1, 4, 4), e(1, 2, 3), 2.0, 8.0, e(R, 2, dtype=torch.int32), e(1, R, 8), e(1, R, 8),
            None, None, 7, 7, 0.0, 8, 8)
        assert [tuple(o.shape) for o in outs[:7]] == [(1, 3, R), (1, R), (1, R), (1, 3, R), (1, R), (1, R), (1, R)]
        assert outs[7].dtype == torch.uint8                                # the pass state (empty unless keep_state)
        outs = torch.ops.kpnerf.render_rays(e(16), [V, 64, 64, 8, 8, 32, 32, 16, 16, 0], [2.0, 5.0, 100.0, 0.1], e(16), e(1, 4, 4),
                                            e(1, 4, 4), e(1, 2, 3), 2.0, 8.0, [0, 0, 1, 6, 5], 8, 8, True)
        assert tuple(outs[0].shape) == (1, 3, 5, 6) and tuple(outs[6].shape) == (1, 5, 6)
    # no CPU kernel exists: the dispatcher refuses CPU tensors
    with pytest.raises((NotImplementedError, RuntimeError)):
        torch.ops.kpnerf.rgba2out(torch.zeros(1, 2, 4, 5), torch.zeros(1, 2, 4))


@pytest.mark.gpu
def test_custom_ops_match_direct_calls():
    import keypointnerf_amd.torch_ops  # noqa: F401
    from keypointnerf_amd import ops
    from keypointnerf_amd.synthetic import to_device
    from tests.golden_io import CASES, load_case, load_weights
    scene, cfg, g = load_case(CASES[0])
    s = to_device(scene, "cuda")
    ps = ops.PreparedScene(s["img"], s["cam"], s["feat_geo"], s["feat_tex"], s["sp_data"], s["src_foreground_mask"])
    w = ops.PackedWeights(load_weights())
    pts, view = torch.from_numpy(g["query.0.pts"]).cuda(), torch.from_numpy(g["query.0.view"]).cuda()
    ws, dims, scal = ps.as_op_args()
    out, valid = torch.ops.kpnerf.field_query(ws, dims, scal, w.tensor, pts, view, 0)
    ref_out, ref_valid = ops.query(ps, w, pts, view, mode=0)
    assert torch.equal(out, ref_out) and torch.equal(valid, ref_valid)
    rgba, z = torch.from_numpy(g["rgba2out.0.rgba"]).cuda(), torch.from_numpy(g["rgba2out.0.z"]).cuda()
    a, b = torch.ops.kpnerf.rgba2out(rgba, z), ops.rgba2out(rgba, z)
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    assert np.abs(a[0].cpu().numpy() - g["rgba2out.0.color"]).max() < 3e-6


@pytest.mark.gpu
def test_rgba2out_op_is_differentiable():
    """torch.ops.kpnerf.rgba2out carries a registered autograd formula (kpnerf::rgba2out_backward = kpn_rgba2out_backward):
    same gradient as the reference's formula differentiated by torch autograd."""
    import keypointnerf_amd.torch_ops  # noqa: F401
    torch.manual_seed(0)
    rgba = torch.rand(1, 33, 24, 5, device="cuda").requires_grad_(True)   # sigma in [0,1): no saturated transmittance
    z = (torch.rand(1, 33, 24, device="cuda") * 3 + 2).sort(-1)[0]
    color, depth, alpha, contrib, sdf = torch.ops.kpnerf.rgba2out(rgba, z)
    G = [torch.rand_like(t) for t in (color, depth, alpha, sdf)]
    (color * G[0]).sum().add((depth * G[1]).sum()).add((alpha * G[2]).sum()).add((sdf * G[3]).sum()).backward()
    got = rgba.grad.clone()
    # src/model.py:1150-1176 in torch
    r = rgba.detach().clone().requires_grad_(True)
    dist = torch.cat([z[..., 1:] - z[..., :-1], torch.full_like(z[..., :1], 1e10)], -1)
    a = 1.0 - torch.exp(-r[..., 0] * dist)
    c = a * torch.cumprod(torch.cat([torch.ones_like(a[..., :1]), 1 - a[..., :-1]], -1), -1)
    col = (c[..., None] * r[..., 2:]).sum(-2)
    al = c.sum(-1)
    dep = (c * z).sum(-1) / (al + 1e-8)
    sd = (c * r[..., 1]).sum(-1) / (al + 1e-8)
    ((col * G[0]).sum() + (dep * G[1]).sum() + (al * G[2]).sum() + (sd * G[3]).sum()).backward()
    assert float((color - col).abs().max()) < 1e-5
    scale = float(r.grad.abs().max())
    assert float((got - r.grad).abs().max()) <= 2e-4 * scale
