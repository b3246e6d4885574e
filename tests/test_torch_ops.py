"""torch.ops.kpnerf.* registrations (keypointnerf_amd/torch_ops.py): schemas + shape inference on CPU, numerics on GPU."""
import numpy as np
import pytest
import torch


def test_ops_are_registered_with_fake_kernels():
    import keypointnerf_amd.torch_ops  # noqa: F401
    for name in ("rgba2out", "importance_sample", "ray_bbox_intersection", "field_query"):
        assert hasattr(torch.ops.kpnerf, name)
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        rgba, z = torch.empty(1, 7, 16, 5, device="cuda"), torch.empty(1, 7, 16, device="cuda")
        color, depth, alpha, contrib, sdf = torch.ops.kpnerf.rgba2out(rgba, z)
        assert color.shape == (1, 7, 3) and contrib.shape == (1, 7, 16) and sdf.shape == (1, 7)
        s = torch.ops.kpnerf.importance_sample(torch.empty(1, 7, 14, device="cuda"), torch.empty(1, 7, 15, device="cuda"), 9)
        assert s.shape == (1, 7, 9)
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
