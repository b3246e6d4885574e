"""The eager-PyTorch restatement (oracle/torch_eager.py, measurement infrastructure for the unfused-PyTorch-on-MI355X
baseline) pinned to the reference's recorded outputs."""
import numpy as np
import pytest
import torch

from oracle import torch_eager as te
from tests.golden_io import CASES, load_case, load_weights, out_as_rays, pixel_list


@pytest.mark.parametrize("case", CASES + ["case_n_v3_sigma_nofine"])
def test_eager_restatement_vs_reference(case):
    from keypointnerf_amd.weights import effective_weights, flatten_plain
    scene, cfg, g = load_case(case)
    P = te.unpack_plain(torch.from_numpy(flatten_plain(effective_weights(load_weights()))))
    pix, _ = pixel_list(cfg, scene["cam_tar"])
    fine = "out.tex_fg_fine" in g
    sigma = 0.25 if case.endswith("nofine") else 0.1
    with torch.no_grad():
        o = te.render_rays(P, scene, torch.from_numpy(pix).long(), cfg["Sc"], cfg["Sf"], fine=fine, sigma=sigma)
    for k in ("tex_fg", "alpha") + (("tex_fg_fine", "alpha_fine") if fine else ()):
        assert np.abs(o[k].numpy() - out_as_rays(g, k)).max() < 2e-5, k
    for k in ("depth",) + (("depth_fine", "sdf") if fine else ()):
        np.testing.assert_allclose(o[k].numpy(), out_as_rays(g, k), rtol=2e-4, atol=2e-4)
