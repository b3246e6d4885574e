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
