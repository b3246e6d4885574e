"""Import the *unmodified* reference (facebookresearch/KeypointNeRF, mounted read-only at
/root/reference) under a stub shim so that its hot path can run on CPU in the build container.

TEST INFRASTRUCTURE ONLY.  Nothing under ``keypointnerf_amd/`` may import this module; it exists to
(1) validate the C restatement in ``oracle/kpnerf_oracle.c`` and (2) generate the committed golden
vectors in ``tests/golden/`` (see ``oracle/make_golden.py``).  ``/root/reference`` does not exist on
the GPU box, so nothing that runs there (``-m gpu`` tests, ``smoke()``, ``bench.py``) imports it.

Why a shim is needed:
  * top-level imports of cv2 / kornia / pytorch_lightning / torchvision / imageio / skimage /
    argcomplete (reference src/model.py:12-20, src/utils.py:12-14) are not installed here;
  * ``VGGLoss`` downloads VGG19 and calls ``.cuda()`` (reference src/utils.py:757,792);
  * the hot path hard-codes ``.cuda()`` (reference src/model.py:920,1004,1013,1020).
No reference source is copied: the modules are imported from where they lie.
"""
import json
import os
import sys
import types

import torch

REFERENCE_ROOT = os.environ.get("KPNERF_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "src", "model.py"))


_loaded = {}


def load_reference():
    """Returns the reference's ``src.model`` module (with ``src.utils`` / ``src.spatial`` loaded)."""
    if "model" in _loaded:
        return _loaded["model"]
    if not reference_available():
        raise RuntimeError(f"reference not found under {REFERENCE_ROOT}")
    sys.dont_write_bytecode = True  # /root/reference is read-only

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules.setdefault(name, m)
        return sys.modules[name]

    stub("cv2")
    k = stub("kornia")
    ku = stub("kornia.utils", tensor_to_image=lambda x: x)
    k.utils = ku
    kg = stub("kornia.geometry")
    kgc = stub("kornia.geometry.conversions", convert_points_to_homogeneous=None)
    k.geometry = kg
    kg.conversions = kgc
    pl = stub("pytorch_lightning", LightningModule=torch.nn.Module)
    plu = stub("pytorch_lightning.utilities")
    plaf = stub("pytorch_lightning.utilities.apply_func", move_data_to_device=lambda b, d: b)
    pl.utilities = plu
    plu.apply_func = plaf
    tv = stub("torchvision")
    tv.models = stub("torchvision.models")
    tv.transforms = stub("torchvision.transforms")
    stub("imageio")
    sk = stub("skimage")
    sk.metrics = stub("skimage.metrics", structural_similarity=None)
    stub("argcomplete")

    if not torch.cuda.is_available():
        # the reference hard-codes .cuda(); on the CPU-only build box make it the identity
        torch.Tensor.cuda = lambda self, *a, **k: self

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import src.utils as rutils  # noqa: E402

    class _NoVGG(torch.nn.Module):
        def forward(self, *a, **k):
            return torch.zeros(())

    rutils.VGGLoss = _NoVGG
    import src.model as rmodel  # noqa: E402

    rmodel.VGGLoss = _NoVGG
    _loaded["model"] = rmodel
    return rmodel


def load_config():
    with open(os.path.join(REFERENCE_ROOT, "configs", "zju.json")) as f:
        return json.load(f)


def build_reference_net(seed=0):
    """KeypointNeRF(cfg) with the reference's own seeded init.  ``torch.manual_seed(seed)`` must
    precede construction because the weight-norm g/v tensors keep nn.Linear's constructor-time
    init."""
    rmodel = load_reference()
    cfg = load_config()
    torch.manual_seed(seed)
    net = rmodel.KeypointNeRF(cfg)
    net.eval()
    return net
