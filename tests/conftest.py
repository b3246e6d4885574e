import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")


def pytest_sessionstart(session):
    # experiments only (scripts/gpu_*.sh): the -m gpu suite against another build of the library, e.g. a code-placement or
    # product-order variant under exp_libs/ — never set by the driver
    path = os.environ.get("KPN_TEST_LIB")
    if path:
        from keypointnerf_amd import lib as kl
        kl._default = kl.KpnLibrary(path)
        print(f"[conftest] GPU suite bound to {path}")
