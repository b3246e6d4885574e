"""CPU checks of the product library: it loads, exports every symbol include/kpnerf.h declares, has no
CPU fallback, and packs weights consistently with the emulator build (no compute is launched)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "kpnerf.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(kpn_[a-z0-9_]+)\s*\(", hdr)))


def test_hip_library_builds_loads_and_exports_the_abi():
    from keypointnerf_amd import build as kb
    from keypointnerf_amd import lib as kl
    so = kb.build(verbose=False)
    L = kl.KpnLibrary(so)
    assert L.kpn_abi_version() == kl.ABI_VERSION
    assert L.kpn_is_device_build() == 1
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(L.cdll, name), f"{name} declared in include/kpnerf.h but not exported"
    assert sorted(declared) == L.exported_symbols(), "lib.py binding and header differ"
    # the device code object is gfx950
    blob = open(so, "rb").read()
    assert b"gfx950" in blob


def test_argument_errors_are_reported_not_crashed():
    from keypointnerf_amd import lib as kl
    L = kl.get_library()
    assert L.kpn_pack_weights(None, None) == -1
    assert b"null" in L.kpn_last_error()
    d = kl.SceneDesc()
    assert L.kpn_scene_workspace_bytes(ctypes.byref(d)) == 0
    assert L.kpn_query_workspace_bytes(0, 3) == 0


def test_no_cpu_fallback():
    import torch
    from keypointnerf_amd import lib as kl
    from keypointnerf_amd import ops
    with pytest.raises(kl.KpnError):
        kl.KpnLibrary("/nonexistent/libkpnerf_hip.so")
    with pytest.raises(RuntimeError):
        ops.rgba2out(torch.zeros(1, 2, 4, 5), torch.zeros(1, 2, 4))
    # the product package never references the oracle or the emulator
    for fn in os.listdir(os.path.join(ROOT, "keypointnerf_amd")):
        if fn.endswith(".py"):
            src = open(os.path.join(ROOT, "keypointnerf_amd", fn)).read()
            assert "import oracle" not in src and "from oracle" not in src and "simt_harness" not in src, fn


def test_weight_packing_matches_emulator_build():
    from keypointnerf_amd import lib as kl
    from keypointnerf_amd.synthetic import random_hotpath_state_dict
    from keypointnerf_amd.weights import effective_weights, flatten_plain
    from tests import simt_harness as sh
    plain = flatten_plain(effective_weights(random_hotpath_state_dict(seed=5)))
    outs = []
    for L in (kl.get_library(), sh.simt_lib()):
        assert plain.size == L.kpn_plain_weight_floats()
        packed = np.zeros(L.kpn_packed_weight_floats(), np.float32)
        L.check(L.kpn_pack_weights(plain.ctypes.data_as(ctypes.c_void_p), packed.ctypes.data_as(ctypes.c_void_p)))
        outs.append(packed)
    assert np.array_equal(outs[0], outs[1])
    # every real weight appears in the stream at least once (nothing dropped by the permutations)
    nz_plain = np.unique(np.abs(plain[np.abs(plain) > 0]))
    assert np.isin(nz_plain[:-1], np.abs(outs[0])).mean() > 0.999
