"""Builds keypointnerf_amd/_lib/libkpnerf_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m keypointnerf_amd.build [--force]
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "_lib", "libkpnerf_hip.so")
# -fno-slp-vectorize: the SLP vectoriser gathers the hand-interleaved scalar VALU work of k_geo_rows_h2 into packed-f32
# lumps (v_pk_mul_f32 ...) placed ahead of the MFMAs they were meant to sit between — measured 7.4 -> 6.8 ms per launch —
# and packed f32 VALU is an anti-lever beside MFMAs anyway (MI355X_MICROARCH.md)
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value", "-fno-slp-vectorize"]


def needs_build():
    if not os.path.exists(OUT):
        return True
    # every file under csrc/ is part of the one translation unit kpn_api.hip (it #includes the other .hip files)
    deps = [os.path.join(CSRC, s) for s in sorted(os.listdir(CSRC)) if s.endswith((".hip", ".h"))]
    deps.append(os.path.join(os.path.dirname(HERE), "include", "kpnerf.h"))
    return any(os.path.getmtime(d) > os.path.getmtime(OUT) for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return OUT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cmd = [hipcc] + HIPCC_FLAGS + [os.path.join(CSRC, "kpn_api.hip"), "-o", OUT]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
