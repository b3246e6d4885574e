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
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"]
# two translation units: the library, and the pair-tile rows kernel with its own flag (see csrc/geo_rows_pair_tu.hip)
UNITS = [("kpn_api.hip", []), ("geo_rows_pair_tu.hip", ["-fno-slp-vectorize"])]


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
    objs = []
    for src, extra in UNITS:
        obj = os.path.join(os.path.dirname(OUT), src.replace(".hip", ".o"))
        cmd = [hipcc] + HIPCC_FLAGS + extra + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        objs.append((obj, subprocess.Popen(cmd)))
    for obj, proc in objs:
        if proc.wait() != 0:
            raise subprocess.CalledProcessError(proc.returncode, obj)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + [o for o, _ in objs] + ["-o", OUT]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
