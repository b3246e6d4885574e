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
# two translation units, compiled in parallel: the library, and the pair-tile rows kernels (csrc/geo_rows_pair_tu.hip).
# -fno-slp-vectorize on both: the fp16 / bf16 operand splits (kpn_common.h) rely on hipcc selecting v_fma_mix_f32 per value
# (the SLP vectoriser turns a pair of them into v_cvt_f32_f16 x2 + v_pk_fma_f32), and packed fp32 beside MFMAs is slower here.
UNITS = [("kpn_api.hip", ["-fno-slp-vectorize"]), ("geo_rows_pair_tu.hip", ["-fno-slp-vectorize"])]


def assembly_files():
    """device assembly kept by the last build, one file per translation unit"""
    return [os.path.join(os.path.dirname(OUT), src.replace(".hip", ".gfx950.s")) for src, _ in UNITS]


def needs_build():
    # (the kept assembly is optional: a deployed _lib with only the .so must not trigger a rebuild — tests/test_isa_audit.py
    # compiles its own -S when the files are absent)
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
        # -save-temps=obj: the device assembly of the very compile that makes the object is kept beside it (<unit>.gfx950.s) for the
        # ISA audit of the CPU suite (tests/test_isa_audit.py, scripts/isa_asm_hazards.py); the other intermediates are removed
        cmd = [hipcc] + HIPCC_FLAGS + extra + ["-save-temps=obj", "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        log = None if verbose else open(obj + ".log", "w+")     # quiet builds keep the compiler's diagnostics for the error message
        objs.append((obj, subprocess.Popen(cmd, stderr=log), log))
    for obj, proc, log in objs:
        rc = proc.wait()
        err = ""
        if log is not None:
            log.seek(0)
            err = log.read()
            log.close()
            os.remove(obj + ".log")
        if rc != 0:
            raise RuntimeError(f"hipcc failed ({rc}) on {obj}:\n{err[-4000:]}")
        stem = obj[:-2]
        for f in os.listdir(os.path.dirname(OUT)):
            full = os.path.join(os.path.dirname(OUT), f)
            if not f.startswith(os.path.basename(stem) + "-") and not f.startswith(os.path.basename(stem) + ".hip-"):
                continue
            if f.endswith("-gfx950.s"):
                os.replace(full, stem + ".gfx950.s")
            else:
                os.remove(full)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + [o for o, _, _ in objs] + ["-o", OUT]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
