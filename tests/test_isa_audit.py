"""Build-time audit of the gfx950 assembly (CPU suite; hipcc cross-compiles without a GPU): the wait-state pairs hipcc does NOT pad
because one side sits inside an inline-asm statement (scripts/isa_asm_hazards.py).  Round 4 shipped a per-point kernel that was
right by luck of register allocation: in the other product order the allocator put an asm statement's outputs into dead registers of
an in-flight MFMA's destination tuple and the MFMA's write-back overwrote them (wrong on 62 % of the points, different from run to
run; scripts/repro_asm_waw_hazard.hip, profiles/r05_a_waw_hazard.txt).  Since round 5 that kernel has no VALU instruction inside an
asm statement; the hand-placed streams of the pair-tile rows kernels keep theirs and are held to the distances below on every build."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
HIPCC = "/opt/rocm/bin/hipcc"

pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")


@pytest.fixture(scope="module")
def assembly(tmp_path_factory):
    from keypointnerf_amd import build as kb
    # the assembly the library's own build kept (same compile as the shipped objects); compiled here only if that is stale
    if not kb.needs_build() and all(os.path.exists(f) for f in kb.assembly_files()):
        return kb.assembly_files()
    out = tmp_path_factory.mktemp("isa")
    procs = []
    for src, extra in kb.UNITS:
        dst = str(out / src.replace(".hip", ".s"))
        cmd = [HIPCC] + kb.HIPCC_FLAGS + extra + ["-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only", os.path.join(kb.CSRC, src), "-o", dst]
        procs.append((dst, subprocess.Popen(cmd, stderr=subprocess.DEVNULL)))
    files = []
    for dst, p in procs:
        assert p.wait() == 0, dst
        files.append(dst)
    return files


def kernels_of(path):
    return sorted(set(re.findall(r'^\s*\.amdhsa_kernel\s+(\S+)', open(path).read(), re.M)))


def test_no_unpadded_mfma_pair_around_asm_statements(assembly):
    import isa_asm_hazards as ia
    n_kernels = n_asm_valu = 0
    for path in assembly:
        for k in kernels_of(path):
            n_kernels += 1
            found, ex = ia.audit(path, k, window=12, mfma_states=True)
            # B: an asm instruction reads an MFMA result, C: overwrites a register of an in-flight MFMA's C/D tuple — 12 states is
            # what hipcc itself puts between such a pair when it can see both sides (lost up to 4, safe from 6 on the MI355X)
            for cls in "BC":
                assert not found[cls], f"{k}: class {cls} pairs {dict(found[cls])}, e.g. {list(ex[cls].values())[:2]}"
            # A: a register written inside an asm statement is an MFMA operand: at least hipcc's two wait states
            short = {st: c for st, c in found["A"].items() if st < 2}
            assert not short, f"{k}: asm VALU write -> MFMA operand after {short} states, e.g. {[ex['A'][s] for s in short][:2]}"
            n_asm_valu += sum(found["A"].values())
    assert n_kernels >= 30


def test_per_point_kernel_has_no_valu_instruction_in_asm(assembly):
    import isa_asm_hazards as ia
    path = [p for p in assembly if os.path.basename(p).startswith("kpn_api.")][0]
    # every instantiation of the two-fp16-piece per-point body: the fused kernel (any V / V = 3 unrolled: the one that ships) and
    # the density-first passes (round 5's test looked at whichever sorted first)
    ks = [k for k in kernels_of(path) if any(n in k for n in ("k_fuse_color_h", "k_density_h", "k_colour_h"))]
    assert len(ks) == 5, ks
    for k in ks:
        prog = ia.parse(path, k)
        inside = [i["text"] for i in prog if i["kind"] == "ins" and i["asm"] >= 0 and i["name"].startswith("v_")]
        assert not inside, (k, inside[:4])
        # and the operand splits are the four-instruction form per pair of values (kpn_common.h kpn_split_f16x8)
        names = [i["name"] for i in prog if i["kind"] == "ins"]
        n_mix, n_cvt = sum(n == "v_fma_mix_f32" for n in names), sum(n == "v_cvt_pk_f16_f32" for n in names)
        assert n_mix >= (100 if "k_density_h" in k else 180) and abs(n_mix - n_cvt) <= 8, (k, n_mix, n_cvt)
        assert not any(n.startswith("v_cvt_f32_f16") for n in names), "the fp16 halves are read in place by v_fma_mix_f32"


def test_the_audit_finds_the_round4_miscompute():
    """The instruction window of the build that was wrong and non-deterministic on the MI355X (round 4, `hh hl lh`; kept as a fixture):
    an asm statement's `v_cvt_pk_f16_f32 v56` three states behind `v_mfma ... v[48:63]`, whose write-back overwrote it.  The audit must
    flag it as class C (an asm instruction overwrites a register of an MFMA in flight) — what test_no_unpadded_mfma_pair_around_asm_
    statements asserts the shipped library is free of."""
    import isa_asm_hazards as ia
    found, ex = ia.audit(os.path.join(ROOT, "tests", "fixtures", "isa_wrong_build_window.s"), "_Z14k_fuse_color_hEXCERPT", window=12, mfma_states=True)
    assert found["C"], "the write-after-write pair of the wrong build went unnoticed"
    assert min(found["C"]) <= 4 and any("v_cvt_pk_f16_f32 v56" in e[1] for e in ex["C"].values())


def test_audit_classes_on_synthetic_streams(tmp_path):
    """The four classes of scripts/isa_asm_hazards.py on hand-written streams: what each one flags and what it lets pass."""
    import isa_asm_hazards as ia

    def run(body, **kw):
        p = tmp_path / "k.s"
        p.write_text("_Z1kv:\n" + body + "\ts_endpgm\n")
        return ia.audit(str(p), "_Z1kv", **kw)[0]

    mfma = "\tv_mfma_f32_32x32x16_f16 v[32:47], v[8:11], v[12:15], v[32:47]\n"
    asm = lambda ins: "\t;;#ASMSTART\n" + "".join("\t" + i + "\n" for i in ins) + "\t;;#ASMEND\n"
    # A: an operand written inside asm, read by the MFMA one state later (hipcc itself would put two wait states there)
    f = run(asm(["v_cvt_pk_f16_f32 v12, v1, v2"]) + mfma)
    assert f["A"] == {0: 1} and not f["B"] and not f["C"]
    f = run(asm(["v_cvt_pk_f16_f32 v12, v1, v2", "s_nop 1"]) + mfma)
    assert f["A"] == {2: 1}
    # B: asm reads the MFMA's result; C: asm overwrites a register of its destination tuple; the same instructions OUTSIDE asm are
    # hipcc's to pad and are not reported
    f = run(mfma + "\tv_add_f32 v1, v2, v3\n" + asm(["v_exp_f32 v5, v33"]))
    assert f["B"] == {1: 1} and not f["C"]
    f = run(mfma + asm(["v_mov_b32 v40, 0"]))
    assert f["C"] == {0: 1}
    f = run(mfma + "\tv_mov_b32 v40, 0\n\tv_exp_f32 v5, v33\n")
    assert not f["B"] and not f["C"]
    # an accumulate chain is not a hazard, and MFMAs in between count as their issue interval when asked to
    f = run(mfma + mfma + asm(["v_mov_b32 v40, 0"]), window=12, mfma_states=True)
    assert f["C"] == {0: 1, 8: 1}
    f = run(mfma + "\tv_mfma_f32_32x32x16_f16 v[48:63], v[8:11], v[12:15], v[48:63]\n" * 2 + asm(["v_mov_b32 v40, 0"]), window=12, mfma_states=True)
    assert not f["C"]                      # 16 states of other MFMAs in between: the write-back has long landed
    # D (overwriting an A / B operand behind the MFMA) is reported for information only: operands are captured at issue
    f = run(mfma + asm(["v_mov_b32 v8, 0"]))
    assert f["D"] == {0: 1} and not f["C"]
