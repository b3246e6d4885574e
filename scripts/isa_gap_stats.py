#!/usr/bin/env python
"""Issue-slot audit of k_geo_rows_h2 from hipcc's assembly (-save-temps): instruction mix, the instructions between consecutive
MFMAs ("gaps"; a transcendental counts two slots), s_nops, spill traffic.  python scripts/isa_gap_stats.py <file.s> [kernel]"""
import collections
import re
import sys

path = sys.argv[1]
kernel = sys.argv[2] if len(sys.argv) > 2 else "k_geo_rows_h2"
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\d+%s\w*:" % kernel, l))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
ins = [l.strip() for l in lines[start:end] if l.startswith("\t") and not l.strip().startswith((".", ";"))]
ops = [l.split()[0] for l in ins]
c = collections.Counter(ops)
TRANS = ("v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32")
def cls(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith(TRANS): return "trans"
    if op.startswith("v_"): return "valu"
    if op == "s_nop": return "nop"
    if op == "s_waitcnt": return "waitcnt"
    if op.startswith("s_"): return "salu"
    if op.startswith(("global_", "scratch_", "buffer_", "flat_")): return "vmem"
    if op.startswith("ds_"): return "lds"
    return "other"
k = collections.Counter(cls(o) for o in ops)
print("kernel %s: %d instructions" % (kernel, len(ops)), dict(k))
n = k["mfma"]
print("per MFMA: valu %.2f  trans %.2f  salu %.2f  nop %.2f  vmem %.2f  waitcnt %.2f  | slots (trans = 2, waitcnt = 0) %.2f" % (
    k["valu"] / n, k["trans"] / n, k["salu"] / n, k["nop"] / n, k["vmem"] / n, k["waitcnt"] / n,
    (k["valu"] + 2 * k["trans"] + k["salu"] + k["nop"] + k["vmem"] + k["lds"]) / n))
print("spills: scratch ops %d, v_readlane %d, v_writelane %d, v_mov %d, v_accvgpr_read %d write %d" % (
    sum(v for o, v in c.items() if o.startswith("scratch_")), c["v_readlane_b32"], c["v_writelane_b32"], c["v_mov_b32_e32"],
    c["v_accvgpr_read_b32"], c["v_accvgpr_write_b32"]))
# gaps
gaps, cur, seen = [], 0, False
for o in ops:
    t = cls(o)
    if t == "mfma":
        if seen: gaps.append(cur)
        seen, cur = True, 0
    elif seen:
        cur += 2 if t == "trans" else (0 if t == "waitcnt" else 1)
h = collections.Counter(min(g, 12) for g in gaps)
print("slots between consecutive MFMAs (12 = 12 or more):", " ".join("%d:%d" % (g, h[g]) for g in sorted(h)))
big = sorted(gaps, reverse=True)[:12]
print("largest gaps:", big, " sum of slots beyond 5 per gap:", sum(max(0, g - 5) for g in gaps))
print("top ops:", ", ".join("%s %d" % kv for kv in c.most_common(28)))
