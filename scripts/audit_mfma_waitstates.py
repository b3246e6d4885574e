#!/usr/bin/env python
"""ISA audit: between an MFMA and any later instruction that reads or writes its result registers there must be at least
REQ wait states (measured on the MI355X for v_mfma_f32_32x32x16_bf16 with scripts/mfma16_result_probe.hip: s_nop 11 = 12
states; the next MFMA taking the result whole as its own C operand needs none).  hipcc counts every instruction as one
state and s_nop N as N + 1 (GCNHazardRecognizer); this script counts the same way over the final assembly and lists every
pair that falls short.  Usage: audit_mfma_waitstates.py file.s kernel_symbol_prefix [REQ]"""
import re
import sys


def regs(tok):
    out = set()
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b', tok):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


LINEAR = "--linear" in sys.argv


def main():
    if LINEAR:
        sys.argv.remove("--linear")
    path, sym = sys.argv[1], sys.argv[2]
    req = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    s = open(path).read()
    m = re.search(r'^(' + re.escape(sym) + r'[^\n:]*):[^\n]*\n(.*?)s_endpgm', s, re.S | re.M)
    lines = []
    for l in m.group(2).split('\n'):
        l = l.split(';')[0].strip()
        if not l or l.startswith('.') or l.endswith(':'):
            if l.endswith(':'):
                lines.append(('label', l))
            continue
        lines.append(('ins', l))
    n_mfma = n_short = 0
    worst = {}
    for i, (kind, l) in enumerate(lines):
        if kind != 'ins' or not l.startswith('v_mfma'):
            continue
        n_mfma += 1
        ops = l.split(None, 1)[1].split(',')
        dst = regs(ops[0])
        states = 0
        for kind2, l2 in lines[i + 1:]:
            if kind2 == 'label':
                continue
            if states >= req:
                break
            name = l2.split()[0]
            if name == 's_nop':
                states += int(l2.split()[1]) + 1
                continue
            if name == 's_endpgm' or (name == 's_branch' and not LINEAR):
                break
            if name.startswith('s_cbranch') or name == 's_branch':
                if not LINEAR:
                    break  # end of the straight-line scan
                states += 1   # LINEAR: keep scanning the fall-through path (over-approximates the layout order)
                continue
            body = l2.split(None, 1)[1] if ' ' in l2 else ''
            touched = regs(body)
            if name.startswith('v_mfma'):
                o2 = body.split(',')
                if regs(o2[3]) == dst and regs(o2[0]) == dst and not (regs(o2[1]) & dst) and not (regs(o2[2]) & dst):
                    states += 1      # accumulate chain: allowed
                    continue
            if touched & dst:
                n_short += 1
                key = (name, states)
                worst[key] = worst.get(key, 0) + 1
                if n_short <= 8:
                    print(f"short: {l[:70]}  ->  after {states} states: {l2[:90]}")
                break
            states += 1
    print(f"{n_mfma} MFMAs audited, {n_short} consumers closer than {req} wait states")
    for (name, st), c in sorted(worst.items(), key=lambda kv: kv[0][1]):
        print(f"   {c:4d} x {name} after {st} states")


if __name__ == "__main__":
    main()
