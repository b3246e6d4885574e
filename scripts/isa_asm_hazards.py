#!/usr/bin/env python
"""ISA audit of the pairs hipcc does NOT pad: hazards whose producer or consumer sits inside an inline-asm statement.

hipcc's hazard recogniser (GCNHazardRecognizer) treats an INLINEASM as one opaque instruction and returns early for it: none of
the MFMA <-> VALU wait-state rules are applied to the instructions inside the string, in either direction.  This script walks the
final assembly of one kernel in layout order and lists, with the wait states between them (one per instruction, N + 1 per s_nop N,
the way hipcc counts):

  A  a VGPR written inside an asm block and read by a later v_mfma as A / B / C operand           (VALU write -> MFMA read)
  B  a VGPR written by a v_mfma (its D) and read inside an asm block                               (MFMA write -> VALU read)
  C  a VGPR written inside an asm block that is the D or C operand of an earlier v_mfma            (MFMA write / read C -> VALU write)
  D  a VGPR written inside an asm block that is the A / B operand of an earlier v_mfma             (MFMA read A/B -> VALU write)

Only pairs closer than WINDOW states are listed; branches end a scan (straight-line code only), labels do not.
--mfma-states: an intervening MFMA counts as its issue interval in quad-cycles (8 for the 32x32x16 16-bit forms, 16 for 32x32x2 fp32)
instead of hipcc's one state — the matrix pipe accepts the next MFMA of a SIMD only that much later, whichever wave it is from.
(scripts/repro_asm_waw_hazard.hip on the MI355X: a VALU write into the destination tuple of a v_mfma_f32_32x32x16_f16 is lost up
to 4 wait states behind it and safe from 6 on; hipcc itself pads such pairs, when it sees them, with 12.)
Usage: isa_asm_hazards.py file.s kernel_symbol_prefix [WINDOW] [--all] [--mfma-states]
tests/test_isa_audit.py runs audit() over every kernel of the library as part of the CPU suite."""
import re
import sys
from collections import Counter


def regs(tok):
    out = set()   # VGPR n -> n, AGPR n -> 1000 + n
    for m in re.finditer(r'(?<![\w.])([va])\[(\d+):(\d+)\]|(?<![\w.\[])([va])(\d+)\b', tok):
        if m.group(1) is not None:
            base = 1000 if m.group(1) == 'a' else 0
            out.update(range(base + int(m.group(2)), base + int(m.group(3)) + 1))
        else:
            out.add((1000 if m.group(4) == 'a' else 0) + int(m.group(5)))
    return out


NO_VDST = ('v_cmp', 'v_nop', 'ds_write', 'ds_store', 'global_store', 'buffer_store', 'scratch_store', 'global_atomic', 'flat_store',
           's_', 'v_readfirstlane', 'v_readlane', 'buffer_wbl2', 'buffer_inv', 'ds_add', 'ds_or')


def split_ops(body):
    return [o.strip() for o in body.split(',')]


def defs_uses(name, body):
    ops = split_ops(body) if body else []
    if not ops:
        return set(), set()
    if name.startswith(NO_VDST) and not name.endswith('_rtn') and '_rtn_' not in name:
        return set(), regs(body)
    d = regs(ops[0])
    u = regs(','.join(ops[1:]))
    if name.startswith(('v_fmac', 'v_mac', 'v_dot2c', 'v_pk_fmac')):
        u |= d
    return d, u


def parse(path, sym):
    s = open(path).read()
    # the whole function, up to its .Lfunc_end label (round 6: a kernel whose early-exit block is laid out first has an s_endpgm long
    # before its body — stopping at the first one audited 53 instructions of the per-point kernels); hand-written test streams
    # without that label end at their last s_endpgm
    m = re.search(r'^(' + re.escape(sym) + r'[^\n:]*):[^\n]*\n(.*?)\n\.Lfunc_end\d+:', s, re.S | re.M)
    if m is None:
        m = re.search(r'^(' + re.escape(sym) + r'[^\n:]*):[^\n]*\n(.*)s_endpgm', s, re.S | re.M)
    if m is None:
        raise SystemExit(f"kernel {sym} not found in {path}")
    out = []
    in_asm = False
    asm_id = -1
    for ln, raw in enumerate(m.group(2).split('\n')):
        if ';;#ASMSTART' in raw:
            in_asm = True
            asm_id += 1
            continue
        if ';;#ASMEND' in raw:
            in_asm = False
            continue
        l = raw.split(';')[0].strip()
        if not l or l.startswith('.') and not l.endswith(':'):
            continue
        if l.endswith(':'):
            out.append(dict(kind='label', text=l, ln=ln))
            continue
        parts = l.split(None, 1)
        name = parts[0]
        body = parts[1] if len(parts) > 1 else ''
        d, u = defs_uses(name, body)
        ins = dict(kind='ins', name=name, body=body, text=l, ln=ln, asm=asm_id if in_asm else -1, d=d, u=u)
        if name.startswith('v_mfma'):
            ops = split_ops(body)
            ins['mD'] = regs(ops[0]); ins['mA'] = regs(ops[1]); ins['mB'] = regs(ops[2]); ins['mC'] = regs(ops[3])
        out.append(ins)
    return out


def states_of(ins, mfma_states=False):
    if ins['name'] == 's_nop':
        return int(ins['body'].split()[0]) + 1
    if mfma_states and ins['name'].startswith('v_mfma'):
        return 16 if '32x32x2' in ins['name'] and 'x16' not in ins['name'] else 8
    return 1


def audit(path, sym, window=20, mfma_states=False):
    """-> (found, examples): per class 'A'..'D' a Counter {wait states: pairs} and one example per distance"""
    prog = parse(path, sym)
    found = {k: Counter() for k in 'ABCD'}
    examples = {k: {} for k in 'ABCD'}
    n = len(prog)
    for i, a in enumerate(prog):
        if a['kind'] != 'ins':
            continue
        is_mfma = a['name'].startswith('v_mfma')
        in_asm = a['asm'] >= 0
        if not (is_mfma or (in_asm and a['d'])):
            continue
        # scan forward from a (producer / earlier instruction)
        st = 0
        pend_A = set(a['d']) if (in_asm and not is_mfma) else set()
        pend_B = set(a['mD']) if is_mfma else set()
        pend_C = (set(a['mD']) | set(a['mC'])) if is_mfma else set()
        pend_D = (set(a['mA']) | set(a['mB'])) if is_mfma else set()
        for j in range(i + 1, n):
            b = prog[j]
            if b['kind'] == 'label':
                continue
            if st >= window:
                break
            nm = b['name']
            if nm == 's_endpgm' or nm == 's_branch' or nm == 's_setpc_b64':
                break
            b_mfma = nm.startswith('v_mfma')

            def rec(kind, hit):
                key = st
                found[kind][key] += 1
                if key not in examples[kind]:
                    examples[kind][key] = (a['text'][:80], b['text'][:80], sorted(hit)[:4], a['ln'], b['ln'])

            if pend_A and b_mfma:
                hit = pend_A & (b['mA'] | b['mB'] | b['mC'])
                if hit:
                    rec('A', hit)
                    pend_A -= hit
            if b['asm'] >= 0 and not b_mfma:
                if pend_B:
                    hit = pend_B & b['u']
                    if hit:
                        rec('B', hit)
                        pend_B -= hit
                if pend_C:
                    hit = pend_C & b['d']
                    if hit:
                        rec('C', hit)
                        pend_C -= hit
                if pend_D:
                    hit = pend_D & b['d']
                    if hit:
                        rec('D', hit)
                        pend_D -= hit
            # a register rewritten by a later instruction is no longer the producer's value
            pend_A -= b['d']
            pend_B -= b['d']
            st += states_of(b, mfma_states)   # (a conditional branch: the fall-through path only)
    return found, examples


def main():
    show_all = '--all' in sys.argv
    mfma_states = '--mfma-states' in sys.argv
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    path, sym = args[0], args[1]
    window = int(args[2]) if len(args) > 2 else 20
    found, examples = audit(path, sym, window, mfma_states)
    title = {'A': 'asm VALU write -> MFMA reads it as A/B/C', 'B': 'MFMA D -> asm instruction reads it',
             'C': 'MFMA D or C -> asm instruction overwrites it', 'D': 'MFMA A/B -> asm instruction overwrites it'}
    for k in 'ABCD':
        tot = sum(found[k].values())
        print(f"[{k}] {title[k]}: {tot} pairs within {window} states")
        for st in sorted(found[k]):
            ex = examples[k][st]
            print(f"     {found[k][st]:4d} x after {st:2d} states   e.g. line {ex[3]}: {ex[0]}  ->  line {ex[4]}: {ex[1]}  (v{ex[2]})")
            if not show_all and st > 12:
                break


if __name__ == "__main__":
    main()
