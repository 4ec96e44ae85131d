#!/usr/bin/env python3
"""Build gate for the 16-byte-store data hazard (HISTORY.md 13).

Found in round 5: a `buffer_store_dwordx4` whose scalar offset is an SGPR can store what a LATER vector instruction wrote into
its data registers (seen once a second wave shared the SIMD).  LLVM's hazard recogniser (GCNHazardRecognizer::createsVALUHazard,
the ">64-bit store data" entry: 2 wait states on gfx940+) inserts wait states only when the store has NO register soffset, so
for this form nothing is inserted.  The sources keep the data registers alive behind such stores (`ALQ_STORE_B128_SOFF` /
explicit `s_nop` + "v" inputs); this script checks the RESULT in the device assembly: for every buffer_store_dwordx3/x4 with an
SGPR soffset it walks the instructions behind it (both arms of a branch) and counts wait states - one per instruction,
N + 1 for `s_nop N` - up to the first VALU / MFMA / DS-read-free instruction that WRITES one of the store's data VGPRs.

  isa_store_hazard.py [--min W] file.s ...     exit 1 when a store has fewer than W wait states (default 2 =
                                               ALQ_STORE_HOLD_STATES: the compiler's figure for the sibling hazard)
  isa_store_hazard.py --report file.s ...      histogram only
"""
import re
import sys

VREG = re.compile(r'\bv(\d+)\b|\bv\[(\d+):(\d+)\]')
STORE = re.compile(r'^\s*buffer_store_dwordx([34])\s+(v\[\d+:\d+\]),\s*(\S+),\s*(s\[\d+:\d+\]),\s*(\S+)')
LABEL = re.compile(r'^([.\w$]+):')
BR = re.compile(r'^\s*s_(c?branch\w*)\s+(\S+)')


def regs(tok):
    m = VREG.search(tok)
    if not m:
        return set()
    if m.group(1) is not None:
        return {int(m.group(1))}
    return set(range(int(m.group(2)), int(m.group(3)) + 1))


def written(op, args):
    """VGPRs an instruction writes at issue (VALU / MFMA results; loads write on return, long after)."""
    if not op.startswith('v_'):
        return set()
    if op.startswith(('v_cmp', 'v_cmpx', 'v_readlane', 'v_readfirstlane', 'v_nop')):
        return set()
    w = regs(args[0]) if args else set()
    if 'swap' in op and len(args) > 1:
        w |= regs(args[1])
    return w


def parse(path):
    ins, labels = [], {}
    for ln in open(path):
        s = ln.split(';')[0].rstrip()
        if not s.strip() or s.strip().startswith(('.', '//', '#')) and not LABEL.match(s.strip()):
            m = LABEL.match(s.strip()) if s.strip() else None
            if not m:
                continue
        m = LABEL.match(s.strip())
        if m:
            labels[m.group(1)] = len(ins)
            continue
        if not s.startswith(('\t', ' ')):
            continue
        parts = s.strip().split(None, 1)
        op = parts[0]
        if op.startswith('.'):
            continue
        args = [a.strip() for a in parts[1].split(',')] if len(parts) > 1 else []
        ins.append((op, args, s.strip()))
    return ins, labels


def same_value_rewrite(ins, store, w):
    """The writer ins[w] is a v_mov from scalar registers / a constant and the last writer of the same registers IN FRONT of the
    store is textually the same move with its sources untouched in between: the registers keep their value (the peeled void
    first pass of the sweep kernels re-materialises its zero rows like this) - not a hazard."""
    op, args, text = ins[w]
    if not op.startswith('v_mov_b') or len(args) != 2 or VREG.search(args[1]):
        return False
    dst = regs(args[0])
    src = set()
    m = re.match(r'^s\[(\d+):(\d+)\]$', args[1])
    if m:
        src = set(range(int(m.group(1)), int(m.group(2)) + 1))
    elif re.match(r'^s\d+$', args[1]):
        src = {int(args[1][1:])}
    for j in range(store - 1, max(store - 400, -1), -1):
        o2, a2, t2 = ins[j]
        if BR.match(t2) or o2 == 's_endpgm':
            return False
        if written(o2, a2) & dst:
            if t2 != text:
                return False
            # the scalar sources must not change between the two moves
            for k in range(j + 1, w):
                o3, a3, _ = ins[k]
                if o3.startswith('s_') and a3:
                    m3 = re.match(r'^s\[(\d+):(\d+)\]$', a3[0])
                    d3 = set(range(int(m3.group(1)), int(m3.group(2)) + 1)) if m3 else ({int(a3[0][1:])} if re.match(r'^s\d+$', a3[0]) else set())
                    if d3 & src:
                        return False
            return True
    return False


def distance(ins, labels, start, data, limit):
    """Fewest wait states from the store at `start` to a write of `data` over all paths (>= limit: limit)."""
    best = limit
    stack = [(start + 1, 0)]
    seen = {}
    while stack:
        i, ws = stack.pop()
        while i < len(ins) and ws < best:
            if seen.get(i, limit + 1) <= ws:
                break
            seen[i] = ws
            op, args, _ = ins[i]
            if written(op, args) & data:
                if not same_value_rewrite(ins, start, i):
                    best = min(best, ws)
                break
            if op == 's_endpgm':
                break
            m = BR.match(ins[i][2])
            if op == 's_nop':
                ws += int(args[0], 0) + 1
            else:
                ws += 1
            if m:
                tgt = labels.get(m.group(2))
                if tgt is not None:
                    stack.append((tgt, ws))
                if m.group(1) == 'branch':
                    break
            i += 1
    return best


def scan(path, limit=64):
    ins, labels = parse(path)
    out = []
    for i, (op, args, text) in enumerate(ins):
        m = STORE.match(text)
        if not m:
            continue
        soff = m.group(5)
        if not re.match(r'^s\d+$', soff):      # immediate / null soffset: the compiler's own hazard entry covers it
            continue
        out.append((distance(ins, labels, i, regs(m.group(2)), limit), text))
    return out


def main():
    argv = sys.argv[1:]
    report = '--report' in argv
    argv = [a for a in argv if a != '--report']
    wmin = 2
    if '--min' in argv:
        k = argv.index('--min')
        wmin = int(argv[k + 1])
        del argv[k:k + 2]
    bad = 0
    for path in argv:
        res = scan(path)
        hist = {}
        for d, _ in res:
            hist[d] = hist.get(d, 0) + 1
        lo = min(hist) if hist else None
        print('%s: %d 12/16-byte stores with a register soffset; fewest wait states before a data register is rewritten: %s  %s'
              % (path, len(res), lo, ' '.join('%d:%d' % kv for kv in sorted(hist.items())[:6])))
        if not report:
            for d, text in res:
                if d < wmin:
                    bad += 1
                    print('  HAZARD (%d wait states < %d): %s' % (d, wmin, text), file=sys.stderr)
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
