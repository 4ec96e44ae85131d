#!/usr/bin/env python3
"""Static instruction mix of the igemm4 instantiations from the device assembly build.sh leaves behind.

    python tools/isa_loop_stats.py [asm] [substring of the mangled kernel name ...]

Per kernel: instruction counts by class over the WHOLE body and between the first and the last s_barrier (the tick loop
and its prologue), SGPR-spill traffic (v_readlane / v_writelane), and the VALU : MFMA ratio the verdict tracks."""
import collections
import re
import sys

ASM = 'nn-active-learning_amd/csrc/build/igemm4-hip-amdgcn-amd-amdhsa-gfx950.s'


def classify(op):
    if op.startswith('v_mfma'):
        return 'mfma'
    if op.startswith(('v_readlane', 'v_writelane', 'v_readfirstlane')):
        return 'lane'
    if op.startswith('v_mov') or op.startswith('v_accvgpr'):
        return 'vmov'
    if op.startswith('v_'):
        return 'valu'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith(('buffer_', 'global_', 'flat_', 'scratch_')):
        return 'vmem'
    if op.startswith('s_waitcnt'):
        return 'wait'
    if op.startswith('s_barrier'):
        return 'barrier'
    if op.startswith('s_load') or op.startswith('s_buffer'):
        return 'smem'
    if op.startswith('s_'):
        return 'salu'
    return 'other'


def main():
    args = sys.argv[1:]
    asm = ASM
    if args and args[0].endswith('.s'):
        asm = args.pop(0)
    pats = args
    cur, body = None, collections.OrderedDict()
    for line in open(asm):
        m = re.match(r'^(_ZN3alq13igemm4_kernel\S*):', line)
        if m:
            cur = m.group(1)
            body[cur] = []
            continue
        if cur is None:
            continue
        if line.startswith('.Lfunc_end') or line.startswith('\t.section'):
            cur = None
            continue
        s = line.strip()
        if not s or s.startswith((';', '.', '//')) or s.endswith(':'):
            continue
        body[cur].append(s.split()[0])
    for name, ops in body.items():
        if pats and not any(p in name for p in pats):
            continue
        bar = [i for i, o in enumerate(ops) if o.startswith('s_barrier')]
        inner = ops[bar[0]:bar[-1]] if len(bar) >= 2 else ops
        tot, inn = collections.Counter(map(classify, ops)), collections.Counter(map(classify, inner))
        vm = (inn['valu'] + inn['vmov'] + inn['lane']) / max(inn['mfma'], 1)
        print(name[len('_ZN3alq13igemm4_kernel'):-len('EEvNS_10Igemm4ArgsE')])
        print('   whole: %s' % dict(tot))
        print('   between first and last barrier: %s   (VALU+mov+lane)/MFMA = %.2f, plain VALU/MFMA = %.2f' %
              (dict(inn), vm, inn['valu'] / max(inn['mfma'], 1)))


if __name__ == '__main__':
    main()
