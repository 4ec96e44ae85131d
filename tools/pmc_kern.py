#!/usr/bin/env python3
"""Per-kernel table (duration, matrix-pipe busy, instruction mix, wait shares) from the passes of tools/prof_c3d.sh.

    python tools/pmc_kern.py <stats dir> <pmcA dir> <pmcB dir>
"""
import collections
import csv
import glob
import os
import sys


def find(d, pat):
    hits = glob.glob(os.path.join(d, '**', pat), recursive=True)
    return hits[0] if hits else None


def short(n):
    n = n.replace('alq::', '').replace('void ', '')
    if '(' in n:
        n = n.split('(')[0]
    return n[:70]


def pmc(d):
    f = find(d, '*counter_collection.csv')
    acc = collections.OrderedDict()
    if not f:
        return acc
    disp = {}
    for r in csv.DictReader(open(f)):
        k = r['Dispatch_Id']
        e = disp.setdefault(k, {'name': short(r['Kernel_Name'])})
        e[r['Counter_Name']] = e.get(r['Counter_Name'], 0) + float(r['Counter_Value'])
    for e in disp.values():
        a = acc.setdefault(e['name'], collections.Counter())
        a['n'] += 1
        for k, v in e.items():
            if k != 'name':
                a[k] += v
    return acc


stats = find(sys.argv[1], '*kernel_stats.csv')
A, B = pmc(sys.argv[2]), pmc(sys.argv[3])
rows = list(csv.DictReader(open(stats)))
print('%-70s %6s %9s %6s | %5s %5s %5s %5s | %5s %5s %5s | %5s' % ('kernel', 'calls', 'avg us', '%', 'mfma%', 'V/M', 'S/M', 'L/M', 'wait%', 'stall%', 'act%', 'ldsC%'))
for r in rows[:22]:
    n = short(r['Name'])
    a, b = A.get(n), B.get(n)
    line = '%-70s %6s %9.1f %6.2f |' % (n, r['Calls'], float(r['AverageNs']) / 1e3, float(r['Percentage']))
    if a and a['SQ_INSTS_MFMA'] > 0:
        cyc = a['GRBM_GUI_ACTIVE'] / 8.0        # summed over the 8 XCDs
        mf = a['SQ_INSTS_MFMA']
        line += ' %5.1f %5.2f %5.2f %5.2f |' % (100.0 * a['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024.0), (a['SQ_INSTS_VALU'] - mf) / mf, a['SQ_INSTS_SALU'] / mf, a['SQ_INSTS_LDS'] / mf)
    else:
        line += ' %5s %5s %5s %5s |' % ('-', '-', '-', '-')
    if b and b['SQ_WAVE_CYCLES'] > 0:
        wc = b['SQ_WAVE_CYCLES']
        line += ' %5.1f %5.1f %5.1f | %5.1f' % (100.0 * b['SQ_WAIT_ANY'] / wc, 100.0 * b['SQ_WAIT_INST_ANY'] / wc, 100.0 * b['SQ_ACTIVE_INST_ANY'] / wc,
                                              100.0 * b['SQ_LDS_BANK_CONFLICT'] / max(b['SQ_LDS_IDX_ACTIVE'], 1))
    print(line)
