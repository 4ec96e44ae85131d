#!/usr/bin/env python3
"""Per-kernel table from tools/run_quick_prof.sh passes:  python tools/pmc_quick.py gpurun_out/<tag> [out.json]"""
import collections
import csv
import json
import sys

sys.path.insert(0, __import__('os').path.dirname(__file__))
from pmc_report import per_kernel_counters, short, mean  # noqa: E402


def main():
    base = sys.argv[1]
    stats = collections.OrderedDict()
    for r in csv.DictReader(open(base + '_stats/stats_kernel_stats.csv')):
        stats[short(r['Name'])] = dict(calls=int(r['Calls']), avg_us=float(r['AverageNs']) / 1e3, pct=float(r['Percentage']))
    mf = per_kernel_counters(base + '_mfma/mfma_counter_collection.csv')
    ld = per_kernel_counters(base + '_lds/lds_counter_collection.csv')
    out = collections.OrderedDict()
    for name, st in stats.items():
        if st['pct'] < 0.5:
            continue
        m, l = mf.get(name, {}), ld.get(name, {})
        busy = mean(m.get('SQ_BUSY_CU_CYCLES', []))
        e = dict(calls=st['calls'], avg_us=round(st['avg_us'], 1), pct=st['pct'])
        if busy > 0:
            e['mfma_busy'] = round(mean(m.get('SQ_VALU_MFMA_BUSY_CYCLES', [])) / (4.0 * busy), 3)
            nm = mean(m.get('SQ_INSTS_MFMA', []))
            e['valu_per_mfma'] = round(mean(m.get('SQ_INSTS_VALU', [])) / nm, 2) if nm else None
            e['salu_per_mfma'] = round(mean(m.get('SQ_INSTS_SALU', [])) / nm, 2) if nm else None
        act = mean(l.get('SQ_LDS_IDX_ACTIVE', []))
        if act > 0:
            e['lds_conflict_frac'] = round(mean(l.get('SQ_LDS_BANK_CONFLICT', [])) / act, 3)
            if busy > 0:
                e['lds_active_of_busy'] = round(act / busy, 3)
            tot = mean(l.get('SQ_WAIT_ANY', [])) + mean(l.get('SQ_WAIT_INST_ANY', [])) + mean(l.get('SQ_ACTIVE_INST_ANY', []))
            if tot > 0:
                e['waiting'] = round(mean(l.get('SQ_WAIT_ANY', [])) / tot, 3)
                e['issue_stalled'] = round(mean(l.get('SQ_WAIT_INST_ANY', [])) / tot, 3)
        out[name] = e
        print('%-75s %4d x %7.1f us %5.1f%%  mfma %5.1f%%  valu/mfma %5s salu/mfma %5s lds %5.1f%% conf %5.1f%% wait %4.1f%% stall %4.1f%%' % (
            name[:75], e['calls'], e['avg_us'], e['pct'], 100 * e.get('mfma_busy', 0), e.get('valu_per_mfma'), e.get('salu_per_mfma'),
            100 * e.get('lds_active_of_busy', 0), 100 * e.get('lds_conflict_frac', 0), 100 * e.get('waiting', 0), 100 * e.get('issue_stalled', 0)))
    if len(sys.argv) > 2:
        json.dump(out, open(sys.argv[2], 'w'), indent=1)


if __name__ == '__main__':
    main()
