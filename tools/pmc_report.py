#!/usr/bin/env python3
"""Summary of the rocprofv3 passes made by tools/run_pmc.sh, per kernel (igemm4 variants separately).

    python tools/pmc_report.py gpurun_out/<tag> profiles/<tag>_pmc_summary.json

Reads <tag>_{stats,mfma,lds,fetch,write}/ and writes one JSON with, per kernel: calls, average duration (kernel trace of
the stats pass), matrix-pipe busy fraction (SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES)), MFMA ops by
operand type, LDS activity / bank conflicts / LDS issue stalls, and HBM traffic per launch (FETCH_SIZE x 2 + WRITE_SIZE,
KiB -> bytes: the gfx950 corrections of MI355X_MICROARCH.md, HBM section).  Counters are summed over the XCDs / SEs a
dispatch ran on (rocprofv3 emits one row per dimension instance)."""
import collections
import csv
import json
import os
import sys


def short(name):
    n = name.replace('alq::', '').replace('void ', '')
    return n.split('(')[0]


def per_kernel_counters(path):
    """kernel -> counter -> [per-dispatch totals]"""
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        d = disp.setdefault(r['Dispatch_Id'], {'name': short(r['Kernel_Name'])})
        d[r['Counter_Name']] = d.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
    out = collections.OrderedDict()
    for d in disp.values():
        k = out.setdefault(d['name'], collections.defaultdict(list))
        for c, v in d.items():
            if c != 'name':
                k[c].append(v)
    return out


# the twelve contraction launches of a NET-C pass; PMC_KERNELS="a,b,..." overrides (NET-B: igemm4_kernel,igemm3_kernel,igemm_kernel,fcgemm_kernel)
CONTRACTION = tuple(os.environ['PMC_KERNELS'].split(',')) if os.environ.get('PMC_KERNELS') else (
    'igemm4_kernel', 'c3d_fwd_kernel', 'c3d_bwd7_kernel', 'c3d_bwd_kernel', 't3d_fwd_kernel', 't3d_bwd_kernel', 't3d8_fwd_kernel', 'e3d_bwd_kernel',
    'd3d_fwd_kernel', 'd3d_bwd_kernel', 't3d8_bwd_kernel', 'f3d_fwd_kernel')


def mean(v):
    return sum(v) / len(v) if v else 0.0


def main():
    base, dst = sys.argv[1], sys.argv[2]
    stats = collections.OrderedDict()
    p = base + '_stats/stats_kernel_stats.csv'
    total_ns = 0.0
    for r in csv.DictReader(open(p)):
        stats[short(r['Name'])] = dict(calls=int(r['Calls']), avg_us=float(r['AverageNs']) / 1e3, pct=float(r['Percentage']))
        total_ns += float(r['TotalDurationNs'])
    mf = per_kernel_counters(base + '_mfma/mfma_counter_collection.csv')
    ld = per_kernel_counters(base + '_lds/lds_counter_collection.csv')
    fe = per_kernel_counters(base + '_fetch/fetch_counter_collection.csv')
    wr = per_kernel_counters(base + '_write/write_counter_collection.csv')
    out = collections.OrderedDict()
    out['source'] = 'tools/run_pmc.sh passes under %s_*; bench.py --pool 8192 --steps 1 --warmup 1' % os.path.basename(base)
    out['notes'] = ('mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES); lds_active = SQ_LDS_IDX_ACTIVE / SQ_BUSY_CU_CYCLES; '
                    'hbm bytes = 2 * FETCH_SIZE KiB + WRITE_SIZE KiB (gfx950 corrections); counters summed over a dispatch')
    kern = collections.OrderedDict()
    ig = dict(ms=0.0, n=0, rd=0.0, wr=0.0)
    for name, st in stats.items():
        if st['pct'] < 0.1:
            continue
        e = collections.OrderedDict(calls=st['calls'], avg_us=round(st['avg_us'], 1), pct_of_gpu_time=st['pct'])
        m = mf.get(name, {})
        busy_cu = mean(m.get('SQ_BUSY_CU_CYCLES', []))
        if busy_cu > 0:
            e['mfma_busy'] = round(mean(m.get('SQ_VALU_MFMA_BUSY_CYCLES', [])) / (4.0 * busy_cu), 4)
            e['mfma_mops_bf16'] = mean(m.get('SQ_INSTS_VALU_MFMA_MOPS_BF', []))
            e['mfma_mops_f16'] = mean(m.get('SQ_INSTS_VALU_MFMA_MOPS_F', []))
            e['insts_mfma'] = mean(m.get('SQ_INSTS_MFMA', []))
            e['insts_valu'] = mean(m.get('SQ_INSTS_VALU', []))
        l = ld.get(name, {})
        act = mean(l.get('SQ_LDS_IDX_ACTIVE', []))
        if act > 0:
            # SQ_BUSY_CU_CYCLES is in the other pass: normalise by the wave-cycle buckets of this one
            tot = mean(l.get('SQ_WAIT_ANY', [])) + mean(l.get('SQ_WAIT_INST_ANY', [])) + mean(l.get('SQ_ACTIVE_INST_ANY', []))
            e['lds_bank_conflict_frac_of_active'] = round(mean(l.get('SQ_LDS_BANK_CONFLICT', [])) / act, 4)
            e['lds_idx_active'] = act
            if busy_cu > 0:
                e['lds_active_frac_of_cu_busy'] = round(act / busy_cu, 4)
            if tot > 0:
                e['wave_cycles_waiting_frac'] = round(mean(l.get('SQ_WAIT_ANY', [])) / tot, 4)
                e['wave_cycles_issue_stalled_frac'] = round(mean(l.get('SQ_WAIT_INST_ANY', [])) / tot, 4)
                e['wave_cycles_lds_issue_stalled_frac'] = round(mean(l.get('SQ_WAIT_INST_LDS', [])) / tot, 4)
        f, w = fe.get(name, {}), wr.get(name, {})
        if f or w:
            rd_b = 2.0 * 1024.0 * mean(f.get('FETCH_SIZE', []))
            wr_b = 1024.0 * mean(w.get('WRITE_SIZE', []))
            e['hbm_read_MB_per_launch'] = round(rd_b / 1e6, 1)
            e['hbm_write_MB_per_launch'] = round(wr_b / 1e6, 1)
            e['hbm_GBps'] = round((rd_b + wr_b) / (st['avg_us'] * 1e-6) / 1e9, 0)
            if name.startswith(CONTRACTION):      # the contraction launches of a pass
                ig['ms'] += st['avg_us'] * st['calls'] / 1e3
                ig['n'] += st['calls']
                ig['rd'] += rd_b * st['calls']
                ig['wr'] += wr_b * st['calls']
        kern[name] = e
    out['kernels'] = kern
    if ig['n']:
        out['igemm4_all'] = dict(launches=ig['n'], avg_launch_ms=ig['ms'] / ig['n'],
                                 read_bytes_per_launch=ig['rd'] / ig['n'], write_bytes_per_launch=ig['wr'] / ig['n'],
                                 hbm_bytes_per_launch=(ig['rd'] + ig['wr']) / ig['n'])
    if ig['n'] and len(sys.argv) > 3:      # the traffic file bench.py reads (profiles/pmc_traffic.json)
        batch = int(sys.argv[4]) if len(sys.argv) > 4 else 2000
        tj = dict(kernel=os.environ.get('PMC_KERNEL_NOTE', 'the 12 contraction launches of a pass (c3d_fwd / c3d_bwd7 + t3d_fwd / t3d_bwd + t3d8_fwd / t3d8_bwd + d3d_fwd / d3d_bwd + f3d_fwd + e3d_bwd + two igemm4_kernel launches), '
                         'bench.py --lanes 1 --pool 8188 --batch %d --steps 1 --warmup 1' % batch),
                  launches=ig['n'], read_bytes_per_launch=ig['rd'] / ig['n'], write_bytes_per_launch=ig['wr'] / ig['n'],
                  hbm_bytes_per_launch=(ig['rd'] + ig['wr']) / ig['n'],
                  corrections='KiB -> bytes; FETCH_SIZE x2 (gfx950 wide reads); source %s (tools/run_pmc.sh)' % dst, batch=batch,
                  hbm_bytes_per_patch_all_contraction_launches=(ig['rd'] + ig['wr']) / ig['n'] * float(os.environ.get('PMC_LAUNCHES_PER_PASS', '12')) / batch)
        json.dump(tj, open(sys.argv[3], 'w'), indent=1)
    json.dump(out, open(dst, 'w'), indent=1)
    print(json.dumps(out.get('igemm4_all', {}), indent=1))
    for k, e in kern.items():
        print('%-62s %4d x %7.1f us  mfma %5.1f%%  lds %5.1f%%  hbm %7.1f MB' % (
            k[:62], e['calls'], e['avg_us'], 100 * e.get('mfma_busy', 0), 100 * e.get('lds_active_frac_of_cu_busy', 0),
            e.get('hbm_read_MB_per_launch', 0) + e.get('hbm_write_MB_per_launch', 0)))


if __name__ == '__main__':
    main()
