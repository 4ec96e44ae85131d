#!/usr/bin/env python3
"""Recomputes bench.py's `roofline` figures from a rocprofv3 `--kernel-trace --stats` run of the SAME command.

    python tools/roofline_from_stats.py <stats_kernel_stats.csv> <bench JSON line of that run> [pmc_traffic.json] > summary.json

Definitions (the same ones bench.py prints):
  * dominant kernel = the twelve contraction launches of a pass: every instantiation of `alq::igemm4_kernel` (conv /
    conv_transpose forward + backward-data) and, since round 4, `alq::c3d_fwd_kernel` / `alq::c3d_bwd_kernel` (the conv under
    the head on the plane-sweep engine), since round 5 `alq::t3d_fwd_kernel` / `t3d_bwd_kernel` / `t3d8_fwd_kernel` (the
    conv_transpose layers on the row-sweep engine) `e3d_bwd_kernel` (enc2's backward fused with both pool backwards) and `d3d_fwd_kernel` / `d3d_bwd_kernel` (dec1's pair) - the keys of the output keep the historical `igemm4_` prefix;
  * executed 16-bit MFMA flops of a pass = the launches' ALGORITHMIC fp32 flops (2 x MACs of the real taps / channels,
    the library's own count, `roofline.igemm4_alg_flops_per_patch` of the bench line) x the 16-bit products issued per
    fp32-accurate MAC: 6 for the bf16x3 launches, 3 for the fp16x2 ones;
  * frac = executed flops / (total igemm4 time of the trace) / 2.5 PFLOP/s dense 16-bit MFMA peak
    (MI355X_MICROARCH.md).  Algebraically this IS achieved_algorithmic / harmonic-mean(2500/6, 2500/3 by flop share),
    the `achieved` / `peak` pair of the bench line;
  * hbm_frac = PMC traffic per launch x launches / igemm4 time / 8 TB/s (traffic from separate --pmc passes,
    profiles/pmc_traffic.json);
  * patches processed in the trace = (warmup + steps) x pool of the bench line.
"""
import csv
import json
import sys

PEAK_16BIT_TFLOPS = 2500.0
PEAK_HBM_TBPS = 8.0


def main():
    stats_csv, bench_json = sys.argv[1], sys.argv[2]
    line = json.loads(open(bench_json).read().strip().splitlines()[-1])
    rf = line['roofline']
    per_patch = rf['igemm4_alg_flops_per_patch']
    patches = (line['warmup'] + line['steps']) * line['config']['pool_per_gpu']
    ig_ns, ig_calls, total_ns = 0.0, 0, 0.0
    variants = []
    for r in csv.DictReader(open(stats_csv)):
        total_ns += float(r['TotalDurationNs'])
        kname = next((k for k in ('igemm4_kernel', 'c3d_fwd_kernel', 'c3d_bwd7_kernel', 'c3d_bwd_kernel', 't3d8_fwd_kernel', 't3d_fwd_kernel', 't3d_bwd_kernel', 'e3d_bwd_kernel', 'd3d_fwd_kernel', 'd3d_bwd_kernel', 't3d8_bwd_kernel', 'f3d_fwd_kernel') if k in r['Name']), None)
        if kname:
            ig_ns += float(r['TotalDurationNs'])
            ig_calls += int(r['Calls'])
            variants.append({'variant': (kname if kname != 'igemm4_kernel' else '') + r['Name'].split(kname)[1].split('(')[0], 'calls': int(r['Calls']),
                             'avg_us': float(r['AverageNs']) / 1e3, 'total_ms': float(r['TotalDurationNs']) / 1e6})
    alg = (per_patch['bf16x3'] + per_patch['f16x2']) * patches
    executed = (6 * per_patch['bf16x3'] + 3 * per_patch['f16x2']) * patches
    secs = ig_ns * 1e-9
    out = {
        'source': {'stats_csv': stats_csv, 'bench_line': bench_json},
        'patches_in_trace': patches, 'igemm4_launches': ig_calls, 'igemm4_total_ms': ig_ns / 1e6,
        'igemm4_avg_launch_ms': ig_ns / 1e6 / max(ig_calls, 1),
        # per device pass as the run cut them (config.patches_per_pass; older lines: config.batch)
        'patches_per_pass': line['config'].get('patches_per_pass', line['config']['batch']),
        'igemm4_ms_per_batch_pass': ig_ns / 1e6 / (patches / float(line['config'].get('patches_per_pass', line['config']['batch']))),
        'igemm4_alg_flops_per_patch': per_patch,
        'algorithmic_tflops': alg / secs / 1e12, 'executed_16bit_tflops': executed / secs / 1e12,
        'frac': executed / secs / 1e12 / PEAK_16BIT_TFLOPS,
        'bench_line': {'frac': rf['frac'], 'avg_launch_ms': rf['avg_launch_ms'], 'executed_16bit_tflops': rf['executed_16bit_tflops'],
                       'value': line['value'], 'note': rf.get('timed', 'HIP events of every %s-th pass inside the SAME run' % rf.get('timed_every_kth_pass'))},
        'igemm4_share_of_all_kernel_time': ig_ns / total_ns,
        'variants': sorted(variants, key=lambda v: -v['total_ms']),
    }
    out['frac_rel_diff_bench_vs_trace'] = rf['frac'] / out['frac'] - 1.0
    if len(sys.argv) > 3:
        tj = json.load(open(sys.argv[3]))
        bytes_per_launch = tj['hbm_bytes_per_launch'] * out['patches_per_pass'] / float(tj.get('batch', line['config']['batch']))
        out['hbm'] = {'traffic_bytes_per_launch': bytes_per_launch, 'hbm_frac': bytes_per_launch * ig_calls / secs / (PEAK_HBM_TBPS * 1e12),
                      'traffic_bytes_per_patch_all_igemm4': bytes_per_launch * ig_calls / patches, 'source': sys.argv[3]}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == '__main__':
    main()
