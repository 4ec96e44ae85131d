"""Median duration (us) of each igemm4 launch of a Fisher pass, by its ordinal in the pass, per tools/tune_sens.sh run."""
import collections
import csv
import sys

settings = sys.argv[1:]
cols = []
for i, _ in enumerate(settings):
    rows = list(csv.DictReader(open('gpurun_out/tune_%d/s_kernel_trace.csv' % i)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    per = collections.defaultdict(list)
    k, total = 0, []
    acc = 0.0
    for r in rows:
        n = r['Kernel_Name']
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        if 'direct_conv_pool' in n:
            k, acc = 0, 0.0
        if 'igemm4' in n:
            per[k].append((d, n.split('igemm4_kernel')[1].split('(')[0]))
            k += 1
            acc += d
        if 'fisher_finalize' in n:
            total.append(acc)
    cols.append((per, total))
print('%-3s %-52s' % ('#', 'variant of the first run'), *['%10s' % ('run%d' % i) for i in range(len(settings))])
for k in sorted(cols[0][0]):
    name = cols[0][0][k][0][1]
    vals = []
    for per, _ in cols:
        v = sorted(x[0] for x in per.get(k, [(0, '')]))
        vals.append(v[len(v) // 2])
    print('%-3d %-52s' % (k, name), *['%10.0f' % v for v in vals])
print('%-56s' % 'sum of igemm4 per pass (median)', *['%10.0f' % sorted(t)[len(t) // 2] for _, t in cols])
for i, s in enumerate(settings):
    print('run%d: %s' % (i, s))
