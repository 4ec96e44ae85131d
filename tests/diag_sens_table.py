"""Median per-call kernel durations of the builds timed by tests/diag_sens.sh (diagnostic)."""
import collections
import csv
import sys

libs = sys.argv[1:]
res = collections.OrderedDict()
for lib in libs:
    d = collections.defaultdict(list)
    for r in csv.DictReader(open('gpurun_out/sens_%s/s_kernel_trace.csv' % lib)):
        n = r['Kernel_Name']
        k = n.split('igemm4_kernel')[1].split('(')[0] if 'igemm4' in n else n.split('(')[0].replace('void ', '').replace('alq::', '')[:44]
        d[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    for k, v in d.items():
        v.sort()
        res.setdefault(k, {})[lib] = (v[len(v) // 2], len(v), sum(v))
print('%-46s' % 'kernel: median us (calls)', *['%14s' % l[:-3] for l in libs])
tot = collections.defaultdict(float)
for k, v in res.items():
    if max(x[2] for x in v.values()) < 200:
        continue
    print('%-46s' % k, *['%9.0f(%3d)' % (v[l][0], v[l][1]) if l in v else '%14s' % '-' for l in libs])
for k, v in res.items():
    for l in v:
        tot[l] += v[l][2]
print('%-46s' % 'all kernels, ms', *['%14.2f' % (tot[l] / 1e3) for l in libs])
