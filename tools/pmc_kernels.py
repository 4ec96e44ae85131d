#!/usr/bin/env python3
"""Per-dispatch PMC table from a rocprofv3 --pmc counter_collection.csv (largest dispatches of a kernel).

    python tools/pmc_kernels.py <counter_collection.csv> [kernel substring] [min us]
"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
key = sys.argv[2] if len(sys.argv) > 2 else 'igemm'
disp = collections.OrderedDict()
for r in rows:
    if key not in r['Kernel_Name']:
        continue
    d = disp.setdefault(r['Dispatch_Id'], {'name': r['Kernel_Name'].replace('alq::', '').replace('void ', '').split('(')[0]})
    d[r['Counter_Name']] = d.get(r['Counter_Name'], 0) + float(r['Counter_Value'])
items = list(disp.values())
n = len(items)
names = sorted({k for d in items for k in d if k != 'name'})
print('%-32s' % 'kernel', ' '.join('%14s' % c[-14:] for c in names))
for d in items[n // 2: n // 2 + int(sys.argv[3]) if len(sys.argv) > 3 else n]:
    print('%-32s' % d['name'][:32], ' '.join('%14.0f' % d.get(c, 0) for c in names))
