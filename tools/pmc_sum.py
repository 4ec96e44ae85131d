"""Sums rocprofv3 --pmc counter CSVs per kernel name (diagnostic).  usage: python tools/pmc_sum.py <substring> <dir>..."""
import csv, glob, sys, collections
sub = sys.argv[1]
for d in sys.argv[2:]:
    for f in sorted(glob.glob(d + '/**/*counter_collection.csv', recursive=True)):
        acc = collections.defaultdict(float); ndisp = set()
        for r in csv.DictReader(open(f)):
            if sub in r['Kernel_Name']:
                acc[r['Counter_Name']] += float(r['Counter_Value']); ndisp.add(r['Dispatch_Id'])
        print(f); print('  dispatches', len(ndisp))
        for k, v in sorted(acc.items()): print('  %-28s %.4g  (per dispatch %.4g)' % (k, v, v / max(1, len(ndisp))))
