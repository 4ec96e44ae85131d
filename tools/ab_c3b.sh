#!/bin/bash
# Same-box A/B of the head conv's backward kernel: 7 k-steps (default) vs 9 (ALQ_C3D_BWD_ROWS=8); then a kernel trace per arm.
mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r06f}
for i in 1 2 3; do
  python3 bench.py --no-cpu-baseline --netb-pool 0 > gpurun_out/${TAG}_a$i.json 2>/dev/null
  ALQ_C3D_BWD_ROWS=8 python3 bench.py --no-cpu-baseline --netb-pool 0 > gpurun_out/${TAG}_b$i.json 2>/dev/null
  python3 - <<PY
import json
a=json.loads(open('gpurun_out/${TAG}_a$i.json').read().strip().splitlines()[-1]); b=json.loads(open('gpurun_out/${TAG}_b$i.json').read().strip().splitlines()[-1])
print('round $i  7 k-steps %.1f (useful %.4f)   9 k-steps %.1f (useful %.4f)'%(a['value'],a['roofline']['useful_frac'],b['value'],b['roofline']['useful_frac']), flush=True)
PY
done
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_statsA -o stats --output-format csv -- python3 bench.py --lanes 1 --no-cpu-baseline --netb-pool 0 > gpurun_out/${TAG}_statsA.json 2>/dev/null
export ALQ_C3D_BWD_ROWS=8
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_statsB -o stats --output-format csv -- python3 bench.py --lanes 1 --no-cpu-baseline --netb-pool 0 > gpurun_out/${TAG}_statsB.json 2>/dev/null
for X in A B; do echo "== arm $X"; python3 - <<PY
import csv,glob
f=glob.glob('gpurun_out/${TAG}_stats$X/**/*kernel_stats.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in sorted(rows, key=lambda r:-float(r['TotalDurationNs']))[:16]:
    print('  %-60s %6s calls %9.1f us avg'%(r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
