#!/bin/bash
# Same-box A/B of an environment switch: default bench, alternating arms; then one rocprofv3 --stats pass per arm.
#   bash tools/ab_env.sh <tag> "<VAR=value>" [rounds]
TAG=$1; ENVB=$2; R=${3:-3}
mkdir -p gpurun_out; export TMPDIR=/tmp
for i in $(seq 1 $R); do
  python3 bench.py --no-cpu-baseline --netb-pool 0 > gpurun_out/${TAG}_a$i.json 2>/dev/null
  env $ENVB python3 bench.py --no-cpu-baseline --netb-pool 0 > gpurun_out/${TAG}_b$i.json 2>/dev/null
  python3 - <<PY
import json
a=json.loads(open('gpurun_out/${TAG}_a$i.json').read().strip().splitlines()[-1]); b=json.loads(open('gpurun_out/${TAG}_b$i.json').read().strip().splitlines()[-1])
print('round $i  default %.1f (frac %.4f)   $ENVB %.1f (frac %.4f)'%(a['value'],a['roofline']['frac'],b['value'],b['roofline']['frac']), flush=True)
PY
done
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_statsA -o stats --output-format csv -- python3 bench.py --no-cpu-baseline --netb-pool 0 > gpurun_out/${TAG}_statsA.json 2>/dev/null
export $ENVB
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_statsB -o stats --output-format csv -- python3 bench.py --no-cpu-baseline --netb-pool 0 > gpurun_out/${TAG}_statsB.json 2>/dev/null
for X in A B; do echo "== arm $X"; python3 - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/${TAG}_stats$X/stats_kernel_stats.csv')))
tot=0
for r in rows:
    n=r['Name']
    if 'igemm4_kernel' in n or 'c3d_' in n:
        tot+=float(r['TotalDurationNs'])/int(r['Calls'])
    if 'c3d_' in n: print('  %-50s %8.1f us'%(n[:50], float(r['AverageNs'])/1e3))
print('  sum of the 12 contraction launches per pass: %.3f ms'%(tot/1e6))
PY
done
