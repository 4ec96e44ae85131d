#!/bin/bash
# Same-box A/B of two builds of libalq (ALQ_LIB selects the file next to the package): default bench, alternating arms.
#   bash tools/ab_lib.sh <tag> <other lib file> [rounds]
TAG=$1; LIBB=$2; R=${3:-3}
mkdir -p gpurun_out
for i in $(seq 1 $R); do
  python bench.py --no-cpu-baseline --netb-pool 0 > gpurun_out/${TAG}_a$i.json 2>/dev/null
  ALQ_LIB=$LIBB python bench.py --no-cpu-baseline --netb-pool 0 > gpurun_out/${TAG}_b$i.json 2>/dev/null
  python - <<PY
import json
a=json.loads(open('gpurun_out/${TAG}_a$i.json').read().strip().splitlines()[-1]); b=json.loads(open('gpurun_out/${TAG}_b$i.json').read().strip().splitlines()[-1])
print('round $i  default %.1f (frac %.4f)   $LIBB %.1f (frac %.4f)'%(a['value'],a['roofline']['frac'],b['value'],b['roofline']['frac']), flush=True)
PY
done
