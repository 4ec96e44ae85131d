#!/bin/bash
# Batch-size sweep of the default bench (same box, back to back): the plane-sweep kernels give one workgroup per CU a whole
# number of patches when the batch is a multiple of 256.    bash tools/batch_sweep.sh <tag> "<batches>"
set -e
TAG=${1:-sweep}; shift || true
BATCHES=${1:-"2000 2048 2560 3072 3584 3840"}
mkdir -p gpurun_out
for b in $BATCHES; do
  python bench.py --no-cpu-baseline --netb-pool 0 --batch $b --steps 2 --warmup 1 > gpurun_out/${TAG}_b$b.json 2> gpurun_out/${TAG}_b$b.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/${TAG}_b$b.json').read().strip().splitlines()[-1])
print('batch %5d  value %9.1f  ms/step %8.2f  frac %.4f'%($b, d['value'], d['ms_per_step'], d['roofline']['frac']), flush=True)
PY
done
