#!/bin/bash
# same-box A/B of the number of scoring pipelines (ALQ_LANES=1 / 2); usage: tools/ab_lanes.sh <tag>
TAG="${1:-lanes}"
mkdir -p gpurun_out
for rep in 1 2; do
  for L in 1 2; do
    ALQ_LANES=$L python bench.py --no-cpu-baseline --netb-pool 0 --steps 3 --warmup 1 > gpurun_out/${TAG}_L${L}_$rep.json 2> gpurun_out/${TAG}_L${L}_$rep.err || echo "failed L=$L"
    echo "lanes $L rep $rep: $(grep -o '"value": [0-9.]*' gpurun_out/${TAG}_L${L}_$rep.json | head -1)"
  done
done
