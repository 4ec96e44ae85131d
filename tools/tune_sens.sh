#!/bin/bash
# Per-launch durations of the igemm4 launches of a Fisher pass under different ALQ_G4_TUNE settings (diagnostic).
# usage: tools/tune_sens.sh "<setting>" "<setting>" ...      ("-" = no override); table by launch ordinal in the pass
set -eo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
OUT="$ROOT/gpurun_out"; mkdir -p "$OUT"; export TMPDIR=/tmp; cd "$ROOT"
export ALQ_BENCH_NO_EVENTS=1 ALQ_NO_SIDE_STREAM=1
i=0
for s in "$@"; do
  if [ "$s" = "-" ]; then unset ALQ_G4_TUNE; else export ALQ_G4_TUNE="$s"; fi
  rocprofv3 --kernel-trace -d "$OUT/tune_$i" -o s --output-format csv -- python3 bench.py --pool 8000 --steps 1 --warmup 1 --no-cpu-baseline --netb-pool 0 > "$OUT/tune_$i.json" 2> "$OUT/tune_$i.err" || echo "run $i failed"
  i=$((i+1))
done
python3 tools/tune_sens_table.py "$@"
