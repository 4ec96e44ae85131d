#!/bin/bash
# Sensitivity runs (diagnostic builds libalq_dN.so made with -DALQ_DIAG=N: results are wrong on purpose, only the kernel
# durations matter): per-kernel average durations of each build under rocprofv3 --kernel-trace --stats.
set -eo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
OUT="$ROOT/gpurun_out"; mkdir -p "$OUT"; export TMPDIR=/tmp; cd "$ROOT"
export ALQ_BENCH_NO_EVENTS=1
LIBS="${@:-libalq.so libalq_d1.so libalq_d2.so libalq_d3.so libalq_d4.so}"
for lib in $LIBS; do
  [ -f "nn-active-learning_amd/$lib" ] || continue
  export ALQ_LIB=$lib
  rocprofv3 --kernel-trace --stats -d "$OUT/sens_$lib" -o s --output-format csv -- python3 bench.py --pool 4000 --steps 1 --warmup 1 --no-cpu-baseline --netb-pool 0 > "$OUT/sens_$lib.json" 2> "$OUT/sens_$lib.err" || echo "run failed for $lib"
done
python3 tests/diag_sens_table.py $LIBS
