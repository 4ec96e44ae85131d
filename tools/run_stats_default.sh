#!/bin/bash
# rocprofv3 --kernel-trace --stats of the DEFAULT bench (100k pool, default batch; program directly after `--`), plus the
# same command without the profiler.  usage: tools/run_stats_default.sh <tag> [steps]
# outputs: gpurun_out/<tag>_default_stats/ (kernel stats csv), gpurun_out/<tag>_default_under_rocprof.json,
#          gpurun_out/<tag>_default.json; fold into profiles/ with tools/roofline_from_stats.py
set -eo pipefail
TAG="${1:-r03}"
STEPS="${2:-2}"
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
OUT="$ROOT/gpurun_out"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
python3 bench.py --steps "$STEPS" --warmup 1 > "$OUT/${TAG}_default.json" 2> "$OUT/${TAG}_default.err"
echo "plain bench done"
rocprofv3 --kernel-trace --stats -d "$OUT/${TAG}_default_stats" -o stats --output-format csv -- python3 bench.py --steps "$STEPS" --warmup 1 --no-cpu-baseline --netb-pool 0 > "$OUT/${TAG}_default_under_rocprof.json" 2> "$OUT/${TAG}_default_under_rocprof.err"
echo "rocprof bench done"
