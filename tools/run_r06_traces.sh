#!/bin/bash
# Step 2 of tools/run_r06_profiles.sh alone: the two kernel traces + the roofline recomputed from the --lanes 1 one.
set -eo pipefail
TAG="${1:-r06}"
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
OUT="$ROOT/gpurun_out"; mkdir -p "$OUT"; export TMPDIR=/tmp
cd "$ROOT"
rocprofv3 --kernel-trace --stats -d "$OUT/${TAG}_default_stats" -o stats --output-format csv -- python3 bench.py --no-cpu-baseline --no-accuracy --netb-pool 0 > "$OUT/${TAG}_bench_default_under_rocprof.json" 2> "$OUT/${TAG}_rp1.err"; echo "rocprof default done"
rocprofv3 --kernel-trace --stats -d "$OUT/${TAG}_lanes1_stats" -o stats --output-format csv -- python3 bench.py --lanes 1 --no-cpu-baseline --no-accuracy --netb-pool 0 > "$OUT/${TAG}_bench_lanes1_under_rocprof.json" 2> "$OUT/${TAG}_rp2.err"; echo "rocprof lanes 1 done"
S="$OUT/${TAG}_summaries"; mkdir -p "$S"
cp "$OUT/${TAG}_default_stats"/*/stats_kernel_stats.csv "$S/${TAG}_bench_default_kernel_stats.csv" 2>/dev/null || cp "$OUT/${TAG}_default_stats"/stats_kernel_stats.csv "$S/${TAG}_bench_default_kernel_stats.csv"
cp "$OUT/${TAG}_lanes1_stats"/*/stats_kernel_stats.csv "$S/${TAG}_bench_lanes1_kernel_stats.csv" 2>/dev/null || cp "$OUT/${TAG}_lanes1_stats"/stats_kernel_stats.csv "$S/${TAG}_bench_lanes1_kernel_stats.csv"
python3 tools/roofline_from_stats.py "$S/${TAG}_bench_lanes1_kernel_stats.csv" "$OUT/${TAG}_bench_lanes1_under_rocprof.json" "$ROOT/profiles/pmc_traffic.json" > "$S/${TAG}_roofline_from_stats.json"
rm -rf "$OUT/${TAG}_default_stats" "$OUT/${TAG}_lanes1_stats"
python3 -c "
import json; d=json.load(open('$S/${TAG}_roofline_from_stats.json')); print('trace frac', d['frac'], 'bench frac', d['bench_line']['frac'], 'ms per pass', d['igemm4_ms_per_batch_pass'], 'launches', d['igemm4_launches'])"
