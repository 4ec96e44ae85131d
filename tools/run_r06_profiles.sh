#!/bin/bash
# The tracked profile set of round 6 (GPU box):  bash tools/run_r06_profiles.sh r06
#   1. default bench (two pipelines timed, roofline from the separate single-pipeline pass) plain
#   2. rocprofv3 --kernel-trace --stats of the default command and of `--lanes 1` (per-launch durations without co-residency);
#      both without the accuracy passes behind the timed region, so that the trace holds the (warmup + steps) x pool patches
#      tools/roofline_from_stats.py divides by and nothing else
#   3. the PMC passes of tools/run_pmc.sh (bench pool 8188) and of NET-B (tools/gpu_netb.py 2048)
#   4. one line per other config (0, 1, 3, 4, 5)
set -eo pipefail
TAG="${1:-r06}"
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
OUT="$ROOT/gpurun_out"; mkdir -p "$OUT"; export TMPDIR=/tmp
cd "$ROOT"
python3 bench.py > "$OUT/${TAG}_bench_default.json" 2> "$OUT/${TAG}_bench_default.err"; echo "plain bench done"
rocprofv3 --kernel-trace --stats -d "$OUT/${TAG}_default_stats" -o stats --output-format csv -- python3 bench.py --no-cpu-baseline --no-accuracy --netb-pool 0 > "$OUT/${TAG}_bench_default_under_rocprof.json" 2> "$OUT/${TAG}_rp1.err"; echo "rocprof default done"
rocprofv3 --kernel-trace --stats -d "$OUT/${TAG}_lanes1_stats" -o stats --output-format csv -- python3 bench.py --lanes 1 --no-cpu-baseline --no-accuracy --netb-pool 0 > "$OUT/${TAG}_bench_lanes1_under_rocprof.json" 2> "$OUT/${TAG}_rp2.err"; echo "rocprof lanes 1 done"
for c in 0 1 3 4 5; do python3 bench.py --config $c > "$OUT/${TAG}_bench_config$c.json" 2> "$OUT/${TAG}_c$c.err"; echo "config $c done"; done
bash tools/run_pmc.sh ${TAG}pmc 8188
PMC_ARGS="tools/gpu_netb.py 2048" bash tools/run_pmc.sh ${TAG}netb
# summaries (the raw counter tables are large: only the summaries and the kernel-stats tables travel back)
S="$OUT/${TAG}_summaries"; mkdir -p "$S"
python3 tools/pmc_report.py "$OUT/${TAG}pmc" "$S/${TAG}_pmc_summary.json" "$S/pmc_traffic.json" 2047 > "$S/${TAG}_pmc_report.txt"
PMC_KERNELS="igemm4_kernel,igemm3_kernel,igemm_kernel,fcgemm_kernel" PMC_LAUNCHES_PER_PASS=11 \
  PMC_KERNEL_NOTE="the contraction launches of a NET-B pass (igemm4 / igemm3 / igemm conv launches + fcgemm fc launches), tools/gpu_netb.py 2048" \
  python3 tools/pmc_report.py "$OUT/${TAG}netb" "$S/${TAG}_netb_pmc_summary.json" "$S/netb_pmc_traffic.json" 2048 > "$S/${TAG}_netb_pmc_report.txt"
cp "$OUT/${TAG}_default_stats"/*/stats_kernel_stats.csv "$S/${TAG}_bench_default_kernel_stats.csv" 2>/dev/null || cp "$OUT/${TAG}_default_stats"/stats_kernel_stats.csv "$S/${TAG}_bench_default_kernel_stats.csv"
cp "$OUT/${TAG}_lanes1_stats"/*/stats_kernel_stats.csv "$S/${TAG}_bench_lanes1_kernel_stats.csv" 2>/dev/null || cp "$OUT/${TAG}_lanes1_stats"/stats_kernel_stats.csv "$S/${TAG}_bench_lanes1_kernel_stats.csv"
cp "$OUT/${TAG}netb_stats"/*/stats_kernel_stats.csv "$S/${TAG}_netb_kernel_stats.csv" 2>/dev/null || cp "$OUT/${TAG}netb_stats"/stats_kernel_stats.csv "$S/${TAG}_netb_kernel_stats.csv"
python3 tools/roofline_from_stats.py "$S/${TAG}_bench_lanes1_kernel_stats.csv" "$OUT/${TAG}_bench_lanes1_under_rocprof.json" "$S/pmc_traffic.json" > "$S/${TAG}_roofline_from_stats.json"
rm -rf "$OUT/${TAG}_default_stats" "$OUT/${TAG}_lanes1_stats" "$OUT/${TAG}pmc_"* "$OUT/${TAG}netb_"*
ls -la "$S"
echo "all done"
