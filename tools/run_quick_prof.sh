#!/bin/bash
# quick per-kernel picture of one build: kernel stats + the MFMA / VALU / LDS counter passes (pool 8000, 1 step)
# usage: tools/run_quick_prof.sh <tag>   -> gpurun_out/<tag>_{stats,mfma,lds}*; summarise with tools/pmc_kernels.py
set -eo pipefail
TAG="${1:-q}"
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
OUT="$ROOT/gpurun_out"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
ARGS="bench.py --pool 8000 --steps 1 --warmup 1 --no-cpu-baseline --netb-pool 0"
export ALQ_BENCH_NO_EVENTS=1
rocprofv3 --kernel-trace --stats -d "$OUT/${TAG}_stats" -o stats --output-format csv -- python3 $ARGS > "$OUT/${TAG}_stats.json" 2> "$OUT/${TAG}_stats.err"
echo stats done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE -d "$OUT/${TAG}_mfma" -o mfma --output-format csv -- python3 $ARGS > "$OUT/${TAG}_mfma.json" 2> "$OUT/${TAG}_mfma.err"
echo mfma done
rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d "$OUT/${TAG}_lds" -o lds --output-format csv -- python3 $ARGS > "$OUT/${TAG}_lds.json" 2> "$OUT/${TAG}_lds.err"
echo lds done
