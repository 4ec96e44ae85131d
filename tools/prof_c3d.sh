#!/bin/bash
# rocprofv3 passes over a short bench run (GPU box): kernel durations + two PMC passes, summarised per kernel by tools/pmc_kern.py
# usage: tools/prof_c3d.sh <tag> [pool]
set -eo pipefail
TAG="${1:-r04}"
POOL="${2:-8000}"
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
OUT="$ROOT/gpurun_out"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
ARGS="bench.py --pool $POOL --steps 1 --warmup 1 --no-cpu-baseline --netb-pool 0"
export ALQ_BENCH_NO_EVENTS=1
run() {
  local name="$1"; shift
  rocprofv3 "$@" -d "$OUT/${TAG}_$name" -o "$name" --output-format csv -- python3 $ARGS > "$OUT/${TAG}_$name.json" 2> "$OUT/${TAG}_$name.err"
  echo "pass $name done"
}
run stats --kernel-trace --stats
run pmcA --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
run pmcB --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE
python3 tools/pmc_kern.py "$OUT/${TAG}_stats" "$OUT/${TAG}_pmcA" "$OUT/${TAG}_pmcB" > "$OUT/${TAG}_kern.txt"
cat "$OUT/${TAG}_kern.txt"
