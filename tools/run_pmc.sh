#!/bin/bash
# rocprofv3 passes of the bench workload on the GPU box (one process per pass, the program directly after `--`):
#   stats  : --kernel-trace --stats               -> per-kernel durations
#   mfma   : matrix-pipe counters                  -> MFMA busy, MFMA ops by type
#   lds    : LDS counters                          -> LDS active / bank conflicts / LDS issue stalls
#   fetch / write : FETCH_SIZE, WRITE_SIZE (separate passes: they do not fit one)
# usage: tools/run_pmc.sh <tag> [pool]      outputs under gpurun_out/<tag>_*; summarise with tools/pmc_report.py
set -eo pipefail
TAG="${1:-r02}"
POOL="${2:-8192}"
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
OUT="$ROOT/gpurun_out"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
# PMC_ARGS overrides the profiled program + arguments (e.g. PMC_ARGS="tools/gpu_netb.py 2048" for the NET-B passes)
ARGS="${PMC_ARGS:-bench.py --lanes 1 --pool $POOL --steps 1 --warmup 1 --no-cpu-baseline --no-accuracy --netb-pool 0}"
export ALQ_BENCH_NO_EVENTS=1
run() {   # name, extra rocprofv3 flags...
  local name="$1"; shift
  rocprofv3 "$@" -d "$OUT/${TAG}_$name" -o "$name" --output-format csv -- python3 $ARGS > "$OUT/${TAG}_$name.json" 2> "$OUT/${TAG}_$name.err"
  echo "pass $name done"
}
run stats --kernel-trace --stats
run mfma --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF SQ_INSTS_VALU_MFMA_MOPS_F SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE
run lds --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
run fetch --kernel-trace --pmc FETCH_SIZE
run write --kernel-trace --pmc WRITE_SIZE
