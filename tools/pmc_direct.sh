#!/bin/bash
# PMC passes for one kernel family (diagnostic): per-kernel sums of a few SQ counters, for the library named by ALQ_LIB.
# usage: tools/pmc_direct.sh <tag>
set -eo pipefail
TAG="${1:-x}"
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
OUT="$ROOT/gpurun_out"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
ARGS="bench.py --pool 4000 --steps 1 --warmup 1 --no-cpu-baseline --netb-pool 0"
export ALQ_BENCH_NO_EVENTS=1
run() {
  local name="$1"; shift
  rocprofv3 --kernel-trace --pmc "$@" -d "$OUT/${TAG}_$name" -o "$name" --output-format csv -- python3 $ARGS > "$OUT/${TAG}_$name.json" 2> "$OUT/${TAG}_$name.err"
  echo "pass $name done"
}
run p1 SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE
run p2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD
run p3 SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
