#!/bin/bash
# launch-constant dumps for tools/gen_igemm4_fixed.py: the default launches, the conflict-free twin layout and the A/B switches
cd "$(dirname "${BASH_SOURCE[0]}")/.."
mkdir -p gpurun_out
run() { ALQ_DUMP_ARGS=1 python bench.py --pool 4000 --steps 1 --warmup 0 --no-cpu-baseline --netb-pool 0 > /dev/null 2> "gpurun_out/tunedump_$1.err"; grep -c G4ARGS "gpurun_out/tunedump_$1.err"; }
run default
ALQ_NO_ALT16=1 run noalt16
ALQ_NO_FLIPFIX=1 run noflip
ALQ_NO_BOUND16=1 run nobound16
ALQ_NO_F16X2=1 run nof16
ALQ_NO_SIGNS=1 run nosigns
ALQ_NO_SIGNS0=1 run nosigns0
# the forward-only pass (the AL loop's entropy filter): its six launches are other instantiations (no sums, no sign bytes)
ALQ_DUMP_ARGS=1 python tools/dump_forward.py > /dev/null 2> gpurun_out/tunedump_forward.err; grep -c G4ARGS gpurun_out/tunedump_forward.err
