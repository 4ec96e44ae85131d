#!/bin/bash
# Samples the GPU's power draw and clocks (rocm-smi, ~5 Hz) beside a bench run.
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
cd "$ROOT"; mkdir -p gpurun_out
OUT=gpurun_out/power_samples.txt
: > $OUT
( for i in $(seq 1 60); do
    /opt/rocm/bin/rocm-smi --showpower --showclocks --showtemp --showperflevel 2>&1 | grep -E "Power|sclk|mclk|fclk|Temp|junction|Performance" | tr '\n' ';' >> $OUT; echo >> $OUT
    sleep 0.2
  done ) &
SP=$!
python3 bench.py --steps 12 --warmup 2 --no-cpu-baseline --netb-pool 0 > gpurun_out/power_bench.json 2> gpurun_out/power_bench.err
kill $SP 2>/dev/null; wait $SP 2>/dev/null
tail -c 300 gpurun_out/power_bench.json; echo
sed -n '1p;10p;20p;30p;40p' $OUT
