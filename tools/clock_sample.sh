#!/bin/bash
# Samples the GPU's clocks and power (rocm-smi) twice a second while the default bench runs: is the chip power-throttled
# under the scoring pass?     bash tools/clock_sample.sh [tag] [bench args...]
tag=${1:-clk}; shift
out=gpurun_out/${tag}_clock_samples.txt
mkdir -p gpurun_out
: > $out
python bench.py --no-accuracy --steps 40 "$@" > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err &
bp=$!
while kill -0 $bp 2>/dev/null; do
    { date +%s.%N; rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -E "sclk|mclk|fclk|Power|busy"; } >> $out
    sleep 0.5
done
wait $bp
tail -c 600 gpurun_out/${tag}_bench.json | head -c 300; echo
grep -E "sclk" $out | sort | uniq -c | sort -rn | head -12
grep -E "Power" $out | awk '{print $NF}' | sort -n | tail -3
