mkdir -p gpurun_out
python bench.py --no-cpu-baseline --netb-pool 0 --steps 4 --warmup 1 > gpurun_out/two_single.json 2>/dev/null
python bench.py --no-cpu-baseline --netb-pool 0 --pool 50000 --steps 8 --warmup 2 > gpurun_out/two_a.json 2>/dev/null &
P1=$!
python bench.py --no-cpu-baseline --netb-pool 0 --pool 50000 --steps 8 --warmup 2 > gpurun_out/two_b.json 2>/dev/null &
P2=$!
wait $P1; wait $P2
for f in two_single two_a two_b; do grep -o '"value": [0-9.]*' gpurun_out/$f.json | head -1; done
