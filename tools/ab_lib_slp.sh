for i in 1 2 3; do
for lib in "" libalq_slp.so; do
  if [ -n "$lib" ]; then export ALQ_LIB=$PWD/nn-active-learning_amd/$lib; else unset ALQ_LIB; fi
  python bench.py --steps 3 --warmup 1 --no-accuracy --no-cpu-baseline --netb-pool 0 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${lib:-default}', round(d['value']), round(d['roofline']['frac'],4), {k: round(v,1) for k,v in d['roofline']['time_share_ms_sampled'].items() if v > 5})"
done; done
