#!/bin/bash
# Small-pass sweep with HIP events on every launch (one pipeline): do intermediates of a small pass stay in the Infinity Cache?
# (round 6: no - per-patch launch times only grow below ~1000 patches per pass)     bash tools/small_pass_sweep.sh
for b in 2047 1000 500 250 125; do
python bench.py --lanes 1 --batch $b --prof-every 1 --no-accuracy --no-cpu-baseline --netb-pool 0 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
ts=r['time_share_ms_sampled']
print('batch', d['config']['batch'], 'value %.0f' % d['value'], 'frac %.3f' % r['frac'], 'avg launch us/patch %.3f' % (1e3*r['avg_launch_ms']/d['config']['batch']), 'sclk', d['clocks'] and round(d['clocks']['sclk_mhz_median']), 'W', d['clocks'] and round(d['clocks']['power_w_mean']))
print('   ', {k: round(v,1) for k,v in ts.items() if v>1})
"
done
