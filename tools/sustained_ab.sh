run() { tag=$1; shift; env "$@" python bench.py --steps 20 --warmup 2 --no-accuracy --no-cpu-baseline --netb-pool 0 ${EXTRA} 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['clocks']; print('$tag', round(d['value']), 'sclk', round(c['sclk_mhz_median']), 'W', round(c['power_w_mean']))"; }
EXTRA="" run "default(2 lanes)" A=1
EXTRA="--lanes 1" run "lanes1" A=1
EXTRA="--lanes 1" run "r5-like(lanes1,bwd8,nobalance)" ALQ_C3D_BWD_ROWS=8 ALQ_NO_PASS_BALANCE=1
EXTRA="" run "default(2 lanes) again" A=1
EXTRA="--lanes 3" run "lanes3" A=1
