#!/bin/bash
# Per-kernel fixed cost of a pass: rocprofv3 --kernel-trace --stats of the bench at two batch sizes, then a + b n per kernel.
set -eo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
OUT="$ROOT/gpurun_out/fixed"
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp ALQ_BENCH_NO_EVENTS=1
cd "$ROOT"
for B in 500 2000; do
  rocprofv3 --kernel-trace --stats -d "$OUT/b$B" -o s --output-format csv -- python3 bench.py --batch $B --pool 8000 --steps 2 --warmup 1 --no-cpu-baseline --netb-pool 0 > "$OUT/b$B.json" 2> "$OUT/b$B.err"
done
python3 - "$OUT" <<'PY'
import csv, sys
out = sys.argv[1]
def load(b):
    d = {}
    for row in csv.DictReader(open('%s/b%d/s_kernel_stats.csv' % (out, b))):
        n = row['Name'].replace('alq::', '').replace('void ', '').split('(')[0]
        d[n] = (float(row['AverageNs']) / 1e3, int(row['Calls']))
    return d
a, b = load(500), load(2000)
ta = tb = 0.0
print('%-96s %9s %9s %9s %9s' % ('kernel', 'us@500', 'us@2000', 'fixed us', 'us/patch'))
for k, (u2, c2) in b.items():
    if k not in a: continue
    u1, c1 = a[k]
    per1, per2 = c1 / 48.0, c2 / 12.0          # launches per pass (8000 patches: 16 passes x 3 iterations at 500, 4 x 3 at 2000)
    slope = (u2 - u1) / 1500.0
    fixed = u1 - slope * 500
    if u2 * per2 < 15: continue
    print('%-96s %9.1f %9.1f %9.1f %9.4f  x%.0f' % (k[:96], u1, u2, fixed, slope, per2))
    ta += fixed * per2; tb += slope * per2
print('sum: fixed %.1f us per pass, %.4f us per patch' % (ta, tb))
PY
