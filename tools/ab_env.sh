#!/bin/bash
# Same-box A/B of one library under two environments at kernel level (rocprofv3 --kernel-trace --stats of a short bench each, alternating).
#   tools/ab_env.sh "ENVA=1" "ENVB=1 ENVC=2" [rounds]     ("-" = no extra variables)
set -eo pipefail
EA="$1"; EB="$2"; R="${3:-2}"
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
OUT="$ROOT/gpurun_out/abenv"
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp ALQ_BENCH_NO_EVENTS=1
cd "$ROOT"
for r in $(seq 1 "$R"); do
  for L in a b; do
    E="$EA"; [ "$L" = b ] && E="$EB"
    [ "$E" = "-" ] && E=""
    ( for kv in $E; do export "$kv"; done
      rocprofv3 --kernel-trace --stats -d "$OUT/${L}_$r" -o s --output-format csv -- python3 bench.py --pool 8000 --steps 2 --warmup 1 --no-cpu-baseline --netb-pool 0 > "$OUT/${L}_$r.json" 2> "$OUT/${L}_$r.err" )
  done
done
python3 - "$OUT" "$R" <<'PY'
import csv, sys, collections, json
out, R = sys.argv[1], int(sys.argv[2])
def load(lib):
    acc = collections.OrderedDict()
    for r in range(1, R + 1):
        for row in csv.DictReader(open('%s/%s_%d/s_kernel_stats.csv' % (out, lib, r))):
            n = row['Name'].replace('alq::', '').replace('void ', '').split('(')[0]
            acc.setdefault(n, []).append((float(row['AverageNs']) / 1e3, int(row['Calls'])))
    return {k: (sum(x[0] for x in v) / len(v), v[0][1]) for k, v in acc.items()}
a, b = load('a'), load('b')
for nm, d in (('A', a), ('B', b)):
    print('== %s: kernels >= 20 us (avg us, calls); sum of avg*calls per pass below' % nm)
    tot = 0.0
    for k, (us, calls) in d.items():
        if us >= 20: print('  %-100s %8.1f %5d' % (k[:100], us, calls))
        tot += us * calls
    print('  total kernel time %.1f ms' % (tot / 1e3))
for lib in ('a', 'b'):
    v = [json.loads(open('%s/%s_%d.json' % (out, lib, r)).read().strip().splitlines()[-1])['value'] for r in range(1, R + 1)]
    print(lib, 'patches/s under rocprof:', ['%.0f' % x for x in v])
PY
