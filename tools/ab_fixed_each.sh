#!/bin/bash
# Folded vs runtime-constant instantiation, one launch at a time.  Needs a TEMPORARY hook that is not in the product: in the generated
# csrc/igemm4_fixed.inc, `if (!((mask >> N) & 1) && g4_matches<G4F_N>(a))` with mask = strtoull(getenv("ALQ_SKIP_FIXED")) (profiles/r03_experiments.txt 17).
set -eo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
OUT="$ROOT/gpurun_out/fixed_each"
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp ALQ_BENCH_NO_EVENTS=1
cd "$ROOT"
run() {  # tag mask
  ( export ALQ_SKIP_FIXED="$2"
    rocprofv3 --kernel-trace --stats -d "$OUT/$1" -o s --output-format csv -- python3 bench.py --pool 8000 --steps 2 --warmup 1 --no-cpu-baseline --netb-pool 0 > "$OUT/$1.json" 2> "$OUT/$1.err" )
}
run base 0
for n in 0 1 2 3 4 5 6 7 8 9 10 11; do run "s$n" $((1 << n)); done
run base2 0
python3 - "$OUT" <<'PY'
import csv, sys, json
out = sys.argv[1]
def load(tag):
    d = {}
    for row in csv.DictReader(open('%s/%s/s_kernel_stats.csv' % (out, tag))):
        if 'igemm4_kernel' not in row['Name']: continue
        n = row['Name'].replace('alq::', '').replace('void ', '').split('(')[0]
        d[n] = (float(row['AverageNs']) / 1e3, int(row['Calls']))
    return d
def val(tag): return json.loads(open('%s/%s.json' % (out, tag)).read().strip().splitlines()[-1])['value']
b1, b2 = load('base'), load('base2')
print('base %.0f  base2 %.0f patches/s' % (val('base'), val('base2')))
for n in range(12):
    d = load('s%d' % n)
    key = [k for k in b1 if k.endswith('G4F_%d>' % n)]
    if not key: continue
    k = key[0]
    rt = [x for x in d if 'G4Runtime' in x]
    tot_b = sum(u * c for u, c in b1.values()); tot_d = sum(u * c for u, c in d.values())
    print('G4F_%-2d folded %7.1f / %7.1f us   runtime %s   all igemm4 per pass %7.1f -> %7.1f us   bench %.0f' % (
        n, b1[k][0], b2[k][0], ' '.join('%.1f' % d[x][0] for x in rt), tot_b / 12, tot_d / 12, val('s%d' % n)))
PY
