#!/bin/bash
# Same-box A/B of two library builds at kernel level: rocprofv3 --kernel-trace --stats of a short bench per build,
# alternating, then the igemm4 launches' average durations side by side.
#   tools/ab_kernel.sh libalq_a.so libalq_b.so [rounds]      (builds: ALQ_OUT=libalq_b.so ALQ_BUILD_TAG=_b bash csrc/build.sh)
set -eo pipefail
A="$1"; B="$2"; R="${3:-2}"
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
OUT="$ROOT/gpurun_out/ab"
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp ALQ_BENCH_NO_EVENTS=1
cd "$ROOT"
for r in $(seq 1 "$R"); do
  for L in "$A" "$B"; do
    ALQ_LIB="$L" rocprofv3 --kernel-trace --stats -d "$OUT/${L}_$r" -o s --output-format csv -- python3 bench.py --pool 8000 --steps 2 --warmup 1 --no-cpu-baseline --netb-pool 0 > "$OUT/${L}_$r.json" 2> "$OUT/${L}_$r.err"
  done
done
python3 - "$OUT" "$A" "$B" "$R" <<'PY'
import csv, sys, collections, json
out, A, B, R = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
def load(lib):
    acc = collections.OrderedDict()
    for r in range(1, R + 1):
        for row in csv.DictReader(open('%s/%s_%d/s_kernel_stats.csv' % (out, lib, r))):
            n = row['Name'].replace('alq::', '').replace('void ', '').split('(')[0]
            acc.setdefault(n, []).append(float(row['AverageNs']) / 1e3)
    return {k: sum(v) / len(v) for k, v in acc.items()}
a, b = load(A), load(B)
ta = tb = 0.0
for k in a:
    if a[k] < 20: continue
    kb = b.get(k)
    print('%-78s %8.1f %8s %s' % (k[:78], a[k], '%.1f' % kb if kb else '-', '%+.1f%%' % (100 * (kb / a[k] - 1)) if kb else ''))
    if 'igemm4' in k and kb: ta += a[k]; tb += kb
for k in b:
    if k not in a and b[k] >= 20: print('%-78s %8s %8.1f' % (k[:78], '-', b[k]))
print('sum of matched igemm4 averages: %.1f -> %.1f us (%+.1f%%)' % (ta, tb, 100 * (tb / ta - 1)))
for lib in (A, B):
    v = [json.loads(open('%s/%s_%d.json' % (out, lib, r)).read().strip().splitlines()[-1])['value'] for r in range(1, R + 1)]
    print(lib, 'patches/s under rocprof:', ['%.0f' % x for x in v])
PY
