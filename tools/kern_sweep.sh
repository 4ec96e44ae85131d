#!/bin/bash
# Per-kernel average durations (rocprofv3 --kernel-trace --stats) of several library builds on one box, R rounds, alternating.
#   tools/kern_sweep.sh "<kernel name substring>[,<substring>...]" R lib1.so lib2.so ...
set -eo pipefail
PAT="$1"; R="$2"; shift 2
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
OUT="$ROOT/gpurun_out/sweep"; rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp ALQ_BENCH_NO_EVENTS=1
cd "$ROOT"
for r in $(seq 1 "$R"); do
  for L in "$@"; do
    ALQ_LIB="$L" rocprofv3 --kernel-trace --stats -d "$OUT/${L}_$r" -o s --output-format csv -- python3 bench.py --lanes 1 --pool 16376 --steps 2 --warmup 1 --no-cpu-baseline --no-accuracy --netb-pool 0 > "$OUT/${L}_$r.json" 2> "$OUT/${L}_$r.err"
    echo "done $L round $r"
  done
done
python3 - "$OUT" "$PAT" "$R" "$@" <<'PY'
import csv, sys, json, glob
out, pat, R = sys.argv[1], sys.argv[2].split(','), int(sys.argv[3])
libs = sys.argv[4:]
for lib in libs:
    acc = {}
    vals = []
    for r in range(1, R + 1):
        f = glob.glob('%s/%s_%d/**/s_kernel_stats.csv' % (out, lib, r), recursive=True)[0]
        for row in csv.DictReader(open(f)):
            for p in pat:
                if p in row['Name']:
                    acc.setdefault(p, []).append(float(row['AverageNs']) / 1e3)
        vals.append(json.loads(open('%s/%s_%d.json' % (out, lib, r)).read().strip().splitlines()[-1])['value'])
    print('%-22s' % lib, '  '.join('%s %s' % (p, '/'.join('%.0f' % v for v in acc.get(p, []))) for p in pat), ' bench', '/'.join('%.0f' % v for v in vals), flush=True)
PY
