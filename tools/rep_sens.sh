set -e
export TMPDIR=/tmp ALQ_BENCH_NO_EVENTS=1 ALQ_NO_FIXED=1
cd $GRAFT_REPO_ROOT
for r in 0 1 2; do
  if [ $r = 0 ]; then unset ALQ_DEBUG_REPEAT; else export ALQ_DEBUG_REPEAT=$r; fi
  rocprofv3 --kernel-trace --stats -d gpurun_out/rep/r$r -o s --output-format csv -- python3 bench.py --pool 4000 --steps 1 --warmup 1 --no-cpu-baseline --netb-pool 0 > gpurun_out/rep/r$r.json 2> gpurun_out/rep/r$r.err
done
python3 - <<'PY'
import csv, collections
acc = collections.OrderedDict()
for r in range(3):
    for row in csv.DictReader(open('gpurun_out/rep/r%d/s_kernel_stats.csv' % r)):
        n = row['Name'].replace('alq::', '').replace('void ', '').split('(')[0]
        if 'igemm4' in n:
            acc.setdefault(n, [None]*3)[r] = float(row['AverageNs'])/1e3
for k, v in acc.items():
    print('%-80s %8.1f %8.1f %8.1f   +%.1f +%.1f' % (k[:80], v[0], v[1], v[2], v[1]-v[0], v[2]-v[1]))
PY
