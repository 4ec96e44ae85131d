cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp ALQ_BENCH_NO_EVENTS=1
for r in 1 2; do for v in base alt; do
  if [ $v = base ]; then export ALQ_NO_ALT16=1; else unset ALQ_NO_ALT16; fi
  rocprofv3 --kernel-trace --stats -d gpurun_out/alt/${v}_$r -o s --output-format csv -- python3 bench.py --pool 8000 --steps 2 --warmup 1 --no-cpu-baseline --netb-pool 0 > gpurun_out/alt/${v}_$r.json 2>/dev/null
done; done
python3 - <<'PY'
import csv, collections, json
def load(v):
    acc=collections.OrderedDict()
    for r in (1,2):
        for row in csv.DictReader(open('gpurun_out/alt/%s_%d/s_kernel_stats.csv'%(v,r))):
            if 'igemm4' in row['Name']:
                n=row['Name'].split('igemm4_kernel')[1].split('(')[0]
                acc.setdefault(n,[]).append(float(row['AverageNs'])/1e3)
    return {k:sum(x)/len(x) for k,x in acc.items()}
a,b=load('base'),load('alt')
print('base only:', {k:round(v) for k,v in a.items() if k not in b})
print('alt only:', {k:round(v) for k,v in b.items() if k not in a})
print('sum base %.0f alt %.0f'%(sum(a.values()),sum(b.values())))
for v in ('base','alt'):
    print(v,[round(json.loads(open('gpurun_out/alt/%s_%d.json'%(v,r)).read().strip().splitlines()[-1])['value']) for r in (1,2)])
PY
