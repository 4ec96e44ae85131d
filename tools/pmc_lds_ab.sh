export TMPDIR=/tmp
for v in alt noalt; do
  if [ $v = noalt ]; then export ALQ_NO_ALT16=1; else unset ALQ_NO_ALT16; fi     # the conflict-free twin is on by default (igemm4.hip)
  ALQ_BENCH_NO_EVENTS=1 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CU_CYCLES -d gpurun_out/lds_$v -o lds --output-format csv -- python3 bench.py --pool 4000 --steps 1 --warmup 0 --no-cpu-baseline --netb-pool 0 > /dev/null 2>&1
done
python3 - <<'PY'
import csv,collections
for v in ('alt','noalt'):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    seen=set()
    for r in csv.DictReader(open('gpurun_out/lds_%s/lds_counter_collection.csv'%v)):
        k=r['Kernel_Name'].replace('alq::','').replace('void ','').split('(')[0]
        if 'igemm4' not in k: continue
        acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
        if (r['Dispatch_Id']) not in seen: seen.add(r['Dispatch_Id']); n[k]+=1
        acc[k]['dur']+= 0
    print(v)
    for k,c in acc.items():
        print('  %-60s n=%d conflict/active %.3f active/busy %.3f' % (k[:60], n[k], c['SQ_LDS_BANK_CONFLICT']/max(c['SQ_LDS_IDX_ACTIVE'],1), c['SQ_LDS_IDX_ACTIVE']/max(c['SQ_BUSY_CU_CYCLES'],1)))
PY
