#!/bin/bash
# One NET-B Fisher pass as the ordered list of its kernels with durations (single pipeline):  bash tools/netb_trace_order.sh [batch]
export TMPDIR=/tmp ALQ_LANES=1
B=${1:-2048}
rocprofv3 --kernel-trace -d gpurun_out/nbo -o t --output-format csv -- python3 tools/gpu_netb.py $B > gpurun_out/nbo.txt 2>&1
f=$(ls gpurun_out/nbo/*/t_kernel_trace.csv gpurun_out/nbo/t_kernel_trace.csv 2>/dev/null | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
# last pass: from the last 'unit_cotangent' back to the previous first-conv launch
idx = [i for i, n in enumerate(names) if 'fisher_finalize' in n]
end = idx[-1]; start = idx[-2] + 1
t0 = int(rows[start]['Start_Timestamp'])
for r in rows[start:end + 1]:
    n = r['Kernel_Name'].replace('alq::', '').replace('void ', '')
    print('%9.1f us  +%9.1f  %-5s %s' % ((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, (int(r['Start_Timestamp']) - t0) / 1e3, r.get('Stream_Id', ''), n[:110]))
PY
rm -rf gpurun_out/nbo
