#!/usr/bin/env python3
"""HBM traffic per launch of one kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

    python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> [kernel substring] [out.json]

Corrections per MI355X_MICROARCH.md (HBM section): both counters are in KiB; on gfx950 FETCH_SIZE
reports half the bytes of wide coalesced reads, so it is doubled."""
import collections
import csv
import json
import sys


def per_dispatch(path, counter, key):
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter or not any(k in r['Kernel_Name'] for k in key.split('|')):
            continue
        acc[r['Dispatch_Id']] = acc.get(r['Dispatch_Id'], 0.0) + float(r['Counter_Value'])
    return list(acc.values())


def main():
    key = sys.argv[3] if len(sys.argv) > 3 else 'igemm4_kernel|igemm3_kernel'
    f = per_dispatch(sys.argv[1], 'FETCH_SIZE', key)
    w = per_dispatch(sys.argv[2], 'WRITE_SIZE', key)
    rd = 2.0 * 1024.0 * sum(f) / max(len(f), 1)
    wr = 1024.0 * sum(w) / max(len(w), 1)
    out = {'kernel': key, 'launches_fetch_pass': len(f), 'launches_write_pass': len(w),
           'read_bytes_per_launch': rd, 'write_bytes_per_launch': wr, 'hbm_bytes_per_launch': rd + wr,
           'corrections': 'KiB -> bytes; FETCH_SIZE x2 (gfx950 wide reads)'}
    print(json.dumps(out, indent=1))
    if len(sys.argv) > 4:
        json.dump(out, open(sys.argv[4], 'w'), indent=1)


if __name__ == '__main__':
    main()
