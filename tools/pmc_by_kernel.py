#!/usr/bin/env python
"""Per-kernel HBM-side traffic from the two --pmc passes of tools/pmc_bench_traffic.sh:
   python tools/pmc_by_kernel.py [gpurun_out/pmc_traffic [out.json]]  ->  (2*FETCH_SIZE + WRITE_SIZE) KB per launch, by kernel."""
import collections
import csv
import glob
import re
import sys

root = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/pmc_traffic'
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob(root + '/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
        k = re.sub(r'\(.*', '', k)
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'FETCH_SIZE':
            cnt[k] += 1
rows = sorted(agg.items(), key=lambda kv: -(2 * kv[1]['FETCH_SIZE'] + kv[1]['WRITE_SIZE']))
if len(sys.argv) > 2:      # machine-readable copy: bench.py reads the dominant kernel's figure from it
    import json
    with open(sys.argv[2], 'w') as f:
        json.dump({k: {'launches': cnt[k], 'bytes_per_launch': int((2 * v['FETCH_SIZE'] + v['WRITE_SIZE']) * 1024 / max(cnt[k], 1))}
                   for k, v in rows}, f, indent=1)
tot = sum(2 * v['FETCH_SIZE'] + v['WRITE_SIZE'] for _, v in rows)
for k, v in rows[:25]:
    t = 2 * v['FETCH_SIZE'] + v['WRITE_SIZE']
    print('%-64s %6d launches  %9.1f MB total  %8.2f MB/launch  %5.1f%%' % (k[:64], cnt[k], t / 1024, t / 1024 / max(cnt[k], 1), 100 * t / tot))
