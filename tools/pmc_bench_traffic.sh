#!/bin/bash
# HBM traffic of the conv kernels in one bench run: two separate --pmc passes (FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2)
export TMPDIR=/tmp
OUT=gpurun_out/pmc_traffic; rm -rf $OUT; mkdir -p $OUT
python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --seqs 1 --save-plans $OUT/plans.json > $OUT/tune.log 2>&1   # tune outside the profiler
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-em --no-graph --no-legs --load-plans $OUT/plans.json --seqs 1 > $OUT/$c.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob('gpurun_out/pmc_traffic/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        k = 'conv' if 'conv_igemm' in k or 'conv_splitk' in k or 'split_bf16x3' in k else 'other'
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'FETCH_SIZE' and 'igemm' in r['Kernel_Name']: n['conv_launches'] += 1
out = {'conv_launches': n['conv_launches'], 'FETCH_SIZE_KB': agg['conv']['FETCH_SIZE'], 'WRITE_SIZE_KB': agg['conv']['WRITE_SIZE'],
       'other_FETCH_KB': agg['other']['FETCH_SIZE'], 'other_WRITE_KB': agg['other']['WRITE_SIZE']}
print(json.dumps(out))
PY
