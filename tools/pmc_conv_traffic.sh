#!/bin/bash
# HBM-side traffic (2*FETCH_SIZE + WRITE_SIZE, KB -> MB per launch) and time of one conv_bench shape under explicit plans:
#   bash tools/pmc_conv_traffic.sh <shape idx> <plan> [<plan> ...]
export TMPDIR=/tmp
IDX=$1; shift
for PLAN in "$@"; do
  OUT=gpurun_out/pmc_ct; rm -rf $OUT; mkdir -p $OUT
  python3 tools/conv_bench.py --reps 20 --only $IDX --plan $PLAN | tail -1
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -- python3 tools/conv_bench.py --reps 5 --only $IDX --plan $PLAN > $OUT/$c.log 2>&1
  done
  python3 - "$PLAN" <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(float); n = 0
for f in glob.glob('gpurun_out/pmc_ct/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'conv_igemm' in r['Kernel_Name']:
            agg[r['Counter_Name']] += float(r['Counter_Value'])
            n += r['Counter_Name'] == 'FETCH_SIZE'
print('   plan %s: fetch %.1f MB/launch (x2 = %.1f)  write %.1f MB/launch  [%d launches]' % (
    sys.argv[1], agg['FETCH_SIZE'] / 1024 / n, 2 * agg['FETCH_SIZE'] / 1024 / n, agg['WRITE_SIZE'] / 1024 / n, n))
PY
done
