#!/bin/bash
# PMC counters for the conv micro-benchmark (one rocprofv3 pass per counter group; no trace domains besides kernel-trace)
export TMPDIR=/tmp
OUT=gpurun_out/pmc_conv; mkdir -p $OUT
IDX=${1:-0}
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_WAVES GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE TCP_TCC_READ_REQ_sum"; do
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/g$i -- python3 tools/conv_bench.py --reps 5 --only $IDX > $OUT/g$i.log 2>&1
  i=$((i+1))
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob('gpurun_out/pmc_conv/g*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:60]
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); 
for k, d in agg.items():
    if 'conv' not in k: continue
    print(k)
    for c, v in sorted(d.items()): print('   %-32s %.4g' % (c, v))
PY
