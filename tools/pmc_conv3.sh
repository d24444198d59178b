#!/bin/bash
# PMC counters of one conv_bench shape under an explicit plan:  bash tools/pmc_conv3.sh <shape idx> <plan>
export TMPDIR=/tmp
OUT=gpurun_out/pmc_conv3; rm -rf $OUT; mkdir -p $OUT
IDX=${1:-0}; PLAN=${2:-0x10021}
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_ACTIVE_INST_MISC" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/g$i -- python3 tools/conv_bench.py --reps 5 --only $IDX --plan $PLAN > $OUT/g$i.log 2>&1
  i=$((i+1))
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob('gpurun_out/pmc_conv3/g*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'][:80]][r['Counter_Name']] += float(r['Counter_Value'])
for k, d in agg.items():
    if 'conv_igemm' not in k: continue
    print(k)
    for c, v in sorted(d.items()): print('   %-36s %.4g' % (c, v))
PY
