#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/pmc_conv2; rm -rf $OUT; mkdir -p $OUT
IDX=${1:-0}
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_VALU_MFMA_COEXEC_CYCLES SQ_IFETCH" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_WAVES GRBM_GUI_ACTIVE" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr" \
           "TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TOTAL_ACCESSES_sum"; do
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/g$i -- python3 tools/conv_bench.py --reps 5 --only $IDX > $OUT/g$i.log 2>&1
  i=$((i+1))
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob('gpurun_out/pmc_conv2/g*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'][:70]][r['Counter_Name']] += float(r['Counter_Value'])
for k, d in agg.items():
    if 'conv' not in k: continue
    print(k)
    for c, v in sorted(d.items()): print('   %-36s %.4g' % (c, v))
PY
