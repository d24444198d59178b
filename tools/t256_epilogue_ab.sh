#!/bin/bash
# A/B of conv_t256_kernel's epilogue (round 6: accumulators staged through the LDS, ONE out_tile32 in a loop -- conv.hip,
# t256_epilogue) against the round-5 form (conv_epilogue16: twelve unrolled copies, 68 spilled registers), layer by layer and in
# the frame.  swem_amd/libswem_hip_oldepi.so = this tree with the two call sites switched back (built by hand beside the library).
# Output: gpurun_out/t256_epilogue_ab.txt
OUT=gpurun_out/t256_epilogue_ab.txt; mkdir -p gpurun_out; : > $OUT
for lib in swem_amd/libswem_hip_oldepi.so swem_amd/libswem_hip.so; do
  for bm in 1 2 10; do
    echo "== $lib  --dominant --t256 --bmul $bm (graph replay)" >> $OUT
    SWEM_HIP_LIB=$lib python3 tools/conv_bench.py --dominant --t256 --bmul $bm --graph --reps 30 >> $OUT 2>&1
  done
done
for rep in 1 2; do
  for lib in swem_amd/libswem_hip_oldepi.so swem_amd/libswem_hip.so; do
    echo "== $lib  bench.py (shipped plans, 4 sequences)" >> $OUT
    SWEM_HIP_LIB=$lib python3 bench.py --no-training --no-cpu-baseline --no-em --no-legs --no-roofline --steps 40 2>/dev/null \
      | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k: d[k] for k in ('value','value_min','value_max')})" >> $OUT
  done
done
cat $OUT
