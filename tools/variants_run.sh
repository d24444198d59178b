#!/bin/bash
# tools/variants_run.sh OUT NAME...: tools/conv_bench.py --dominant on each experimental build (tools/conv_variant.sh), the
# base build first and last (clock / box drift shows as the difference between the two base runs)
out=$1; shift
: > $out
for n in base "$@" base; do
  echo "== $n" >> $out
  SWEM_HIP_LIB=swem_amd/libswem_hip_$n.so python tools/conv_bench.py --dominant --reps 40 2>&1 | grep -v amdgpu.ids >> $out
done
