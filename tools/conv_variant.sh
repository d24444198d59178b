#!/bin/bash
# Build an experimental variant of conv.hip into its own library:  tools/conv_variant.sh NAME "-DSOME_SWITCH=1 ..."
# (the experiment switches of rounds 1-5 -- stagger, set-priority, MFMAs in front of the hand-over, late issue, ablations -- are no
# longer in the shipped source: apply profiles/r06_experiments/conv_dead_switches.patch to get them back)
# -> swem_amd/libswem_hip_NAME.so (conv.hip compiled with -DSWEM_ISA_SUBSET: a fifth of the instantiations, about a minute;
# every other object as built).  Run with  SWEM_HIP_LIB=swem_amd/libswem_hip_NAME.so python tools/conv_bench.py --dominant
set -e
cd "$(dirname "$0")/.."
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DSWEM_ISA_SUBSET $@ \
  -c swem_amd/csrc/conv.hip -o /tmp/conv_$name.o
# (the second unit of conv.hip: conv_t256_kernel alone, with the same flags)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DSWEM_CONV_T256_ONLY $@ \
  -c swem_amd/csrc/conv.hip -o /tmp/conv_t256_$name.o
objs="/tmp/conv_t256_$name.o"
for o in api bneck pointwise em match train train_conv; do objs="$objs swem_amd/csrc/$o.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o swem_amd/libswem_hip_$name.so /tmp/conv_$name.o $objs
echo swem_amd/libswem_hip_$name.so
