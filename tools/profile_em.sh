#!/bin/bash
# rocprofv3 kernel stats of the EM / matching launches alone (tools/em_loop.py).  Output: gpurun_out/prof_em/kernel_stats.csv
export TMPDIR=/tmp
OUT=gpurun_out/prof_em; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 tools/em_loop.py "$@" > $OUT/log.txt 2>&1
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
cut -d, -f1-4 $OUT/kernel_stats.csv | head -30
