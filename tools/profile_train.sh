#!/bin/bash
# Round profile of the training step (tools/train_bench.py): tune once outside the profiler, then rocprofv3
# --kernel-trace --stats of the graph-replayed step, fp32-accurate and config.AMP.  Results land in gpurun_out/prof_train.
export TMPDIR=/tmp
OUT=gpurun_out/prof_train; rm -rf $OUT; mkdir -p $OUT
python3 tools/train_bench.py --steps 1 --save-plans $OUT/plans_fp32.json > $OUT/tune_fp32.log 2>&1
python3 tools/train_bench.py --steps 1 --amp --save-plans $OUT/plans_amp.json > $OUT/tune_amp.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/fp32 -- python3 tools/train_bench.py --steps 4 --load-plans $OUT/plans_fp32.json > $OUT/bench_fp32.json 2> $OUT/fp32.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/amp -- python3 tools/train_bench.py --steps 4 --amp --load-plans $OUT/plans_amp.json > $OUT/bench_amp.json 2> $OUT/amp.log
cp $(ls $OUT/fp32/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_fp32.csv
cp $(ls $OUT/amp/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_amp.csv
rm -rf $OUT/fp32 $OUT/amp
tail -1 $OUT/bench_fp32.json; tail -1 $OUT/bench_amp.json
head -16 $OUT/kernel_stats_amp.csv | cut -c1-160
