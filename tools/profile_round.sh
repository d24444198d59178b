#!/bin/bash
# Round profile (results under gpurun_out/prof_round/; copy the summaries into profiles/):
#  1. tune the conv plans once outside the profiler
#  2. rocprofv3 --kernel-trace --stats of an eager one-sequence run + the per-launch conv list -> conv_by_layer.csv
#  3. rocprofv3 --kernel-trace --stats of the DEFAULT bench command (four sequences, graph replay)
export TMPDIR=/tmp
OUT=gpurun_out/prof_round; rm -rf $OUT; mkdir -p $OUT
python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-em --seqs 1 --save-plans $OUT/plans.json > $OUT/tune.json 2> $OUT/tune.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/eager -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-em --no-graph --no-legs --seqs 1 --load-plans $OUT/plans.json --trace-layers $OUT/layers.json > $OUT/bench_eager.json 2> $OUT/eager.err
cp $(ls $OUT/eager/*/*kernel_stats.csv | head -1) $OUT/eager_kernel_stats.csv
python3 tools/conv_by_layer.py $OUT/layers.json $(ls $OUT/eager/*/*kernel_trace.csv | head -1) $OUT/conv_by_layer.csv > $OUT/conv_by_layer.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/default -- python3 bench.py --no-cpu-baseline --no-legs --load-plans $OUT/plans.json > $OUT/bench_default.json 2> $OUT/default.err
cp $(ls $OUT/default/*/*kernel_stats.csv | head -1) $OUT/default_kernel_stats.csv
rm -rf $OUT/eager $OUT/default
cat $OUT/conv_by_layer.txt; tail -c 600 $OUT/bench_default.json
