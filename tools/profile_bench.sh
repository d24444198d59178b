#!/bin/bash
# Round profile of the bench command: tune once outside the profiler, then (1) rocprofv3 --kernel-trace --stats of the
# eager single-sequence run (one launch per kernel record), (2) the two --pmc traffic passes (tools/pmc_bench_traffic.sh).
# Results land in gpurun_out/; copy the summaries into profiles/.
export TMPDIR=/tmp
OUT=gpurun_out/prof; rm -rf $OUT; mkdir -p $OUT
python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --seqs 1 --save-plans $OUT/plans.json > $OUT/tune.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-em --no-graph --no-legs --seqs 1 --load-plans $OUT/plans.json > $OUT/bench_eager.json 2> $OUT/stats.log
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
head -25 $OUT/kernel_stats.csv
tail -1 $OUT/bench_eager.json
bash tools/pmc_bench_traffic.sh | tail -1 > $OUT/traffic.json
cat $OUT/traffic.json
