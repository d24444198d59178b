#!/bin/bash
# Round-6 profile set (results under gpurun_out/prof_r06/; the summaries are copied into profiles/r06_* by hand):
#   bash tools/profile_r06.sh [stats] [pmc] [traffic] [em]        (default: all)
# Default configuration since round 6: two lock-step lanes of four sequences (bench.py --seqs 8 --lockstep 4).
# Both arithmetics: the default leg (shipped plans, f16x3) and the exact-split leg (bench.py --math-modes 0 1: fp32 MFMA / bf16x6).
# Under rocprofv3 the program itself follows `--` (no shell / env wrapper).  Counters in their own passes, --kernel-trace only.
export TMPDIR=/tmp
OUT=gpurun_out/prof_r06; mkdir -p $OUT
WHAT="${*:-stats pmc traffic em}"
COMMON="--no-cpu-baseline --no-em --no-legs"
for leg in default exact; do
  MM=""; TAG=""
  if [ $leg = exact ]; then MM="--math-modes 0 1"; TAG="_exact"; fi
  if [[ $WHAT == *stats* ]]; then
    # ONE lock-step lane eagerly (the launches the timed graphs replay: four sequences in lock step, ten frames each per group), every
    # kernel a record, joined with the per-launch layer list -> conv_by_layer
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/eager$TAG -- python3 bench.py --steps 10 --warmup 2 $COMMON --no-graph --seqs 4 --lockstep 4 $MM --trace-layers $OUT/layers.json > $OUT/bench_eager$TAG.json 2> $OUT/eager$TAG.err
    cp $(ls $OUT/eager$TAG/*/*kernel_stats.csv | head -1) $OUT/bench_lane_eager_kernel_stats$TAG.csv
    python3 tools/conv_by_layer.py $OUT/layers.json$TAG $(ls $OUT/eager$TAG/*/*kernel_trace.csv | head -1) $OUT/conv_by_layer$TAG.csv > $OUT/conv_by_layer$TAG.txt 2>&1
    cp $OUT/layers.json$TAG $OUT/conv_layers_traced$TAG.json
    # eager one-sequence frames (the single-sequence leg's launches; rounds 2-5 profiled this form)
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/eager1$TAG -- python3 bench.py --steps 8 --warmup 2 $COMMON --no-graph --seqs 1 $MM --trace-layers $OUT/layers1.json > $OUT/bench_eager_seq1$TAG.json 2> $OUT/eager1$TAG.err
    cp $(ls $OUT/eager1$TAG/*/*kernel_stats.csv | head -1) $OUT/bench_seq1_eager_kernel_stats$TAG.csv
    python3 tools/conv_by_layer.py $OUT/layers1.json$TAG $(ls $OUT/eager1$TAG/*/*kernel_trace.csv | head -1) $OUT/conv_by_layer_seq1$TAG.csv > $OUT/conv_by_layer_seq1$TAG.txt 2>&1
    # the bench's own configuration (two lock-step lanes of four sequences, graph replay)
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/graph$TAG -- python3 bench.py --steps 20 --warmup 5 $COMMON --no-roofline $MM > $OUT/bench_profiled$TAG.json 2> $OUT/graph$TAG.err
    cp $(ls $OUT/graph$TAG/*/*kernel_stats.csv | head -1) $OUT/bench_default_kernel_stats$TAG.csv
    rm -rf $OUT/eager$TAG $OUT/eager1$TAG $OUT/graph$TAG
  fi
  if [[ $WHAT == *traffic* ]]; then
    for c in FETCH_SIZE WRITE_SIZE; do
      rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/traffic$TAG/$c -- python3 bench.py --steps 10 --warmup 2 $COMMON --no-graph --no-roofline --seqs 4 --lockstep 4 $MM > $OUT/traffic$TAG.$c.log 2>&1
    done
    python3 tools/pmc_by_kernel.py $OUT/traffic$TAG $OUT/conv_traffic_by_kernel$TAG.json > $OUT/conv_traffic_by_kernel$TAG.txt
    rm -rf $OUT/traffic$TAG
  fi
done
if [[ $WHAT == *pmc* ]]; then
  # the 128x128 kernel of each leg on its largest one-sequence layer (2x120x216 k3 256->256), the plans the shipped file holds for it
  bash tools/pmc_kernels.sh $OUT/conv_pmc.txt conv_igemm python3 tools/conv_bench.py --reps 8 --only 0 --plan 0x670122 > /dev/null
  bash tools/pmc_kernels.sh $OUT/conv_pmc_exact.txt conv_igemm python3 tools/conv_bench.py --reps 8 --only 0 --plan 0x8810122 > /dev/null
  # the 256-column tile kernel (round 5) on the same layer: 224-row tiles, and the ten-frame batch on 256-row tiles
  bash tools/pmc_kernels.sh $OUT/conv_pmc_t256.txt conv_t256 python3 tools/conv_bench.py --reps 8 --only 0 --plan 0x770144 > /dev/null
  bash tools/pmc_kernels.sh $OUT/conv_pmc_t256_b10.txt conv_t256 python3 tools/conv_bench.py --reps 8 --only 0 --plan 0x70144 --bmul 5 > /dev/null
  bash tools/pmc_kernels.sh $OUT/conv_pmc_t256_exact.txt conv_t256 python3 tools/conv_bench.py --reps 8 --only 0 --plan 0x410144 > /dev/null
  # the DOMINANT kernels of the lock-step lanes (round 6) on their largest layer, 8 objects x 120x216 k3 256->256, the shipped plans
  bash tools/pmc_kernels.sh $OUT/conv_pmc_t256_lane.txt conv_t256 python3 tools/conv_bench.py --reps 8 --only 0 --plan 0x770144 --bmul 4 > /dev/null
  bash tools/pmc_kernels.sh $OUT/conv_pmc_t256_lane_exact.txt conv_t256 python3 tools/conv_bench.py --reps 8 --only 0 --plan 0x410144 --bmul 4 > /dev/null
  # bench.py reads the dominant kernel's counters from conv_pmc[_exact].json: the lane entries join the 128x128 kernel's there
  python3 - $OUT <<'PY'
import json, sys
out = sys.argv[1]
for tag in ('', '_exact'):
    base = json.load(open('%s/conv_pmc%s.json' % (out, tag)))
    lane = json.load(open('%s/conv_pmc_t256_lane%s.json' % (out, tag)))
    base.update({k: dict(v, layer='8x120x216 k3 256->256 (a lock-step lane of four sequences x two objects)') for k, v in lane.items()})
    json.dump(base, open('%s/conv_pmc%s.json' % (out, tag), 'w'), indent=1, sort_keys=True)
PY
fi
if [[ $WHAT == *em* ]]; then
  bash tools/pmc_kernels.sh $OUT/em_pmc.txt 'em_|match_|conv_igemm' python3 tools/em_loop.py --reps 10 > /dev/null
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/emloop -- python3 tools/em_loop.py --reps 50 > $OUT/emloop.log 2>&1
  cp $(ls $OUT/emloop/*/*kernel_stats.csv | head -1) $OUT/em_loop_kernel_stats.csv; rm -rf $OUT/emloop
fi
rm -rf gpurun_out/pmc_conv_pmc gpurun_out/pmc_conv_pmc_exact gpurun_out/pmc_em_pmc gpurun_out/pmc_conv_pmc_t256 gpurun_out/pmc_conv_pmc_t256_b10 gpurun_out/pmc_conv_pmc_t256_exact gpurun_out/pmc_conv_pmc_t256_lane gpurun_out/pmc_conv_pmc_t256_lane_exact
ls -la $OUT | head -40
