#!/bin/bash
# A/B of how the training step runs its clips (train.py, SWEMTrainer(lanes=..., wgrad_stream=...)): 4 lanes x 1 clip (rounds 1-5),
# 2 lanes x 2 clips, 1 lane x 4 clips (the reference's batched step), each with the weight gradients on the lane's own stream (0) or
# on a second stream beside the data-gradient chain (1), in both arithmetics, one process each on ONE box.
# Output: gpurun_out/train_lanes_ab.txt        usage: tools/train_lanes_ab.sh [steps] ["amp-flags"...]
STEPS=${1:-30}
OUT=gpurun_out/train_lanes_ab.txt; mkdir -p gpurun_out; : > $OUT
for amp in "" "--amp"; do
  for lanes in 4 2 1; do
    for ws in 0 1; do
      echo "== lanes $lanes wgrad_stream $ws $amp" >> $OUT
      SWEM_TRAIN_WGRAD_STREAM=$ws python3 tools/train_bench.py --clips 4 --steps $STEPS --lanes $lanes --no-roofline $amp 2>> gpurun_out/train_lanes_ab.err \
        | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: d[k] for k in ('value','ms_per_step','lanes','clips_per_lane','peak_mem_GB')})" >> $OUT
    done
  done
done
cat $OUT
