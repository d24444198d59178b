#!/bin/bash
# A/B of how the training step runs its clips (train.py, SWEMTrainer(lanes=...)): 4 lanes x 1 clip (rounds 1-5), 2 lanes x 2 clips,
# 1 lane x 4 clips (the reference's batched step), in both arithmetics, one process each on ONE box.
# Output: gpurun_out/train_lanes_ab.txt        usage: tools/train_lanes_ab.sh [steps]
STEPS=${1:-30}
OUT=gpurun_out/train_lanes_ab.txt; mkdir -p gpurun_out; : > $OUT
for amp in "" "--amp"; do
  for lanes in 4 2 1; do
    echo "== lanes $lanes $amp" >> $OUT
    python3 tools/train_bench.py --clips 4 --steps $STEPS --lanes $lanes $amp >> $OUT 2>> gpurun_out/train_lanes_ab.err
  done
done
cat $OUT
