#!/bin/bash
# Round-6 measurement set on one box: shipped training plans, the r06 profile set, launches per training clip, the default bench line.
export TMPDIR=/tmp
mkdir -p gpurun_out swem_amd/plans
python3 tools/profile_stamp.py
# training plans (what bench.py's `training` leg and tools/train_bench.py --load-plans read)
python3 tools/train_bench.py --clips 4 --steps 10 --save-plans swem_amd/plans/mi355x_train_384_k256_fp32_level.json --cpu-baseline > gpurun_out/r06_train_bench_fp32_level.json 2> gpurun_out/r06_train_bench_fp32_level.err
python3 tools/train_bench.py --clips 4 --steps 10 --amp --save-plans swem_amd/plans/mi355x_train_384_k256_amp.json > gpurun_out/r06_train_bench_amp.json 2> gpurun_out/r06_train_bench_amp.err
cp swem_amd/plans/mi355x_train_384_k256_*.json gpurun_out/
# launches / kernel time per clip, shipped default (four lanes)
tools/train_launches.sh > gpurun_out/r06_tl_f16x3.txt 2>&1; cp gpurun_out/train_launches/by_kernel.csv gpurun_out/r06_train_launches_f16x3.csv
tools/train_launches.sh --amp > gpurun_out/r06_tl_amp.txt 2>&1; cp gpurun_out/train_launches/by_kernel.csv gpurun_out/r06_train_launches_amp.csv
# the default bench line (with the training sub-record), then the profile set
python3 bench.py > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err
bash tools/profile_r06.sh > gpurun_out/r06_profile.log 2>&1
tail -3 gpurun_out/r06_bench_default.json | cut -c1-600
cat gpurun_out/r06_train_bench_fp32_level.json gpurun_out/r06_train_bench_amp.json | cut -c1-1500
