#!/bin/bash
# The round's evidence in one GPU call (results under gpurun_out/; copy the summaries into profiles/rNN_*):
#   kernel stats + per-layer join (profile_round.sh), SQ / TCC counters of the dominant conv kernel and of the EM loop
#   (pmc_kernels.sh: rNN_conv_pmc.{txt,json}, rNN_em_pmc.{txt,json}), fabric traffic (pmc_bench_traffic.sh + pmc_by_kernel.py),
#   config E (long_video.py), the training step (train_bench.py, AMP and fp32 level, 100 steps), the EM bench.
export TMPDIR=/tmp
R=${1:-r03}
bash tools/profile_round.sh > gpurun_out/prof_round.log 2>&1
bash tools/pmc_kernels.sh gpurun_out/${R}_conv_pmc.txt "conv_igemm_bf3s" python3 tools/conv_bench.py --reps 8 --only 0 --plan 0x630122 > gpurun_out/pmc_conv.log 2>&1
bash tools/pmc_kernels.sh gpurun_out/${R}_em_pmc.txt "em_|match_|conv_igemm" python3 tools/em_loop.py --reps 10 > gpurun_out/pmc_em.log 2>&1
bash tools/pmc_bench_traffic.sh > gpurun_out/${R}_conv_traffic.json 2> gpurun_out/pmc_traffic.err
python3 tools/pmc_by_kernel.py gpurun_out/pmc_traffic gpurun_out/${R}_conv_traffic_by_kernel.json > gpurun_out/${R}_conv_traffic_by_kernel.txt 2>&1
python3 tools/long_video.py --load-plans gpurun_out/prof_round/plans.json 2>/dev/null | tail -1 > gpurun_out/${R}_config_e_long_video.json
python3 tools/train_bench.py --amp --steps 100 2>/dev/null | tail -1 > gpurun_out/${R}_train_bench_amp_100steps.json
python3 tools/train_bench.py --steps 100 2>/dev/null | tail -1 > gpurun_out/${R}_train_bench_fp32_100steps.json
python3 tools/em_bench.py --autotune > gpurun_out/${R}_em_bench.txt 2>&1
ls -la gpurun_out/${R}_*
