export TMPDIR=/tmp
bash tools/profile_round.sh > gpurun_out/prof_round.log 2>&1
bash tools/pmc_kernels.sh gpurun_out/r03_conv_pmc.txt "conv_igemm_bf3s" python3 tools/conv_bench.py --reps 8 --only 0 --plan 0x630122 > gpurun_out/pmc_conv.log 2>&1
bash tools/pmc_kernels.sh gpurun_out/r03_em_pmc.txt "em_|match_|conv_igemm" python3 tools/em_loop.py --reps 10 > gpurun_out/pmc_em.log 2>&1
bash tools/pmc_bench_traffic.sh > gpurun_out/r03_conv_traffic.json 2> gpurun_out/pmc_traffic.err
python3 tools/pmc_by_kernel.py gpurun_out/pmc_traffic gpurun_out/r03_conv_traffic_by_kernel.json > gpurun_out/r03_conv_traffic_by_kernel.txt 2>&1
python3 tools/long_video.py --load-plans gpurun_out/prof_round/plans.json 2>/dev/null | tail -1 > gpurun_out/r03_config_e_long_video.json
python3 tools/train_bench.py --amp --steps 100 2>/dev/null | tail -1 > gpurun_out/r03_train_bench_amp_100steps.json
python3 tools/train_bench.py --steps 100 2>/dev/null | tail -1 > gpurun_out/r03_train_bench_fp32_100steps.json
python3 tools/em_bench.py --autotune > gpurun_out/r03_em_bench.txt 2>&1
ls -la gpurun_out/r03_*
