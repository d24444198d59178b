export TMPDIR=/tmp
bash tools/pmc_kernels.sh gpurun_out/r03_conv_pmc.txt "conv_igemm_bf3s" python3 tools/conv_bench.py --reps 8 --only 0 --plan 0x630122 > gpurun_out/pmc_conv.log 2>&1
bash tools/pmc_kernels.sh gpurun_out/r03_em_pmc.txt "em_|match_|conv_igemm" python3 tools/em_loop.py --reps 10 > gpurun_out/pmc_em.log 2>&1
python3 tools/train_bench.py --amp --steps 100 2>/dev/null | tail -1 > gpurun_out/r03_train_bench_amp_100steps.json
python3 tools/train_bench.py --steps 100 2>/dev/null | tail -1 > gpurun_out/r03_train_bench_fp32_100steps.json
python3 tools/train_bench.py --amp --steps 100 2>/dev/null | tail -1 > gpurun_out/r03_train_bench_amp_100steps_b.json
cat gpurun_out/r03_train_bench_*.json | cut -c1-200
tail -12 gpurun_out/r03_conv_pmc.txt
