for pl in 0x630122 0x1630022 0x1530022 0x1230022; do python tools/conv_bench.py --reps 30 --only 0 --plan $pl | tail -1; done
for pl in 0x630222 0x1630022 0x1530022; do python tools/conv_bench.py --reps 30 --only 3 --plan $pl | tail -1; done
for pl in 0x630422 0x1630022 0x1530022; do python tools/conv_bench.py --reps 30 --only 2 --plan $pl | tail -1; done
for pl in 0x630422 0x1630022; do python tools/conv_bench.py --reps 30 --only 1 --plan $pl | tail -1; done
for pl in 0x630222 0x1630022; do python tools/conv_bench.py --reps 30 --only 4 --plan $pl | tail -1; done
for pl in 0x30111 0x1030011 0x1030021 0x1630022; do python tools/conv_bench.py --reps 30 --only 9 --plan $pl | tail -1; done
for pl in 0x30811 0x1030011 0x1630022; do python tools/conv_bench.py --reps 30 --only 11 --plan $pl | tail -1; done
