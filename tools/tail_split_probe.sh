# tail split (plan bits 24-27) on the eight-wave 16x16x32 tile (variant 6), which the tuner never offered together
for pl in 0x670122 0x2670122 0x4670122 0x8670122 0x4270122 0x8270122 0xe70122 0x4e70122; do
  echo "plan $pl"
  python tools/conv_bench.py --reps 40 --only 0 --plan $pl | tail -1
  python tools/conv_bench.py --reps 40 --only 5 --plan $pl | tail -1
done
