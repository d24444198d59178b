for sk in 0 5 10 15 20 30; do
  export SWEM_KSPLIT_SKEW=$sk
  echo "skew $sk"
  python tools/conv_bench.py --reps 40 --only 1 --plan 0x670422 | tail -1
  python tools/conv_bench.py --reps 40 --only 2 --plan 0x670422 | tail -1
  python tools/conv_bench.py --reps 40 --only 3 --plan 0x670222 | tail -1
  python tools/conv_bench.py --reps 40 --only 11 --plan 0x670822 | tail -1
done
