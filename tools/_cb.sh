for a in "3 0x630222" "1 0x630422" "2 0x630422" "4 0x630222" "11 0x630822" "7 0x30411"; do set -- $a; python tools/conv_bench.py --reps 40 --only $1 --plan $2 | tail -1; done
