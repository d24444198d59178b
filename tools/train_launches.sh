#!/bin/bash
# Launches per training clip by kernel: rocprofv3 kernel stats of two tools/train_bench.py runs that differ by 10 steps
# (everything the warm-up, the tuner and the graph capture launch cancels in the difference).
# Output: gpurun_out/train_launches/{by_kernel.csv,summary.txt}      usage: tools/train_launches.sh [--amp]
export TMPDIR=/tmp
OUT=gpurun_out/train_launches; rm -rf $OUT; mkdir -p $OUT
python3 tools/train_bench.py --clips 4 --steps 2 --warmup 1 --save-plans $OUT/plans.json "$@" > $OUT/tune.log 2>&1
for s in 3 13; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s$s -- python3 tools/train_bench.py --clips 4 --steps $s --warmup 1 \
      --load-plans $OUT/plans.json "$@" > $OUT/run$s.log 2>&1
  cp $(ls $OUT/s$s/*/*kernel_stats.csv | head -1) $OUT/stats$s.csv
done
python3 - "$OUT" <<'PY'
import csv, sys, re
out = sys.argv[1]
def load(p):
    d = {}
    for r in csv.DictReader(open(p)):
        d[r['Name']] = (int(r['Calls']), int(r['TotalDurationNs']))
    return d
a, b = load(out + '/stats3.csv'), load(out + '/stats13.csv')
rows = []
for k in b:
    c = (b[k][0] - a.get(k, (0, 0))[0]) / 40.0
    t = (b[k][1] - a.get(k, (0, 0))[1]) / 40.0 / 1e3
    if c > 0:
        rows.append((c, t, re.sub(r'\(anonymous namespace\)::', '', k)[:110]))
rows.sort(reverse=True)
with open(out + '/by_kernel.csv', 'w') as f:
    f.write('launches_per_clip,us_per_clip,kernel\n')
    for c, t, k in rows:
        f.write('%.1f,%.1f,"%s"\n' % (c, t, k))
tot = sum(r[0] for r in rows); tt = sum(r[1] for r in rows)
s = 'launches per clip: %.0f   kernel time per clip: %.2f ms\n' % (tot, tt / 1e3)
open(out + '/summary.txt', 'w').write(s)
print(s)
for c, t, k in rows[:45]:
    print('%7.1f %9.1f us  %s' % (c, t, k))
PY
