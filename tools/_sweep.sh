python bench.py --no-cpu-baseline --no-em --no-legs --save-plans /tmp/p.json > /tmp/b0.json 2>/dev/null
for cfg in "--seqs 4 --lookahead 4" "--seqs 4 --lookahead 8" "--seqs 6 --lookahead 4" "--seqs 3 --lookahead 4" "--seqs 2 --lookahead 4" "--seqs 1 --lookahead 4" "--seqs 1 --lookahead 8" "--seqs 1 --lookahead 2"; do
  python bench.py --no-cpu-baseline --no-em --no-legs $cfg > /tmp/b.json 2>/dev/null
  python -c "
import json,sys
d=json.loads([l for l in open('/tmp/b.json') if l.startswith('{')][-1]); print('$cfg', d['value'])"
done
