#!/bin/bash
cd "$(dirname "$0")/.."
for rows in 0 80 108 160; do
  echo "== VB2_PD_ROWS=$rows"
  export VB2_PD_ROWS=$rows
  python bench.py --steps 1500 --warmup 200 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('  headline %.1f k evals/s  %.2f us' % (d['value']/1e3, d['ms_per_step']*1e3))"
  python tools/opt_time.py 2>&1 | grep "M=100000"
done
