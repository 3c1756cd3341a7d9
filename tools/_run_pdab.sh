#!/bin/bash
# same-box A/B: probability-domain layout (default) against run words (VB2_PD=0)
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for pd in 1 0; do
    echo "== VB2_PD=$pd"
    export VB2_PD=$pd
    python bench.py --steps 1500 --warmup 200 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('  headline %.1f k evals/s  %.2f us' % (d['value']/1e3, d['ms_per_step']*1e3))"
    python tools/opt_time.py 2>&1 | grep "M="
    python tools/quality_profile_time.py 2>&1 | grep "codes" | head -2
    VB2_STEPS_ONLY=${STEPS_ONLY:-1} python tools/cohort_steps.py 2>&1 | grep samples
  done
done
