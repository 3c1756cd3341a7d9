#!/bin/bash
# same-box A/B of whole-library builds: build_variants/<name>/libvb2.so for name in $VARIANTS
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for v in $VARIANTS; do
    echo "== $v"
    export VB2_LIB_PATH=$PWD/build_variants/$v/libvb2.so
    python bench.py --steps 1500 --warmup 200 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('  headline %.1f k evals/s  %.2f us' % (d['value']/1e3, d['ms_per_step']*1e3))"
    python tools/opt_time.py 2>&1 | grep "M="
    python tools/quality_profile_time.py 2>&1 | grep "codes" | head -2
    VB2_STEPS_ONLY=1 python tools/cohort_steps.py 2>&1 | grep samples
  done
done
