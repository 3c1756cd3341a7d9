#!/bin/bash
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for v in $VARIANTS; do
    export VB2_LIB_PATH=$PWD/build_variants/$v/libvb2.so
    echo "$v: $(python bench.py --steps 1500 --warmup 200 --no-extras --no-cpu-baseline --no-optimize 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.2f us' % (d['ms_per_step']*1e3))")"
  done
done
