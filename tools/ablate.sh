#!/bin/bash
# Phase ablation of llk_eval_kernel (VB2_ABLATE bits: 1 table math, 2 read loop, 4 epilogue math)
for M in 100000 10000; do for b in 4 8; do for a in 0 1 2 4 6 7; do
  VB2_ABLATE=$a python bench.py --markers $M --batch $b --no-cpu-baseline --no-optimize 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('M=$M B=$b ablate=$a', 'dev %.1f us'%r['roofline']['device_us_per_launch'])"
done; done; done
