#!/bin/bash
for a in 0 1 2 4 6 7; do
  VB2_ABLATE=$a python bench.py --batch 32 --no-cpu-baseline --no-optimize 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('B=32 ablate=$a', 'dev %.1f us'%r['roofline']['device_us_per_launch'])"
done
