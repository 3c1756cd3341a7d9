#!/bin/bash
python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for g in "16,1" "12,1" "8,2" "8,1"; do
  VB2_GEOM2=$g python bench.py --batch 8 --no-cpu-baseline --no-optimize 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('B=8 geom=$g', 'dev %.1f us'%r['roofline']['device_us_per_launch'])"
done
for g in "12,2" "12,1" "8,2" "6,2"; do
  VB2_GEOM1=$g python bench.py --batch 4 --no-cpu-baseline --no-optimize 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('B=4 geom=$g', 'dev %.1f us'%r['roofline']['device_us_per_launch'])"
done
VB2_B=8 python tools/stamps.py 2>&1 | tail -8
VB2_B=4 python tools/stamps.py 2>&1 | tail -8
