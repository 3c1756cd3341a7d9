#!/bin/bash
for sl in 1 0; do for b in 4 8; do
  VB2_SINGLE_LAUNCH=$sl python bench.py --batch $b --no-cpu-baseline --no-optimize 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('single=$sl B=$b', 'dev %.1f us'%r['roofline']['device_us_per_launch'], 'rel %.1e'%r['parity_probe_max_rel_err'])"
done; done
python -m pytest tests -x -q -m gpu 2>&1 | tail -3
