#!/bin/bash
# A/B of the cross-workgroup hand-off protocols on one box (alternating runs).
for rep in 1 2 3; do
  for mode in ticket tagged; do
    for B in 4 8 48; do
      us=$(VB2_REDUCE=$mode python bench.py --no-cpu-baseline --no-optimize --batch $B 2>/dev/null | tail -1 | python -c "import json,sys; print('%.2f' % json.loads(sys.stdin.read())['roofline']['device_us_per_launch'])")
      echo "rep $rep $mode B=$B $us us"
    done
  done
done
