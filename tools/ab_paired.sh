#!/bin/bash
# A/B of the wave shape for 4-point launches (MODE 3 paired micro-tiles vs MODE 1), same box.
for rep in 1 2; do for pm in 0 1; do
  us=$(VB2_PAIRED=$pm python bench.py --no-cpu-baseline --no-optimize --batch 4 2>/dev/null | tail -1 | python -c "import json,sys; print('%.2f' % json.loads(sys.stdin.read())['roofline']['device_us_per_launch'])")
  echo "rep $rep VB2_PAIRED=$pm B=4 $us us/launch; $(VB2_PAIRED=$pm python tools/opt_time.py 2>&1 | grep M=100000 | cut -c1-60); $(VB2_PAIRED=$pm python tools/cohort_time.py 2>&1 | tail -2 | tr '\n' ' ' | cut -c1-330)"
done; done
