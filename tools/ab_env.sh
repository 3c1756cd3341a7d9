#!/bin/bash
# usage: tools/ab_env.sh "<bench args>" ENVA ENVB   -- alternate two environments on one box
for rep in 1 2 3; do for e in "$2" "$3"; do
  us=$(env $e python bench.py --no-cpu-baseline --no-optimize $1 2>/dev/null | tail -1 | python -c "import json,sys; print('%.2f' % json.loads(sys.stdin.read())['roofline']['device_us_per_launch'])")
  echo "rep $rep [$e] $us us/launch"
done; done
