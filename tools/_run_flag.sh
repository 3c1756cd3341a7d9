#!/bin/bash
cd "$(dirname "$0")/.."
for rep in 1 2 3; do
  for v in base flagrel; do
    echo "== $v"
    export VB2_LIB_PATH=$PWD/build_variants/$v/libvb2.so
    python tools/single_point_time.py 2>&1 | grep "one point"
    VB2_STEPS_ONLY=1 python tools/cohort_steps.py 2>&1 | grep samples
  done
done
