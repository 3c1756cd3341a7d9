#!/bin/bash
for g in "16,1" "12,1" "10,2" "8,2"; do
  VB2_FORCE_BTL1=1 VB2_GEOM1=$g python bench.py --batch 16 --no-cpu-baseline --no-optimize 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('BTL1 geom=$g B=16', 'step %.1f us'%(1e3*r['ms_per_step']), '%.2f us/eval'%(1e3*r['ms_per_step']/16), 'rel %.1e'%r.get('parity_probe_max_rel_err',-1))"
done
python bench.py --batch 16 --no-cpu-baseline --no-optimize 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('BTL2 B=16', 'step %.1f us'%(1e3*r['ms_per_step']))"
