#!/bin/bash
for b in 4 8 16 24 32 64; do
  python bench.py --batch $b --no-cpu-baseline --no-optimize 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('B=%d'%r['config']['batch_points_per_step'], '%.0f evals/s'%r['value'], 'step %.1f us'%(1e3*r['ms_per_step']), '%.2f us/eval'%(1e3*r['ms_per_step']/r['config']['batch_points_per_step']), 'frac %.3f'%r['roofline']['frac'], 'rel %.1e'%r.get('parity_probe_max_rel_err',-1))"
done
