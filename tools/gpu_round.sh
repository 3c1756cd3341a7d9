#!/bin/bash
# One GPU round: parity smoke, bench at B=1/4/8 (A/B lane maps), kernel-trace profile.
# Usage (on the GPU box, from the repo root): bash tools/gpu_round.sh <tag>
TAG=${1:-r}
O=$GRAFT_REPO_ROOT/gpurun_out
python tools/gpu_first.py 2>&1 | grep -v abi_version | grep -E "rel err|deterministic|optimize|dalpha" | tail -12
for lm in ${LANEMAPS:-hw}; do for b in 1 4 8; do
  VB2_LANE_MAP=$lm python bench.py --batch $b --no-cpu-baseline --no-optimize 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$lm', 'B=%d'%r['config']['batch_points_per_step'], '%.0f evals/s'%r['value'], 'step %.1f us'%(1e3*r['ms_per_step']), 'dev %.1f us'%r['roofline']['device_us_per_launch'], 'frac %.3f'%r['roofline']['frac'], 'rel %.1e'%r.get('parity_probe_max_rel_err',-1))"
done; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof_$TAG -o p -- python $GRAFT_REPO_ROOT/tools/prof_eval.py > /dev/null 2>&1
