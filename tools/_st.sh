cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.txt 2>&1; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/pytest_gpu.txt | tail -4 | cut -c1-250
for st in 0 1; do
VB2_COHORT_STREAM=$st python bench.py --no-cpu-baseline --no-optimize --steps 200 2>/dev/null | tail -1 | python -c "
import sys,json
c=json.loads(sys.stdin.read())['cohort']; print('stream $st bench from_text', c['from_text']['samples_per_s'], 'search only', c['samples_per_s_search_only'])"
done
VB2_CPUS=8 python bench.py --no-cpu-baseline --no-optimize --steps 200 2>/dev/null | tail -1 | python -c "
import sys,json
c=json.loads(sys.stdin.read())['cohort']; print('VB2_CPUS=8 bench from_text', c['from_text']['samples_per_s'])"
