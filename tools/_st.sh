cd $GRAFT_REPO_ROOT
export VB2_S=256 VB2_DISTINCT=8 VB2_GROUPS=32,32,32,32
for t in 10 12 14 15 16; do
VB2_THREADS=$t timeout 600 python tools/cohort_run_time.py 2>&1 | grep -E "host threads" | awk '{print $3, $4, $5, $13, $14}' | tr '\n' ' '; echo
done
