#!/bin/bash
# PMC passes of the headline launch, probability-domain layout against run words (VB2_PD=0)
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_pd
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-optimize --no-extras --soft-exit --steps 50 --warmup 20 --prewarm-ms 0"
for pd in 1 0; do
  export VB2_PD=$pd
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU --output-format csv -d $O/sq1_$pd -o b -- $B > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM --output-format csv -d $O/sq2_$pd -o b -- $B > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/grbm_$pd -o b -- $B > /dev/null 2>&1
  echo "=== VB2_PD=$pd"
  for d in sq1 sq2 grbm; do python $GRAFT_REPO_ROOT/tools/pmc_summary.py $(find $O/${d}_$pd -name "*counter_collection.csv" | head -1) | grep -A 12 "llk_eval_kernel"; done
done
find $O -name "*.csv" -size +1M -delete; find $O -name "*.db" -delete
