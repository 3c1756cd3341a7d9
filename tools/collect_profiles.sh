#!/bin/bash
# Regenerates the numbers kept under profiles/<round>/ (run on the GPU box from the repo root):
#   bench line, rocprofv3 kernel-trace stats of the SAME command, PMC passes (FETCH_SIZE,
#   WRITE_SIZE, SQ counters, GRBM_GUI_ACTIVE; each in its own run, with --kernel-trace only),
#   kernel stats of the other wave shapes (4-point / 1-point launches, cohort steps) and of a search.
# Usage: bash tools/collect_profiles.sh r02
R=${1:-r03}
O=$GRAFT_REPO_ROOT/gpurun_out/$R
mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
tail -1 $O/bench.json | head -c 400; echo
rm -f $O/bench_batch_sweep.jsonl
for b in 1 4 8 16 32 48; do python bench.py --batch $b --no-cpu-baseline --no-optimize --no-extras 2>/dev/null | tail -1 >> $O/bench_batch_sweep.jsonl; done
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-optimize --no-extras --soft-exit"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o b -- $B > $O/trace.log 2>&1
P="--steps 50 --warmup 20 --prewarm-ms 0"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o b -- $B $P > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o b -- $B $P > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc_sq1 -o b -- $B $P > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM --output-format csv -d $O/pmc_sq2 -o b -- $B $P > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_grbm -o b -- $B $P > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_modes -o m -- python $GRAFT_REPO_ROOT/tools/prof_modes.py > $O/trace_modes.log 2>&1
# HBM-side bytes of the cohort steps (32 samples; 1, 2, 4 and 8 points per sample): FETCH_SIZE of llk_eval_multi_kernel<*>
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_modes_fetch -o m -- python $GRAFT_REPO_ROOT/tools/prof_modes.py > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_opt -o o -- python $GRAFT_REPO_ROOT/tools/opt_time.py > $O/trace_opt.log 2>&1
ls $O
