#!/bin/bash
# Regenerates the numbers kept under profiles/<round>/ (run on the GPU box from the repo root):
#   bench line, rocprofv3 kernel-trace stats of the SAME command, PMC passes (FETCH_SIZE,
#   WRITE_SIZE, SQ counters, GRBM_GUI_ACTIVE; each in its own run, with --kernel-trace only),
#   kernel stats of the other wave shapes (4-point / 1-point launches, cohort steps) and of a search.
# Usage: bash tools/collect_profiles.sh r02
R=${1:-r06}
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
# FP64 vector instructions by class (the flop roofline of bench.py: roofline.fp64): adds, multiplies, FMAs, transcendentals
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 --output-format csv -d $O/pmc_f64 -o b -- $B $P > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_modes -o m -- python $GRAFT_REPO_ROOT/tools/prof_modes.py > $O/trace_modes.log 2>&1
# HBM-side bytes of the cohort steps (32 samples; 1, 2, 4 and 8 points per sample): FETCH_SIZE of llk_eval_multi_kernel<*>
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_modes_fetch -o m -- python $GRAFT_REPO_ROOT/tools/prof_modes.py > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_opt -o o -- python $GRAFT_REPO_ROOT/tools/opt_time.py > $O/trace_opt.log 2>&1
# the cohort steps of a search (1 and 2 points per sample: llk_eval_multi_kernel<4,...>, <5,...>): kernel stats, SQ passes, FETCH_SIZE
C="python $GRAFT_REPO_ROOT/tools/prof_cohort.py"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_cohort -o c -- $C > $O/trace_cohort.log 2>&1
VB2_STEPS=40 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc_cohort_sq1 -o c -- $C > /dev/null 2>&1
VB2_STEPS=40 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM --output-format csv -d $O/pmc_cohort_sq2 -o c -- $C > /dev/null 2>&1
VB2_STEPS=40 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_cohort_fetch -o c -- $C > /dev/null 2>&1
VB2_STEPS=40 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_cohort_grbm -o c -- $C > /dev/null 2>&1
# the headline launch on wider quality alphabets (2..60 = BAQ-like, 118 codes: bench.py -> roofline_wide_alphabet; 10..45, 72 codes:
# roofline_mid_alphabet): kernel stats + the SQ / GRBM passes of each
for A in "wide 2 60" "mid 10 45"; do
  set -- $A
  W="$B --q-lo $2 --q-hi $3"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$1 -o w -- $W > $O/trace_$1.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc_$1_sq1 -o w -- $W $P > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_$1_grbm -o w -- $W $P > /dev/null 2>&1
done
# vb2_ctx_create (the flatten on the device: classify_kernel, pack_layout_kernel, pack_sched_kernel): host times and kernel stats
cd /tmp
python $GRAFT_REPO_ROOT/tools/create_time.py > $O/create_time.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_create -o cr -- python $GRAFT_REPO_ROOT/tools/create_time.py > /dev/null 2>&1
# the issue ceilings bench.py quotes the kernels against, stand-alone: FP64 FMA alone / LDS reads alone / both, by waves per CU
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off $GRAFT_REPO_ROOT/tools/ubench/lds_fma_mix.hip -o /tmp/lds_fma_mix 2>/dev/null && /tmp/lds_fma_mix > $O/ubench_lds_fma_mix.txt 2>&1
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off $GRAFT_REPO_ROOT/tools/ubench/valu_rates.hip -o /tmp/valu_rates 2>/dev/null && /tmp/valu_rates > $O/ubench_valu_rates.txt 2>&1
# a search round's timeline (in-kernel stamps; the build with the stamps frozen at round 200 if it was made: make stamps_round)
cd $GRAFT_REPO_ROOT
python tools/stamps_resident.py > $O/search_round_stamps.txt 2>&1
if [ -f verifybamid_amd/libvb2_stamps_r.so ]; then VB2_STAMPS_ROUND=1 VB2_STAMPS_DETAIL=1 python tools/stamps_resident.py > $O/search_round200_stamps.txt 2>&1; fi
# the big raw tables stay on the box: only the per-kernel summaries travel (gpurun_out is capped at 64 MiB)
for d in $O/pmc_* $O/trace*; do [ -d $d ] && find $d -name "*kernel_trace.csv" -size +2M -delete; done
find $O -name "*.db" -delete
ls $O
