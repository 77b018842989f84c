#!/bin/bash
# Full profile of bench.py on the GPU box: kernel trace + stats, then PMC passes in SEPARATE runs
# (never combined with trace domains other than --kernel-trace).  Usage (via gpurun):
#   bash scripts/prof_run.sh <tag>     -> gpurun_out/prof_<tag>/summary.txt
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
CMDP="$CMD --skip-extras"   # PMC passes: only the 2^22-state launches, so max/dispatch = the bench launch
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY -d $OUT/pmc1 -o pmc1 -- $CMDP > $OUT/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_ANY -d $OUT/pmc2 -o pmc2 -- $CMDP > $OUT/pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc3 -o pmc3 -- $CMDP > $OUT/pmc3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc4 -o pmc4 -- $CMDP > $OUT/pmc4.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc5 -o pmc5 -- $CMDP > $OUT/pmc5.log 2>&1
python $GRAFT_REPO_ROOT/scripts/prof_summary.py $OUT $OUT/pmc_traffic.json > $OUT/summary.txt 2>&1
rm -f $OUT/*/*.db

