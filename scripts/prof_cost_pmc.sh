#!/bin/bash
# PMC counters of the motion-cost kernels (separate passes, kernel trace only)
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_cost_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAVES -d $OUT/pmc1 -o pmc1 -- $CMD > $OUT/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_LDS -d $OUT/pmc2 -o pmc2 -- $CMD > $OUT/pmc2.log 2>&1
python $GRAFT_REPO_ROOT/scripts/prof_summary.py $OUT | grep -E "conv_lds|conv_ksplit" > $OUT/summary.txt
rm -f $OUT/*/*.db
cat $OUT/summary.txt
