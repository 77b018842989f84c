#!/bin/bash
# PMC counters of the sampler kernel alone (two passes; never combined with other trace domains)
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_sampler
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --skip-extras"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $OUT/pmc1 -o pmc1 -- $CMD > $OUT/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS GRBM_GUI_ACTIVE SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 -d $OUT/pmc2 -o pmc2 -- $CMD > $OUT/pmc2.log 2>&1
python $GRAFT_REPO_ROOT/scripts/prof_summary.py $OUT > $OUT/summary.txt 2>&1
rm -f $OUT/*/*.db
grep "sample_states" $OUT/summary.txt | cut -c1-150
