#!/bin/bash
# kernel trace + one PMC pass of bench.py (see prof_run.sh for the full set)
TAG=${1:-q}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY -d $OUT/pmc1 -o pmc1 -- $CMD > $OUT/pmc1.log 2>&1
python $GRAFT_REPO_ROOT/scripts/prof_summary.py $OUT > $OUT/summary.txt 2>&1
rm -f $OUT/*/*.db
