#!/bin/bash
# kernel trace only (no PMC) of the headline loop: per-kernel durations of the 2^22-state launches
TAG=${1:-t}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --skip-extras > $OUT/trace.log 2>&1
python $GRAFT_REPO_ROOT/scripts/prof_summary.py $OUT > $OUT/summary.txt 2>&1
rm -f $OUT/*/*.db
head -16 $OUT/summary.txt
