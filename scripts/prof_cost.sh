#!/bin/bash
# kernel trace of bench.py restricted to the motion-cost kernels (conv / pool / fc)
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_cost
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench.log 2>&1
python $GRAFT_REPO_ROOT/scripts/prof_summary.py $OUT | grep -E "conv|maxpool|f32_to|fc_cost" > $OUT/summary.txt
rm -f $OUT/*/*.db
cat $OUT/summary.txt
grep '^{' $OUT/bench.log | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['motion_cost_c3'])"
