#!/bin/bash
# A/B of the 15 x 15 layer: conv15_pair32_kernel (default above 256 tiles) against conv_ksplit_kernel, + per-kernel trace
# the switches below are read by the VARIANTS build only (make -C art_planner_amd/csrc variants)
export ARTP_LIB=${ARTP_LIB:-$GRAFT_REPO_ROOT/art_planner_amd/csrc/libartp_variants.so}
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python -m pytest tests/test_motion_cost.py -m gpu -q -x -p no:cacheprovider -k "c3_c4 or anchor or reference" 2>&1 | tail -5
for v in 1 0 1 0; do echo "ARTP_CONV15_PAIR32=$v"; ARTP_CONV15_PAIR32=$v python scripts/cnn_bench.py 50 2>&1 | grep -v "^$" | tail -3; done
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  rocprofv3 --kernel-trace --stats -d $OUT/prof_cnn32/v$v -o trace -- env ARTP_CONV15_PAIR32=$v python $GRAFT_REPO_ROOT/scripts/cnn_bench.py 30 > /dev/null 2>&1
done
python $GRAFT_REPO_ROOT/scripts/prof_summary.py $OUT/prof_cnn32 2>&1 | grep -E "^==|conv|kernel " | head -30
rm -f $OUT/prof_cnn32/*/*.db $OUT/prof_cnn32/*/*/*.db
