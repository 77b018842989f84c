#!/bin/bash
# conv1 o conv2 on the matrix cores vs the VALU form: parity tests, A/B timing, kernel trace.
#   gpurun --timeout 900 -- 'bash scripts/conv12_ab.sh <tag>'
# the switches below are read by the VARIANTS build only (make -C art_planner_amd/csrc variants)
export ARTP_LIB=${ARTP_LIB:-$GRAFT_REPO_ROOT/art_planner_amd/csrc/libartp_variants.so}
TAG=${1:-r05k}
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
echo "== parity"
timeout 500 python -m pytest tests/test_motion_cost.py tests/test_mfma_hazard.py -m gpu -q -p no:cacheprovider -x > $OUT/pytest_cnn_$TAG.log 2>&1
echo "pytest rc $?"; tail -6 $OUT/pytest_cnn_$TAG.log
echo "== timing"
for rep in 1 2; do
echo "-- default (conv12 MFMA)"; timeout 120 python scripts/cnn_bench.py 50
echo "-- ARTP_CONV12_FUSED=1"; ARTP_CONV12_FUSED=1 timeout 120 python scripts/cnn_bench.py 50
echo "-- ARTP_CONV12_FUSED=0 (conv12_mfma_kernel as a launch)"; ARTP_CONV12_FUSED=0 timeout 120 python scripts/cnn_bench.py 50
echo "-- ARTP_CONV12_MFMA=0"; ARTP_CONV12_MFMA=0 timeout 120 python scripts/cnn_bench.py 50
done
echo "== phase cycles (timing build)"
for f in 1 0; do echo "-- ARTP_CONV12_FUSED=$f"; ARTP_CONV12_FUSED=$f ARTP_LIB=art_planner_amd/csrc/libartp_timing.so timeout 120 python scripts/cnn_timing.py 2>&1 | head -30; done
echo "== kernel trace"
export TMPDIR=/tmp
rm -rf $OUT/prof_$TAG; mkdir -p $OUT/prof_$TAG
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/prof_$TAG/trace -o trace -- python $GRAFT_REPO_ROOT/scripts/cnn_bench.py 20 > $OUT/prof_$TAG/trace.log 2>&1)
python scripts/prof_summary.py $OUT/prof_$TAG > $OUT/prof_$TAG/summary.txt 2>&1
rm -f $OUT/prof_$TAG/*/*.db $OUT/prof_$TAG/*/*/*.db
head -8 $OUT/prof_$TAG/summary.txt
