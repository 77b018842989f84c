#!/bin/bash
# One GPU-box visit for the feature extractor: phase cycles (timing build), parity tests, timing (fused vs round-2
# path), per-kernel trace.   gpurun --timeout 600 -- 'bash scripts/cnn_round.sh <tag>'
TAG=${1:-cnn}
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
[ -f art_planner_amd/csrc/libartp_timing.so ] && ARTP_LIB=art_planner_amd/csrc/libartp_timing.so timeout 120 python scripts/cnn_timing.py
timeout 300 python -m pytest tests/test_motion_cost.py -m gpu -q -p no:cacheprovider -x > $OUT/pytest_cnn_$TAG.log 2>&1
echo "pytest rc $?"; tail -5 $OUT/pytest_cnn_$TAG.log
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/motion_cost_err_*.json")):
    for case, d in json.load(open(f)).items():
        print(case, {k: f"{v[0]:.2e}/{v[1]:.2e}" for k, v in d.items()})
PY
timeout 120 python scripts/cnn_bench.py 50
[ "$2" = "ab" ] && ARTP_CNN_UNFUSED=1 timeout 120 python scripts/cnn_bench.py 50
export TMPDIR=/tmp
rm -rf $OUT/prof_$TAG; mkdir -p $OUT/prof_$TAG
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/prof_$TAG/trace -o trace -- python $GRAFT_REPO_ROOT/scripts/cnn_bench.py 20 > $OUT/prof_$TAG/trace.log 2>&1)
python scripts/prof_summary.py $OUT/prof_$TAG > $OUT/prof_$TAG/summary.txt 2>&1
rm -f $OUT/prof_$TAG/*/*.db $OUT/prof_$TAG/*/*/*.db
head -8 $OUT/prof_$TAG/summary.txt
