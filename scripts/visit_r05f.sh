#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_motion_cost.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -2
ARTP_KWALK=1 timeout 300 python -m pytest tests/test_motion_cost.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -2
echo "-- ksplit"; timeout 100 python scripts/cnn_bench.py 50 2>&1 | grep map
echo "-- kwalk (A row a step ahead)"; ARTP_KWALK=1 timeout 100 python scripts/cnn_bench.py 50 2>&1 | grep map
echo "-- kwalk variant 1"; ARTP_KWALK=1 ARTP_KWALK_VARIANT=1 timeout 100 python scripts/cnn_bench.py 50 2>&1 | grep map
echo "-- kwalk TR 8"; ARTP_KWALK=1 ARTP_KWALK_TR=8 timeout 100 python scripts/cnn_bench.py 50 2>&1 | grep map
