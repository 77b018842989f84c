#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
echo "== full gpu suite"; timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $OUT/pytest_gpu_r05d.log 2>&1; echo "rc $?"; tail -6 $OUT/pytest_gpu_r05d.log
echo "== fc A/B"; timeout 200 python scripts/fc_ab.py 2>&1 | grep -v amdgpu.ids
echo "== DVFS probe: CNN on a flat map vs the terrain"; timeout 100 python scripts/cnn_bench.py 50 2>&1 | grep map; ARTP_BENCH_FLAT=1 timeout 100 python scripts/cnn_bench.py 50 2>&1 | grep map
echo "   kwalk"; ARTP_KWALK=1 timeout 100 python scripts/cnn_bench.py 50 2>&1 | grep map; ARTP_KWALK=1 ARTP_BENCH_FLAT=1 timeout 100 python scripts/cnn_bench.py 50 2>&1 | grep map
echo "== stage timing"; ARTP_LIB=art_planner_amd/csrc/libartp_timing.so timeout 200 python scripts/stage_timing.py 2>&1 | grep -v amdgpu.ids
