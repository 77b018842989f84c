#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
echo "== PMC ksplit (default)"; bash scripts/cnn_pmc.sh > $OUT/cnn_pmc_r05e_ksplit.txt 2>&1; grep '^==' $OUT/cnn_pmc_r05e_ksplit.txt
echo "== PMC kwalk"; ARTP_KWALK=1 bash scripts/cnn_pmc.sh > $OUT/cnn_pmc_r05e_kwalk.txt 2>&1; grep '^==' $OUT/cnn_pmc_r05e_kwalk.txt
echo "== stage timing"; ARTP_LIB=art_planner_amd/csrc/libartp_timing.so timeout 200 python scripts/stage_timing.py 2>&1 | grep -v amdgpu.ids
