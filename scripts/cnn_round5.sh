#!/bin/bash
# Round-5 GPU visit for the feature extractor: MFMA hazard probe, parity tests, timing of the variants, phase cycles,
# kernel trace.    gpurun --timeout 900 -- 'bash scripts/cnn_round5.sh <tag>'
# the switches below are read by the VARIANTS build only (make -C art_planner_amd/csrc variants)
export ARTP_LIB=${ARTP_LIB:-$GRAFT_REPO_ROOT/art_planner_amd/csrc/libartp_variants.so}
TAG=${1:-r05a}
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
CS=art_planner_amd/csrc
echo "== hazard probe"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -std=c++17 -o /tmp/mfma_probe tests/cpp/mfma_hazard_probe.hip && timeout 120 /tmp/mfma_probe | tee $OUT/mfma_hazard_probe_$TAG.txt
echo "== parity (default = kwalk, guarded)"
timeout 400 python -m pytest tests/test_motion_cost.py -m gpu -q -p no:cacheprovider -x > $OUT/pytest_cnn_$TAG.log 2>&1
echo "pytest rc $?"; tail -4 $OUT/pytest_cnn_$TAG.log
echo "== timing"
echo "-- default (kwalk variant 0: BD 5, rows by LDS-DMA at tile start; XCD-aware tiles)"; timeout 120 python scripts/cnn_bench.py 50
for v in 1 2; do echo "-- ARTP_KWALK_VARIANT=$v"; ARTP_KWALK_VARIANT=$v timeout 120 python scripts/cnn_bench.py 30; done
echo "-- ARTP_KWALK=0 (round-4 15x15 kernel, XCD-aware tiles)"; ARTP_KWALK=0 timeout 120 python scripts/cnn_bench.py 50
echo "-- ARTP_KWALK=0 ARTP_CNN_XCD=0 (round-4 15x15 kernel and conv345 in launch order)"; ARTP_KWALK=0 ARTP_CNN_XCD=0 timeout 120 python scripts/cnn_bench.py 50
echo "-- ARTP_CNN_XCD=0 (kwalk, conv345 in launch order)"; ARTP_CNN_XCD=0 timeout 120 python scripts/cnn_bench.py 30
for tr in 8 9; do echo "-- ARTP_KWALK_TR=$tr"; ARTP_KWALK_TR=$tr timeout 120 python scripts/cnn_bench.py 30; done
echo "== phase cycles (timing build)"
[ -f $CS/libartp_timing.so ] && ARTP_LIB=$CS/libartp_timing.so timeout 120 python scripts/cnn_timing.py
echo "== kernel trace"
export TMPDIR=/tmp
rm -rf $OUT/prof_$TAG; mkdir -p $OUT/prof_$TAG
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/prof_$TAG/trace -o trace -- python $GRAFT_REPO_ROOT/scripts/cnn_bench.py 20 > $OUT/prof_$TAG/trace.log 2>&1)
python scripts/prof_summary.py $OUT/prof_$TAG > $OUT/prof_$TAG/summary.txt 2>&1
rm -f $OUT/prof_$TAG/*/*.db $OUT/prof_$TAG/*/*/*.db
head -8 $OUT/prof_$TAG/summary.txt
if [ "$2" = "pmc" ]; then
echo "== PMC (default)"; bash scripts/cnn_pmc.sh > $OUT/cnn_pmc_$TAG.txt 2>&1; grep -E 'MFMA_BUSY|GRBM_GUI|WAIT_INST_ANY|SQ_WAIT_ANY|WAVE_CYCLES|ACTIVE_INST_ANY|INSTS_MFMA' $OUT/cnn_pmc_$TAG.txt | grep -v conv12
echo "== PMC (ARTP_KWALK=0)"; ARTP_KWALK=0 bash scripts/cnn_pmc.sh > $OUT/cnn_pmc_${TAG}_ksplit.txt 2>&1; grep -E 'MFMA_BUSY|GRBM_GUI|WAIT_INST_ANY|SQ_WAIT_ANY|WAVE_CYCLES|ACTIVE_INST_ANY|INSTS_MFMA' $OUT/cnn_pmc_${TAG}_ksplit.txt | grep -v conv12
fi
