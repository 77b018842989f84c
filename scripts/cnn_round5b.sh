#!/bin/bash
# Round-5 evidence for the feature extractor's final form: conv1 o conv2 A/B (fused / own launch / VALU), phase cycles, the 15 x 15
# launch as a Gantt chart, what the matrix cores sustain, kernel trace, MFMA PMC.   gpurun --timeout 900 -- 'bash scripts/cnn_round5b.sh'
# the switches below are read by the VARIANTS build only (make -C art_planner_amd/csrc variants)
export ARTP_LIB=${ARTP_LIB:-$GRAFT_REPO_ROOT/art_planner_amd/csrc/libartp_variants.so}
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
CS=art_planner_amd/csrc
{
echo "== what the matrix cores sustain (tests/cpp/mfma_clock_probe.hip)"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma_clock tests/cpp/mfma_clock_probe.hip && timeout 60 /tmp/mfma_clock
} > $OUT/r05_mfma_clock_probe.txt 2>&1
{
echo "== conv1 o conv2: inside conv345's patch phase (default) / as a launch of its own / the VALU form; us per feature map at 400^2, 800^2 (scripts/cnn_bench.py 50)"
for rep in 1 2 3; do
echo "-- default (fused)"; timeout 120 python scripts/cnn_bench.py 50
echo "-- ARTP_CONV12_FUSED=0 (conv12_mfma_kernel)"; ARTP_CONV12_FUSED=0 timeout 120 python scripts/cnn_bench.py 50
echo "-- ARTP_CONV12_MFMA=0 (conv12_pool_kernel, rounds 3-4)"; ARTP_CONV12_MFMA=0 timeout 120 python scripts/cnn_bench.py 50
done
echo "== phase cycles (timing build, stamps held in registers)"
for f in 1 0; do echo "-- ARTP_CONV12_FUSED=$f"; ARTP_CONV12_FUSED=$f ARTP_LIB=$CS/libartp_timing.so timeout 120 python scripts/cnn_timing.py 2>&1; done
} > $OUT/r05_conv12_ab.txt 2>&1
{
echo "== the 15 x 15 layer's launch, per workgroup (scripts/ksplit_gantt.py, timing build)"
ARTP_LIB=$CS/libartp_timing.so timeout 120 python scripts/ksplit_gantt.py
echo "== one workgroup per CU (ARTP_KSPLIT_ONE_PER_CU=1)"
ARTP_KSPLIT_ONE_PER_CU=1 ARTP_LIB=$CS/libartp_timing.so timeout 120 python scripts/ksplit_gantt.py 800
} > $OUT/r05_ksplit_gantt.txt 2>&1
export TMPDIR=/tmp
rm -rf $OUT/prof_cnn5b; mkdir -p $OUT/prof_cnn5b
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/prof_cnn5b/trace -o trace -- python $GRAFT_REPO_ROOT/scripts/cnn_bench.py 20 > $OUT/prof_cnn5b/trace.log 2>&1)
python scripts/prof_summary.py $OUT/prof_cnn5b > $OUT/r05_cnn_kernel_trace.txt 2>&1
rm -f $OUT/prof_cnn5b/*/*.db $OUT/prof_cnn5b/*/*/*.db
bash scripts/cnn_pmc.sh > $OUT/r05_cnn_pmc.txt 2>&1
tail -5 $OUT/r05_mfma_clock_probe.txt; grep -A2 'default' $OUT/r05_conv12_ab.txt | head -12; head -8 $OUT/r05_cnn_kernel_trace.txt; grep -E 'MFMA busy|clock' $OUT/r05_cnn_pmc.txt | head
