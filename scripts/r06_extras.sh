#!/bin/bash
# Round-6 evidence that is not part of gpu_round.sh: the latency form of checkMotion (per-call times, kernel durations, the
# phase timestamps of one workgroup), the feature extractor's per-kernel trace, the LazyPRM* solve breakdown.
#   gpurun --timeout 1200 -- 'bash scripts/r06_extras.sh'
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06_extras
mkdir -p $OUT
{
  echo "== per-call latency of the host-buffer edge API (scripts/edge_latency.py: C2 map, YAML robot, edges between accepted states < 2 m apart)"
  python scripts/edge_latency.py 2>/dev/null
  echo
  echo "== kernel durations of check_motions_few_kernel by call size (rocprofv3 --kernel-trace, scripts/prof_few.sh)"
  bash scripts/prof_few.sh 2>/dev/null | grep "n= "
  echo
  echo "== phase timestamps of ONE workgroup (edge 0, chunk 1), timing build (scripts/few_trace.py; us)"
  make -s -C art_planner_amd/csrc timing > /dev/null 2>&1
  ARTP_LIB=art_planner_amd/csrc/libartp_timing.so python scripts/few_trace.py 2>/dev/null | grep -A1 "^edge [0-5] " | grep -v "^--"
} > $OUT/few_edges.txt 2>&1
{
  echo "== round trip of a request number: host -> polling workgroups -> host (tests/cpp/bar_pingpong_probe.hip)"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -w -o /tmp/bar_pingpong_probe tests/cpp/bar_pingpong_probe.hip 2>/dev/null && timeout 120 /tmp/bar_pingpong_probe 2>/dev/null
  echo
  echo "== checkMotion per call by edge length and verdict: one launch per call vs the resident pool (scripts/pool_slope.py)"
  python scripts/pool_slope.py 2>/dev/null
  echo
  echo "== the resident pool: phases of workgroup 1 for one request, timing build (scripts/pool_trace.py; us)"
  ARTP_LIB=art_planner_amd/csrc/libartp_timing.so python scripts/pool_trace.py 2>/dev/null | grep "^edge"
  echo
  echo "== pool size, variants build (scripts/edge_latency.py --few-only)"
  for w in 32 64 128; do echo "ARTP_POOL_WGS=$w"; ARTP_LIB=art_planner_amd/csrc/libartp_variants.so ARTP_POOL_WGS=$w python scripts/edge_latency.py --few-only 2>/dev/null | grep "^pool"; done
} > $OUT/edge_pool.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/cnn/trace -o trace -- python $GRAFT_REPO_ROOT/scripts/cnn_bench.py 50 > $OUT/cnn_bench.log 2>&1
{
  echo "== feature extractor, kernels only (scripts/cnn_bench.py 50, HIP events)"; grep "^map" $OUT/cnn_bench.log
  echo "== per kernel (rocprofv3 --kernel-trace --stats of the same command)"
  python $GRAFT_REPO_ROOT/scripts/prof_summary.py $OUT/cnn 2>&1 | grep -E "^==|^kernel|conv"
} > $OUT/cnn_kernel_trace.txt 2>&1
rm -rf $OUT/cnn
cd $GRAFT_REPO_ROOT
ARTP_SOLVE_TIMING=1 ARTP_LIB=art_planner_amd/csrc/libartp_variants.so python scripts/lazy_timing.py > $OUT/lazy_solve_breakdown.txt 2>&1
tail -4 $OUT/few_edges.txt; tail -12 $OUT/edge_pool.txt; cat $OUT/cnn_kernel_trace.txt | head -12; tail -3 $OUT/lazy_solve_breakdown.txt | cut -c1-300
