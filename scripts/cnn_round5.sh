#!/bin/bash
# Round-5 GPU visit for the feature extractor: MFMA hazard probe, parity tests, timing of the variants, phase cycles,
# kernel trace.    gpurun --timeout 900 -- 'bash scripts/cnn_round5.sh <tag>'
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
if [ -f $CS/libartp_noguard.so ]; then
  echo "== parity (kwalk, NO guard)"
  ARTP_LIB=$CS/libartp_noguard.so timeout 400 python -m pytest tests/test_motion_cost.py -m gpu -q -p no:cacheprovider -x > $OUT/pytest_cnn_${TAG}_noguard.log 2>&1
  echo "pytest rc $?"; tail -4 $OUT/pytest_cnn_${TAG}_noguard.log
fi
echo "== timing"
echo "-- default"; timeout 120 python scripts/cnn_bench.py 50
[ -f $CS/libartp_noguard.so ] && { echo "-- no guard"; ARTP_LIB=$CS/libartp_noguard.so timeout 120 python scripts/cnn_bench.py 50; }
echo "-- ARTP_KWALK=0 (round-4 15x15 kernel, new conv345)"; ARTP_KWALK=0 timeout 120 python scripts/cnn_bench.py 50
for tr in 8 9 10; do echo "-- ARTP_KWALK_TR=$tr"; ARTP_KWALK_TR=$tr timeout 120 python scripts/cnn_bench.py 30; done
for t in 12 16 18; do echo "-- ARTP_C345_T=$t"; ARTP_C345_T=$t timeout 120 python scripts/cnn_bench.py 30; done
echo "== phase cycles (timing build)"
[ -f $CS/libartp_timing.so ] && ARTP_LIB=$CS/libartp_timing.so timeout 120 python scripts/cnn_timing.py
echo "== kernel trace"
export TMPDIR=/tmp
rm -rf $OUT/prof_$TAG; mkdir -p $OUT/prof_$TAG
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/prof_$TAG/trace -o trace -- python $GRAFT_REPO_ROOT/scripts/cnn_bench.py 20 > $OUT/prof_$TAG/trace.log 2>&1)
python scripts/prof_summary.py $OUT/prof_$TAG > $OUT/prof_$TAG/summary.txt 2>&1
rm -f $OUT/prof_$TAG/*/*.db $OUT/prof_$TAG/*/*/*.db
head -8 $OUT/prof_$TAG/summary.txt
echo "== lane utilisation of the validity pipeline (PMC pass: SQ_THREAD_CYCLES_VALU)"
rm -rf $OUT/prof_lane_$TAG; mkdir -p $OUT/prof_lane_$TAG
for MODE in states check_motion; do
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $OUT/prof_lane_$TAG/$MODE -o lane -- python $GRAFT_REPO_ROOT/bench.py --pmc-child $MODE > $OUT/prof_lane_$TAG/$MODE.log 2>&1)
python - <<PY
import glob, sys
sys.path.insert(0, "$GRAFT_REPO_ROOT")
import bench
for db in glob.glob("$OUT/prof_lane_$TAG/$MODE/**/*_results.db", recursive=True):
    for kn, v in sorted(bench._read_pass(db).items(), key=lambda kv: -kv[1].get("max_us", 0)):
        if "SQ_ACTIVE_INST_VALU" in v and v.get("max_us", 0) > 3:
            a, t, n = v["SQ_ACTIVE_INST_VALU"], v.get("SQ_THREAD_CYCLES_VALU", 0), v.get("SQ_INSTS_VALU", 0)
            print("$MODE %-44s us %8.1f  ACTIVE_INST_VALU %.4g THREAD_CYCLES_VALU %.4g INSTS_VALU %.4g  thread/(64*active*4) %.3f  thread/(64*insts) %.3f"
                  % (kn.split("(")[0][-44:], v["max_us"], a, t, n, t / (256.0 * a) if a else 0, t / (64.0 * n) if n else 0))
PY
done
rm -f $OUT/prof_lane_$TAG/*/*.db $OUT/prof_lane_$TAG/*/*/*.db $OUT/prof_lane_$TAG/*/*/*/*.db
