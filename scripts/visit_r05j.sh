#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
one() { timeout 300 python bench.py --steps 20 --warmup 5 --skip-extras --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms/step %.4f' % d['ms_per_step'], 'label hash', d['label_hash_batch0'], 'kernel_ms %.4f' % d['roofline']['kernel_ms'])"; }
echo "== dense, 6 waves/SIMD"; one
echo "== dense, 5 waves/SIMD"; ARTP_LIB=art_planner_amd/csrc/libartp_f5.so one
echo "== dense, 4 waves/SIMD"; ARTP_LIB=art_planner_amd/csrc/libartp_f4.so one
echo "== ARTP_FEET_DENSE=0"; ARTP_FEET_DENSE=0 one
export TMPDIR=/tmp
for V in 6 5 4; do
rm -rf $OUT/prof_fs_$V; mkdir -p $OUT/prof_fs_$V
L=art_planner_amd/csrc/libartp.so; [ $V != 6 ] && L=art_planner_amd/csrc/libartp_f$V.so
(cd /tmp && ARTP_LIB=$GRAFT_REPO_ROOT/$L timeout 300 rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS GRBM_GUI_ACTIVE -d $OUT/prof_fs_$V -o lane -- python $GRAFT_REPO_ROOT/bench.py --pmc-child states > $OUT/prof_fs_$V/log.txt 2>&1)
python - <<PY
import glob, sys
sys.path.insert(0, "$GRAFT_REPO_ROOT")
import bench
for db in glob.glob("$OUT/prof_fs_$V/**/*_results.db", recursive=True):
    for kn, v in sorted(bench._read_pass(db).items(), key=lambda kv: -kv[1].get("max_us", 0)):
        if "SQ_INSTS_VALU" in v and "feet_stream" in kn:
            cyc = v["GRBM_GUI_ACTIVE"] / 8
            print("waves=$V %-36s us %7.1f insts_valu %.4g lane util %.3f valu_busy %.3f wait %.3f lds insts %.4g" % (kn.split("(")[0][-36:], v["max_us"], v["SQ_INSTS_VALU"], v["SQ_THREAD_CYCLES_VALU"] / (64.0 * v["SQ_INSTS_VALU"]), v["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * cyc), v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], v["SQ_INSTS_LDS"]))
PY
rm -rf $OUT/prof_fs_$V
done
