#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
echo "== parity"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_host_mirror.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -4
echo "== bench (dense feet)"; timeout 300 python bench.py --steps 20 --warmup 5 --skip-extras --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms/step', d['ms_per_step'], 'label hash', d['label_hash_batch0'], 'kernel_ms', d['roofline']['kernel_ms'])"
echo "== bench (ARTP_FEET_DENSE=0)"; ARTP_FEET_DENSE=0 timeout 300 python bench.py --steps 20 --warmup 5 --skip-extras --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms/step', d['ms_per_step'], 'label hash', d['label_hash_batch0'], 'kernel_ms', d['roofline']['kernel_ms'])"
echo "== kernel trace + lane util"
export TMPDIR=/tmp
for D in 1 0; do
rm -rf $OUT/prof_fs_$D; mkdir -p $OUT/prof_fs_$D
(cd /tmp && ARTP_FEET_DENSE=$D timeout 300 rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $OUT/prof_fs_$D -o lane -- python $GRAFT_REPO_ROOT/bench.py --pmc-child states > $OUT/prof_fs_$D/log.txt 2>&1)
python - <<PY
import glob, sys
sys.path.insert(0, "$GRAFT_REPO_ROOT")
import bench
for db in glob.glob("$OUT/prof_fs_$D/**/*_results.db", recursive=True):
    for kn, v in sorted(bench._read_pass(db).items(), key=lambda kv: -kv[1].get("max_us", 0)):
        if "SQ_INSTS_VALU" in v and v.get("max_us", 0) > 30:
            print("dense=$D %-40s us %7.1f  insts_valu %.4g  lane util %.3f" % (kn.split("(")[0][-40:], v["max_us"], v["SQ_INSTS_VALU"], v["SQ_THREAD_CYCLES_VALU"] / (64.0 * v["SQ_INSTS_VALU"])))
PY
rm -f $OUT/prof_fs_$D/*/*.db $OUT/prof_fs_$D/*/*/*.db
done
