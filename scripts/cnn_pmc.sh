#!/bin/bash
# PMC counters of the feature-extractor kernels (separate passes, kernel trace only; never with other trace domains)
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_cnn_pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ[C]*_[A-Z0-9_]*" | sort -u > $OUT/sq_counters.txt
grep -i "icache\|ifetch\|inst_cache" $OUT/sq_counters.txt | tr '\n' ' '; echo
CMD="python $GRAFT_REPO_ROOT/scripts/cnn_bench.py 10"
P1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU"
P2="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_MFMA"
P3="$(grep -i "SQ_IFETCH$\|SQ_IFETCH_LEVEL$\|SQC_ICACHE_REQ$\|SQC_ICACHE_HITS$\|SQC_ICACHE_MISSES$\|SQC_ICACHE_MISSES_DUPLICATE$" $OUT/sq_counters.txt | tr '\n' ' ')"
i=1
for P in "$P1" "$P2" "$P3"; do
  [ -z "$P" ] && continue
  timeout 200 rocprofv3 --kernel-trace --pmc $P -d $OUT/pmc$i -o pmc$i -- $CMD > $OUT/pmc$i.log 2>&1 || tail -3 $OUT/pmc$i.log
  i=$((i+1))
done
python - > $OUT/summary.txt <<PY
import glob, sqlite3
print("rocprofv3 --kernel-trace --pmc <pass> -- python scripts/cnn_bench.py 10  (feature extractor at 400^2 and 800^2, 13 launches each;")
print("avg over both sizes, max = the 800^2 launches).  MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8).")
for db in sorted(glob.glob("$OUT/**/*_results.db", recursive=True)):
    c = sqlite3.connect(db)
    try:
        rows = c.execute("select kernel_name, counter_name, avg(value), max(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall()
    except Exception as e:
        print(db, e); continue
    for kn, cn, avg, mx, n in rows:
        if any(k in kn for k in ("conv345", "conv_ksplit", "conv12")):
            print(kn.split("(")[0][-40:], cn, f"avg {avg:.4g} max {mx:.4g} n {n}")
PY
cat $OUT/summary.txt
rm -f $OUT/*/*.db $OUT/*/*/*.db
