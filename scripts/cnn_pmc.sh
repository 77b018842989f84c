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
print("rocprofv3 --kernel-trace --pmc <pass> -- python scripts/cnn_bench.py 10  (feature extractor at 400^2 and 800^2, 13 launches each).")
print("Per kernel NAME (the 400^2 and the 800^2 launch are different template instances): avg over its launches.")
print("MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8); clock = GRBM_GUI_ACTIVE / 8 / duration of the same pass.")
for db in sorted(glob.glob("$OUT/**/*_results.db", recursive=True)):
    c = sqlite3.connect(db)
    try:
        rows = c.execute("select kernel_name, counter_name, avg(value), max(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall()
        durs = dict(c.execute("select name, avg(end-start) from kernels group by name").fetchall())
    except Exception as e:
        print(db, e); continue
    per = {}
    for kn, cn, avg, mx, n in rows:
        if any(k in kn for k in ("conv345", "conv_ksplit", "conv_kwalk", "conv12")):
            per.setdefault(kn, {})[cn] = avg
            print(kn.split("(")[0][-48:], cn, f"avg {avg:.4g} max {mx:.4g} n {n}")
    for kn, v in per.items():
        d = next((x for k, x in durs.items() if k.split("(")[0] == kn.split("(")[0]), None)
        if d and "GRBM_GUI_ACTIVE" in v:
            cyc = v["GRBM_GUI_ACTIVE"] / 8.0
            line = f"== {kn.split('(')[0][-48:]}: {d / 1e3:.1f} us in this pass, {cyc:.0f} cycles -> {cyc / d:.2f} GHz"
            if "SQ_VALU_MFMA_BUSY_CYCLES" in v:
                line += f", MFMA busy {v['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * cyc):.3f}"
            print(line)
PY
cat $OUT/summary.txt
rm -f $OUT/*/*.db $OUT/*/*/*.db
