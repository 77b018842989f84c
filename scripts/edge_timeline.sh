#!/bin/bash
# Timeline of one artp_check_motions_dev batch (kernel start / duration / gap to the previous kernel's end)
OUT=/tmp/edge_tl
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace -d $OUT/t -o t -- python $GRAFT_REPO_ROOT/bench.py --pmc-child check_motion > $OUT/log 2>&1
python - <<PY
import glob, sqlite3
db = sqlite3.connect(glob.glob("$OUT/t/*_results.db")[0])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
try:
    mc = db.execute("select name, start, end from memory_copies order by start").fetchall()
except Exception as e:
    mc = []
ev = sorted([(s, e, n[:60]) for n, s, e in rows] + [(s, e, "COPY " + str(n)[:40]) for n, s, e in mc])
# the last motion_plan_kernel starts the last batch
idx = max(i for i, x in enumerate(ev) if "motion_plan" in x[2])
t0 = ev[idx][0]
prev_end = t0
for s, e, n in ev[idx:]:
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {(s - prev_end) / 1e3:8.1f}  {n}")
    prev_end = max(prev_end, e)
PY
