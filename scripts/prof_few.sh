#!/bin/bash
# kernel trace of the small-batch edge latency path (scripts/edge_latency.py --few-only)
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_few
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $GRAFT_REPO_ROOT/scripts/edge_latency.py --few-only > $OUT/trace.log 2>&1
tail -5 $OUT/trace.log
python - <<PY
import glob, sqlite3
for p in glob.glob("$OUT/trace/*_results.db") + glob.glob("$OUT/trace/*/*_results.db"):
    db = sqlite3.connect(p)
    print([r[1] for r in db.execute("pragma table_info(kernels)")])
    rows = db.execute("select name, end-start from kernels where name like '%check_motions_few%' order by start").fetchall()
    import numpy as np
    d = np.array([r[1] for r in rows]) / 1e3
    # edge_latency.py --few-only: per n in (1, 8, 32, 64): 5 + 300 checkMotion calls, then 300 lastValid calls
    at = 0
    for n in (1, 8, 32, 64):
        cm, lv = d[at + 5:at + 305], d[at + 305:at + 605]
        at += 605
        print(f"n={n:3d}: check_motions_few_kernel median {np.median(cm):.1f} us (min {cm.min():.1f}, max {cm.max():.1f}); lastValid form median {np.median(lv):.1f} us")
PY
rm -f $OUT/trace/*.db $OUT/trace/*/*.db
