#!/bin/bash
# Tuning loop on the GPU box: headline step time + label hash, then per-kernel times of the pipeline launches
# (kernel trace of the PMC child workload: only 2^22-state batches).  Optional: ARTP_LIB=<variant .so>
cd $GRAFT_REPO_ROOT
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --skip-extras --no-pmc 2>/dev/null | tail -1 > /tmp/qb.json
python - <<'PY'
import json
d = json.loads(open("/tmp/qb.json").read())
print("value %.4g states/s  ms/step %.3f  validate_ms %.3f  fused_ms %.3f  sampler_ms %.3f  hash %s (r1: 15be340c7659f510)" % (
    d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["fused_sample_validate_ms"], d["sampler_ms_per_batch"], d["label_hash_batch0"]))
print(d["pipeline_counts_batch0"])
PY
mkdir -p /tmp/qp && cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/qp/t -o t -- python $GRAFT_REPO_ROOT/bench.py --pmc-child > /tmp/qp/log 2>&1
python - <<'PY'
import glob, sqlite3
db = sqlite3.connect(glob.glob("/tmp/qp/t/**/*_results.db", recursive=True)[0])
rows = db.execute("select name, count(*), max(end-start) from kernels group by name order by max(end-start) desc").fetchall()
tot = 0
for n, c, mx in rows[:9]:
    print("  %-58s calls %3d  max %8.1f us" % (n[:58], c, mx / 1e3))
PY
rm -rf /tmp/qp
