#!/bin/bash
# bench.py without the extras / CPU baseline: headline value, per-step time and queue counts
python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --skip-extras 2>&1 | tail -1 > /tmp/qb.json
python - <<'PY'
import json
d = json.loads(open("/tmp/qb.json").read())
print("value %.4g states/s  ms/step %.3f  kernel_ms %.3f  hash %s" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["label_hash_batch0"]))
print(d["pipeline_counts_batch0"])
PY
