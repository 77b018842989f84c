#!/bin/bash
python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --skip-extras 2>&1 | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('sampler_ms', d['sampler_ms_per_batch'], 'value', d['value'])"
