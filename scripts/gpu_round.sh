#!/bin/bash
# One GPU-box visit: the -m gpu suite, the default bench (incl. its live PMC passes), a kernel-trace profile.
#   gpurun --timeout 1500 -- 'bash scripts/gpu_round.sh <tag>'
TAG=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu_$TAG.log 2>&1
echo "pytest rc $?" >> $OUT/pytest_gpu_$TAG.log
tail -15 $OUT/pytest_gpu_$TAG.log
timeout 600 python bench.py > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err
echo "bench rc $?"
python - <<PY
import json
try:
    lines = open("$OUT/bench_$TAG.json").read().strip().splitlines()
    short = lines[-1]
    print("contract line bytes", len(short), "parses", bool(json.loads(short)))
    d = json.loads([l for l in lines if l.startswith("BENCH_DETAIL ")][-1][len("BENCH_DETAIL "):])
    r = d["roofline"]
    print("value", d["value"], "ms/step", d["ms_per_step"], "kernel_ms", r["kernel_ms"], "fused_ms", r.get("fused_sample_validate_ms"))
    print("pmc", r["pmc_source"])
    b = r.get("binding")
    if b:
        print("bound", r["bound"], "frac", r["frac"], "fractions", r["occupancy_fractions"], "traffic", r["traffic"], "alg", r["algorithmic_hbm"]["ratio_to_peak"])
        for k, v in b["per_kernel"].items():
            print("  ", k, {x: (round(y, 3) if isinstance(y, float) else y) for x, y in v.items()})
    for k in ("edges", "motion_cost_c3", "replan_cycle_c5", "c4_800_defaults", "preprocess_n2"):
        print(k, json.dumps(d.get(k))[:900])
    print("cpu", json.dumps(d.get("cpu_baseline"))[:1500])
    print("roadmap", json.dumps(d.get("roadmap_n1"))[:800])
except Exception as e:
    print("bench parse failed", e)
    print(open("$OUT/bench_$TAG.err").read()[-2000:])
PY
# the multi-GPU code path with ONE rank (torch.distributed / RCCL world size 1: communicator, bitmap all-gather, re-materialisation and
# the edge exchange are real, the scaling is not): its contract line goes to profiles/ next to the default run's
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --gpus 1 --force-dist --skip-extras --no-pmc --no-cpu-baseline > $OUT/bench_force_dist_$TAG.json 2> $OUT/bench_force_dist_$TAG.err
echo "bench --force-dist rc $?"
tail -1 $OUT/bench_force_dist_$TAG.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('force-dist value', d['value'], 'ms/step', d['ms_per_step'], 'distributed', json.dumps(d.get('distributed'))[:600], 'gather_error', d.get('gather_error'))" || tail -5 $OUT/bench_force_dist_$TAG.err
mkdir -p $OUT/prof_$TAG
cd /tmp && export TMPDIR=/tmp
# --lanes 1: one batch at a time on one stream, the configuration of roofline.kernel_ms (rocprofv3 serialises the
# kernels of concurrent streams anyway)
rocprofv3 --kernel-trace --stats -d $OUT/prof_$TAG/trace -o trace -- python $GRAFT_REPO_ROOT/bench.py --lanes 1 --steps 3 --warmup 1 --no-cpu-baseline --no-pmc > $OUT/prof_$TAG/trace.log 2>&1
python $GRAFT_REPO_ROOT/scripts/prof_summary.py $OUT/prof_$TAG > $OUT/prof_$TAG/summary.txt 2>&1
# the headline workloads ALONE (bench.py --pmc-child <mode>: 4 batches and nothing else), so that a kernel's avg column
# IS its per-launch time: states = fused sample + validate of 2^22 states; check_motion = 2^18 edges; sampler alone
for MODE in states check_motion sampler; do
  rocprofv3 --kernel-trace --stats -d $OUT/prof_$TAG/child_$MODE -o trace -- python $GRAFT_REPO_ROOT/bench.py --pmc-child $MODE > $OUT/prof_$TAG/child_$MODE.log 2>&1
done
python $GRAFT_REPO_ROOT/scripts/prof_summary.py $OUT/prof_$TAG > $OUT/prof_$TAG/summary_all.txt 2>&1
# the same child WITHOUT the profiler, and what it measured under it, cold (batches 2-4 of the process) and warm (41-48):
# the warm_us column of child_states adds up to the warm figure = ms_per_step of the bench line
echo "== step time of the states child by HIP events (same process as the trace / without the profiler)" >> $OUT/prof_$TAG/summary_all.txt
grep PMC_CHILD_STEP_MS $OUT/prof_$TAG/child_states.log | sed 's/^/under rocprofv3 --kernel-trace: /' >> $OUT/prof_$TAG/summary_all.txt
python $GRAFT_REPO_ROOT/bench.py --pmc-child states 2>/dev/null | grep PMC_CHILD_STEP_MS | sed 's/^/unprofiled:                     /' >> $OUT/prof_$TAG/summary_all.txt
rm -f $OUT/prof_$TAG/*/*.db $OUT/prof_$TAG/*/*/*.db
head -30 $OUT/prof_$TAG/summary.txt
