#!/bin/bash
# tuning: memory-pipeline counters of the pipeline kernels (PMC child workload), several small passes
cd /tmp && export TMPDIR=/tmp
run() {
  rm -rf /tmp/pm; mkdir -p /tmp/pm
  timeout 150 rocprofv3 --kernel-trace --pmc "$@" -d /tmp/pm -o t -- python $GRAFT_REPO_ROOT/bench.py --pmc-child > /tmp/pm/log 2>&1 || tail -3 /tmp/pm/log
  python - <<PY
import glob, sqlite3
dbs = glob.glob("/tmp/pm/**/*_results.db", recursive=True)
if dbs:
    db = sqlite3.connect(dbs[0])
    for kn, cn, mx in db.execute("select kernel_name, counter_name, max(value) from counters_collection group by kernel_name, counter_name order by kernel_name, counter_name"):
        if any(x in kn for x in ("classify", "feet_stream", "resolve_boxes_kernel<2, 64, 0>", "sample_states")):
            print("  %-42s %-40s %16.0f" % (kn.split("(")[0][-42:], cn, mx))
PY
}
# each pass under its own timeout (a TA / TD pass once hung the box until gpurun's limit); PASSES="1 3" selects
PASSES=${PASSES:-1 2 3}
for p in $PASSES; do
  case $p in
    1) run TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum ;;
    2) run FETCH_SIZE ;;
    3) run SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU ;;
    4) run SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE ;;
  esac
done
