#!/bin/bash
# tuning: memory-pipeline counters of the pipeline kernels (PMC child workload), several small passes
cd /tmp && export TMPDIR=/tmp
run() {
  rm -rf /tmp/pm; mkdir -p /tmp/pm
  rocprofv3 --kernel-trace --pmc "$@" -d /tmp/pm -o t -- python $GRAFT_REPO_ROOT/bench.py --pmc-child > /tmp/pm/log 2>&1 || tail -3 /tmp/pm/log
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
run TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum
run TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
run TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum
run TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_TCC_WRITE_REQ_sum
run TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum GRBM_GUI_ACTIVE
