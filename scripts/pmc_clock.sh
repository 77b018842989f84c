#!/bin/bash
# tuning: calibrate the "VALU busy" normalisation -- GRBM_GUI_ACTIVE (cycles per XCD) against kernel duration
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pc; mkdir -p /tmp/pc
timeout 150 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d /tmp/pc -o t -- python $GRAFT_REPO_ROOT/bench.py --pmc-child > /tmp/pc/log 2>&1 || tail -3 /tmp/pc/log
python - <<'PY'
import glob, sqlite3
db = sqlite3.connect(glob.glob("/tmp/pc/**/*_results.db", recursive=True)[0])
dur = {n: d for n, d in db.execute("select name, max(end-start) from kernels group by name")}
rows = db.execute("select kernel_name, counter_name, max(value) from counters_collection group by kernel_name, counter_name").fetchall()
by = {}
for kn, cn, v in rows:
    by.setdefault(kn, {})[cn] = v
for kn, c in by.items():
    if not any(x in kn for x in ("classify", "feet_stream", "resolve_boxes_kernel<2, 64, 0>", "sample_states")):
        continue
    d_us = dur.get(kn, 0) / 1e3
    g = c.get("GRBM_GUI_ACTIVE", 0) / 8.0
    print("%-40s dur %.1f us  GRBM/8 %.0f cycles -> %.0f MHz  ACTIVE_INST_VALU %.3g  INSTS_VALU %.3g  ratio %.3f  busy(GRBM) %.3f  busy(2400 MHz) %.3f  SQ_BUSY_CYCLES %.3g WAVE_CYCLES %.3g" % (
        kn.split("(")[0][-40:], d_us, g, g / max(d_us, 1e-9), c.get("SQ_ACTIVE_INST_VALU", 0), c.get("SQ_INSTS_VALU", 0),
        c.get("SQ_ACTIVE_INST_VALU", 0) / max(c.get("SQ_INSTS_VALU", 1), 1),
        c.get("SQ_ACTIVE_INST_VALU", 0) * 4 / (1024 * max(g, 1)), c.get("SQ_ACTIVE_INST_VALU", 0) * 4 / (1024 * d_us * 2400),
        c.get("SQ_BUSY_CYCLES", 0), c.get("SQ_WAVE_CYCLES", 0)))
PY
