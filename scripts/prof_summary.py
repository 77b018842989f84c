#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd (sqlite) outputs: per-kernel time stats and PMC sums per dispatch.
Usage: python scripts/prof_summary.py gpurun_out/prof_<tag> > profiles/<name>.txt"""
import glob
import os
import sqlite3
import sys


def q(db, sql):
    return db.execute(sql).fetchall()


def main(d):
    for path in sorted(glob.glob(os.path.join(d, "*", "*_results.db"))):
        db = sqlite3.connect(path)
        print(f"== {os.path.relpath(path, d)}")
        cols = [r[1] for r in q(db, "pragma table_info(kernels)")]
        rows = q(db, "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                     "from kernels group by name order by sum(end-start) desc")
        tot = sum(r[2] for r in rows) or 1
        print(f"{'kernel':70s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'%':>6s}")
        for name, n, s, a, mn, mx in rows:
            print(f"{name[:70]:70s} {n:6d} {s/1e6:10.3f} {a/1e3:10.1f} {mn/1e3:10.1f} {mx/1e3:10.1f} {100*s/tot:6.1f}")
        try:
            ccols = [r[1] for r in q(db, "pragma table_info(counters_collection)")]
            pm = q(db, "select kernel_name, counter_name, count(*), sum(value), avg(value) from counters_collection "
                       "group by kernel_name, counter_name order by kernel_name, counter_name")
            if pm:
                print(f"  {'kernel':50s} {'counter':26s} {'dispatches':>10s} {'avg/dispatch':>18s}")
                for kn, cn, n, s, a in pm:
                    print(f"  {kn[:50]:50s} {cn:26s} {n:10d} {a:18.1f}")
        except sqlite3.OperationalError as e:
            print("  (no counters:", e, ")")
        # resources of our kernels
        try:
            kc = [r[1] for r in q(db, "pragma table_info(kernels)")]
            want = [c for c in ("name", "vgpr_count", "accum_vgpr_count", "sgpr_count", "lds_size", "scratch_size",
                                "workgroup_size", "grid_size") if c in kc]
            for r in q(db, f"select distinct {','.join(want)} from kernels where name like '%artp%'"):
                print("  res:", dict(zip(want, r)))
        except sqlite3.OperationalError:
            pass


if __name__ == "__main__":
    main(sys.argv[1])
