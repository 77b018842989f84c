"""Per-call latency of checkMotion on ONE edge by edge length and verdict: one launch per call (check_motions_few_kernel)
against the resident pool (artp_set_persistent_latency).  Edges between random accepted states of the C2 map."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from art_planner_amd.context import Context
from synthetic import make_map
gm = make_map(400, 0.04, seed=1234)
ctx = Context(0, "yaml"); ctx.upload_map(gm)
se3 = ctx.sample_states(1, 0, 20000)
acc = se3[ctx.validate_states(se3) != 0]
rng = np.random.default_rng(3)
ia = rng.integers(0, len(acc), 4000); ib = rng.integers(0, len(acc), 4000)
a, b = np.ascontiguousarray(acc[ia]), np.ascontiguousarray(acc[ib])
d = np.hypot(a[:, 0] - b[:, 0], a[:, 1] - b[:, 1])
ok, t, _ = ctx.check_motions_last_valid(a, b)
_, ni = ctx.check_edges_interp(a, b)
# nd of checkMotion: count via last_t? use a proxy: distance
for mode in ("few", "pool"):
    ctx.set_persistent_latency(mode == "pool")
    rows = []
    for lo, hi in ((0, 0.25), (0.25, 0.5), (0.5, 1.0), (1.0, 2.0), (2.0, 4.0)):
        for want in (1, 0):
            idx = np.flatnonzero((d >= lo) & (d < hi) & (ok == want))[:40]
            if len(idx) < 5: continue
            for i in idx: ctx.check_motions(a[i:i+1], b[i:i+1])
            t0 = time.perf_counter()
            for _ in range(10):
                for i in idx: ctx.check_motions(a[i:i+1], b[i:i+1])
            us = (time.perf_counter() - t0) / (10 * len(idx)) * 1e6
            rows.append(f"{mode} d in [{lo},{hi}) valid={want} edges={len(idx)}: {us:.1f} us/call")
    print("\n".join(rows))
ctx.close()
