"""Tuning: search time of large roadmaps (device label-correcting search)."""
import sys, time, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from art_planner_amd.context import Context
from art_planner_amd.roadmap import Roadmap
from synthetic import map_from_device, raw_map
ctx = Context(0, "yaml"); gm = map_from_device(ctx, raw_map(400, 0.04, seed=1234))
probe = ctx.sample_states(42, 9_000_000, 1 << 15); okp = probe[ctx.validate_states(probe) != 0]
s = okp[np.argmin(np.hypot(okp[:, 0] + 6.4, okp[:, 1] + 6.4))]; g = okp[np.argmin(np.hypot(okp[:, 0] - 6.4, okp[:, 1] - 6.4))]
for n in (30000, 100000):
    rm = Roadmap(ctx, s, g, n_milestones=n, seed=42)
    t0 = time.perf_counter(); p, c, r = rm.solve(); t1 = time.perf_counter()
    p2, c2, r2 = rm.solve(); t2 = time.perf_counter()
    print(n, "first solve ms", (t1 - t0) * 1e3, "second", (t2 - t1) * 1e3, "cost", c, c2, "states", len(p), "removals", r, r2)
    rm.close()
