"""Where the reference-order LazyPRM* solve spends its time ($ARTP_SOLVE_TIMING, variants build):
   ARTP_SOLVE_TIMING=1 ARTP_LIB=art_planner_amd/csrc/libartp_variants.so python scripts/lazy_timing.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from art_planner_amd.context import Context
from art_planner_amd.roadmap import Roadmap
from synthetic import map_from_device, raw_map
ctx = Context(0, "yaml"); gm = map_from_device(ctx, raw_map(400, 0.04, seed=1234))
probe = ctx.sample_states(42, 9_000_000, 1 << 15); okp = probe[ctx.validate_states(probe) != 0]
s = okp[np.argmin(np.hypot(okp[:, 0] - (gm.pos_x - 0.4 * gm.len_x), okp[:, 1] - (gm.pos_y - 0.4 * gm.len_y)))]
g = okp[np.argmin(np.hypot(okp[:, 0] - (gm.pos_x + 0.4 * gm.len_x), okp[:, 1] - (gm.pos_y + 0.4 * gm.len_y)))]
for n in (10000, 20000):
    for rep in range(2):
        rm = Roadmap(ctx, s, g, seed=42, n_milestones=n, construction=2)
        t0 = time.perf_counter(); p, c, r = rm.solve(); t1 = time.perf_counter()
        print(f"n {n} rep {rep}: solve {1e3 * (t1 - t0):.2f} ms, removals {r}, cost {c:.4f}", flush=True)
        rm.close()
ctx.close()
