"""Per-call latency of the host-buffer validity API for small batches (the per-state isValid() of the host mirror)."""
import time, numpy as np, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from art_planner_amd.context import Context
from synthetic import make_map
gm = make_map(400, 0.04, seed=1234)
ctx = Context(0, "yaml"); ctx.upload_map(gm)
se3 = ctx.sample_states(1, 0, 4096)
for n in (1, 64, 128, 129, 1024):
    ctx.validate_states(se3[:n])
    t0 = time.perf_counter()
    for _ in range(200): ctx.validate_states(se3[:n])
    print(n, "states:", (time.perf_counter() - t0) / 200 * 1e6, "us per call")
ctx.set_persistent_latency(True)
for n in (1, 4, 16):
    ctx.validate_states(se3[:n])
    t0 = time.perf_counter()
    for r in range(2000): ctx.validate_states(se3[r:r + n])
    print(n, "states through the persistent service:", (time.perf_counter() - t0) / 2000 * 1e6, "us per call", ctx.persistent_latency_stats())
ctx.set_persistent_latency(False)
