"""Per-call latency of the host-buffer edge API for small batches (ob::MotionValidator::checkMotion of the host mirror:
one edge per call; a solution path of a few dozen edges per call) -- latency kernel vs the batch pipeline."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from art_planner_amd.context import Context  # noqa: E402
from synthetic import make_map  # noqa: E402

gm = make_map(400, 0.04, seed=1234)
ctx = Context(0, "yaml")
ctx.upload_map(gm)
se3 = ctx.sample_states(1, 0, 20000)
acc = se3[ctx.validate_states(se3) != 0]
order = np.argsort(acc[:, 0])
acc = acc[order]
a, b = acc[:-1], acc[1:]
keep = np.hypot(a[:, 0] - b[:, 0], a[:, 1] - b[:, 1]) < 2.0
a, b = np.ascontiguousarray(a[keep][:512]), np.ascontiguousarray(b[keep][:512])
modes = (("few", True), ("batch", False)) if "--few-only" not in sys.argv else (("few", True),)
for name, few in modes:
    ctx.set_few_edges(few)
    for n in (1, 8, 32, 64):
        reps = 300 if few else 60
        for _ in range(5):
            ctx.check_motions(a[:n], b[:n])
        t0 = time.perf_counter()
        for r in range(reps):
            i = (r * n) % (len(a) - n)
            ctx.check_motions(a[i:i + n], b[i:i + n])
        t1 = time.perf_counter()
        for r in range(reps):
            i = (r * n) % (len(a) - n)
            ctx.check_motions_last_valid(a[i:i + n], b[i:i + n])
        t2 = time.perf_counter()
        print(f"{name:5s} n={n:3d}: checkMotion {(t1 - t0) / reps * 1e6:8.1f} us/call, lastValid overload {(t2 - t1) / reps * 1e6:8.1f} us/call")
# the resident pool (artp_set_persistent_latency): calls of one and two edges without a launch
ctx.set_few_edges(True)
ctx.set_persistent_latency(True)
for n in (1, 2):
    reps = 2000
    for _ in range(20):
        ctx.check_motions(a[:n], b[:n])
    s0 = ctx.persistent_latency_stats()
    t0 = time.perf_counter()
    for r in range(reps):
        i = (r * n) % (len(a) - n)
        ctx.check_motions(a[i:i + n], b[i:i + n])
    t1 = time.perf_counter()
    for r in range(reps):
        i = (r * n) % (len(a) - n)
        ctx.check_motions_last_valid(a[i:i + n], b[i:i + n])
    t2 = time.perf_counter()
    s1 = ctx.persistent_latency_stats()
    print(f"pool  n={n:3d}: checkMotion {(t1 - t0) / reps * 1e6:8.1f} us/call, lastValid overload {(t2 - t1) / reps * 1e6:8.1f} us/call "
          f"({s1['requests'] - s0['requests']} requests, {s1['launches'] - s0['launches']} launches)")
ctx.set_persistent_latency(False)
ctx.close()
