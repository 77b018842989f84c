"""Reproducer for the stop-word race of the resident edge pool: every call is made 186 - 202 us after the previous one returned (around
the workgroups' idle limit, so that requests keep going out while part of the pool has left), modes with different task counts
alternate.  python scripts/pool_race_repro.py [calls]   (ARTP_LIB selects the library)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from art_planner_amd.context import Context  # noqa: E402
from synthetic import make_map  # noqa: E402

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
gm = make_map(400, 0.04, seed=1234)
ctx = Context(0, "yaml")
ctx.upload_map(gm)
rng = np.random.default_rng(5)
se3 = ctx.sample_states(5, 0, 20000)
lab = ctx.validate_states(se3)
acc = se3[lab != 0]
m = 2000
ia = rng.integers(0, len(acc), m)
a = acc[ia]
d = np.hypot(a[:, None, 0] - acc[None, :, 0], a[:, None, 1] - acc[None, :, 1])
d[np.arange(m), ia] = np.inf
b = acc[np.argsort(d, axis=1)[np.arange(m), rng.integers(0, 10, m)]]
ok, t, st = ctx.check_motions_last_valid(a, b)
oki, ni = ctx.check_edges_interp(a, b)
ctx.set_persistent_latency(True)
bad = 0
t0 = time.time()
for r in range(calls):
    i = int(rng.integers(0, m - 1))
    if r & 1:
        o3, n3 = ctx.check_edges_interp(a[i:i + 1], b[i:i + 1])
        bad += int(o3[0] != oki[i]) + int(n3[0] != ni[i])
    else:
        o2, t2, s2 = ctx.check_motions_last_valid(a[i:i + 1], b[i:i + 1])
        bad += int(o2[0] != ok[i]) + int(t2[0] != t[i]) + int(not ((s2[0] == st[i]) | (np.isnan(s2[0]) & np.isnan(st[i]))).all())
    t_end = time.perf_counter() + rng.uniform(186e-6, 202e-6)
    while time.perf_counter() < t_end:
        pass
s = ctx.persistent_latency_stats()
print(f"{calls} calls in {time.time() - t0:.0f} s: {bad} wrong answers; launches {s['launches']}, requests {s['requests']}")
ctx.close()
sys.exit(1 if bad else 0)
