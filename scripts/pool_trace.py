"""Phase timestamps of one request through the resident edge pool (timing build: `make -C art_planner_amd/csrc timing`;
ARTP_LIB=art_planner_amd/csrc/libartp_timing.so python scripts/pool_trace.py)."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from art_planner_amd import _capi  # noqa: E402
from art_planner_amd.context import Context  # noqa: E402
from synthetic import make_map  # noqa: E402

gm = make_map(400, 0.04, seed=1234)
ctx = Context(0, "yaml")
ctx.upload_map(gm)
L = _capi.load()
se3 = ctx.sample_states(1, 0, 20000)
acc = se3[ctx.validate_states(se3) != 0]
acc = acc[np.argsort(acc[:, 0])]
a, b = acc[:-1], acc[1:]
keep = np.hypot(a[:, 0] - b[:, 0], a[:, 1] - b[:, 1]) < 2.0
a, b = np.ascontiguousarray(a[keep][:64]), np.ascontiguousarray(b[keep][:64])
ctx.set_persistent_latency(True)
names = ["count+slerp", "state", "box", "barrier, verdict, slot written"]
for i in range(12):
    for _ in range(3):
        t0 = time.perf_counter()
        ok = ctx.check_motions(a[i:i + 1], b[i:i + 1])
        t1 = time.perf_counter()
    time.sleep(0.01)   # the pool has left: its marks are in memory
    out = (C.c_ulonglong * 16)()
    L.artp_debug_pool_trace(out)
    ph = np.array(list(out), dtype=np.float64)[:5] * 0.01
    d = np.diff(ph)
    print(f"edge {i} valid={int(ok[0])} host call {1e6 * (t1 - t0):.1f} us; workgroup 1 from holding the request to its slot {ph[4] - ph[0]:.2f} us: "
          + ", ".join(f"{n} {x:5.2f}" for n, x in zip(names, d)))
ctx.close()
