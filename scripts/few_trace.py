"""Phase timestamps of one workgroup of check_motions_few_kernel (timing build: `make -C art_planner_amd/csrc timing`;
ARTP_LIB=art_planner_amd/csrc/libartp_timing.so python scripts/few_trace.py)."""
import ctypes as C
import os
import sys

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
names = ["edge loaded", "count+slerp", "state", "box", "barrier+verdict", "arrive atomic"]
for i in range(12):
    ok = ctx.check_motions(a[i:i + 1], b[i:i + 1])
    out = (C.c_ulonglong * 40)()
    L.artp_debug_few_trace(out)
    t = np.array(list(out), dtype=np.float64).reshape(5, 8) * 0.01   # 100 MHz -> us
    d = np.diff(t[:, :7], axis=1)
    print(f"edge {i} valid={int(ok[0])}")
    for k in range(5):
        print(f"   box {k}: " + ", ".join(f"{n} {x:5.2f}" for n, x in zip(names, d[k])) + f"   (wg total {t[k, 6] - t[k, 0]:.2f} us)")
ctx.close()
