"""Soak of the one-launch latency paths (no resident kernels): calls of 1..64 edges through the three edge entry points and of
1..16 states, back to back and with random gaps, batch launches and map writes in between; every answer compared with the
batch pipeline's.  python scripts/few_soak.py [calls]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from art_planner_amd.context import Context  # noqa: E402
from synthetic import make_map  # noqa: E402

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
gm = make_map(400, 0.04, seed=1234)
ctx = Context(0, "yaml")
ctx.upload_map(gm)
rng = np.random.default_rng(123)
se3 = ctx.sample_states(5, 0, 20000)
lab = ctx.validate_states(se3)
acc = se3[lab != 0]
m = 4000
ia = rng.integers(0, len(acc), m)
a = acc[ia]
d = np.hypot(a[:, None, 0] - acc[None, :, 0], a[:, None, 1] - acc[None, :, 1])
d[np.arange(m), ia] = np.inf
b = acc[np.argsort(d, axis=1)[np.arange(m), rng.integers(0, 10, m)]]
b[::11] = acc[rng.integers(0, len(acc), len(b[::11]))]
b[5::13] = se3[lab == 0][:len(b[5::13])]
ok, t, st = ctx.check_motions_last_valid(a, b)
oki, ni = ctx.check_edges_interp(a, b)
sizes = np.array([1, 1, 1, 2, 2, 3, 5, 8, 13, 21, 34, 55, 64])
bad = 0
t_start = time.time()
for r in range(calls):
    w = r % 4
    if w < 3:
        k = int(sizes[int(rng.integers(0, len(sizes)))])
        i = int(rng.integers(0, m - k))
        if w == 0:
            bad += int((ctx.check_motions(a[i:i + k], b[i:i + k]) != ok[i:i + k]).sum())
        elif w == 1:
            o2, t2, s2 = ctx.check_motions_last_valid(a[i:i + k], b[i:i + k])
            bad += int((o2 != ok[i:i + k]).sum()) + int((t2 != t[i:i + k]).sum())
            bad += int((~((s2 == st[i:i + k]) | (np.isnan(s2) & np.isnan(st[i:i + k])))).any(axis=1).sum())
        else:
            o3, n3 = ctx.check_edges_interp(a[i:i + k], b[i:i + k])
            bad += int((o3 != oki[i:i + k]).sum()) + int((n3 != ni[i:i + k]).sum())
    else:
        k = int(rng.integers(1, 17))
        j = int(rng.integers(0, len(se3) - k))
        bad += int((ctx.validate_states(se3[j:j + k]) != lab[j:j + k]).sum())
    if r % 5003 == 0:
        bad += int((ctx.validate_states(se3[:4096]) != lab[:4096]).sum())
    if r % 20011 == 0:
        ctx.upload_layer(0, gm["elevation"], gm.len_x, gm.len_y, gm.pos_x, gm.pos_y)
    if r % 7 == 0:
        t_end = time.perf_counter() + float(rng.uniform(0, 300e-6))
        while time.perf_counter() < t_end:
            pass
    if bad:
        print("MISMATCH at call", r, "kind", w, "size", k)
        break
line = f"one-launch latency paths soak: {r + 1} calls in {time.time() - t_start:.0f} s, {bad} mismatches against the batch pipeline"
print(line)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "few_soak.txt"), "w").write(line + "\n")
ctx.close()
sys.exit(1 if bad else 0)
