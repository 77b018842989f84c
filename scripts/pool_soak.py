"""Soak of the resident latency paths (artp_set_persistent_latency): N calls of one / two edges through all three edge entry
points and of one / two states, with random gaps around the 200 us idle limit, batch launches and map writes in between; every
answer compared with the batch pipeline's for the same edge / state.  python scripts/pool_soak.py [calls]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from art_planner_amd.context import Context  # noqa: E402
from synthetic import make_map  # noqa: E402

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
gm = make_map(400, 0.04, seed=1234)
ctx = Context(0, "yaml")
ctx.upload_map(gm)
rng = np.random.default_rng(99)
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
ctx.set_persistent_latency(True)
gaps = np.array([0, 0, 0, 0, 0, 0, 100e-6, 180e-6, 195e-6, 200e-6, 205e-6, 220e-6, 400e-6, 2e-3])
bad = 0
t_start = time.time()
elev = gm["elevation"]
for r in range(calls):
    i = int(rng.integers(0, m - 1))
    k = 1 + (r & 1)
    w = r % 5
    if w == 0:
        got0 = ctx.check_motions(a[i:i + k], b[i:i + k])
        bad += int((got0 != ok[i:i + k]).sum())
    elif w == 1:
        o2, t2, s2 = ctx.check_motions_last_valid(a[i:i + k], b[i:i + k])
        bad += int((o2 != ok[i:i + k]).sum()) + int((t2 != t[i:i + k]).sum())
        bad += int((~((s2 == st[i:i + k]) | (np.isnan(s2) & np.isnan(st[i:i + k])))).any(axis=1).sum())
    elif w == 2:
        o3, n3 = ctx.check_edges_interp(a[i:i + k], b[i:i + k])
        bad += int((o3 != oki[i:i + k]).sum()) + int((n3 != ni[i:i + k]).sum())
    else:
        j = int(rng.integers(0, len(se3) - 2))
        bad += int((ctx.validate_states(se3[j:j + k]) != lab[j:j + k]).sum())
    if r % 5003 == 0:
        bad += int((ctx.validate_states(se3[:4096]) != lab[:4096]).sum())       # a batch launch next to the resident kernels
    if r % 20011 == 0:                                                           # a map write (same samples): both restart
        ctx.upload_layer(0, elev, gm.len_x, gm.len_y, gm.pos_x, gm.pos_y)
    if w < 3 and r % 997 == 0:
        req_before = ctx.persistent_latency_stats()["requests"]
    g = gaps[int(rng.integers(0, len(gaps)))]
    if g:
        t_end = time.perf_counter() + g
        while time.perf_counter() < t_end:
            pass
    if bad:
        print("MISMATCH at call", r, "kind", w, "edge/state index", i if w < 3 else j, "k", k, flush=True)
        if w < 3:
            np.set_printoptions(precision=17, linewidth=200)
            print("  expected ok", ok[i:i + k], "t", t[i:i + k], "interp ok", oki[i:i + k], "n_interp", ni[i:i + k])
            if w == 0:
                print("  got", got0)
            elif w == 1:
                print("  got ok", o2, "t", t2, "state equal", (s2 == st[i:i + k]).all(axis=1))
            else:
                print("  got interp ok", o3, "n", n3)
            for rep in range(3):
                s0 = ctx.persistent_latency_stats()
                o2, t2, s2 = ctx.check_motions_last_valid(a[i:i + k], b[i:i + k])
                o3, n3 = ctx.check_edges_interp(a[i:i + k], b[i:i + k])
                s1 = ctx.persistent_latency_stats()
                print("  again:", o2, t2, o3, n3, "requests +", s1["requests"] - s0["requests"], "launches +", s1["launches"] - s0["launches"])
        break
stats = ctx.persistent_latency_stats()
line = (f"pool / service soak: {r + 1} calls in {time.time() - t_start:.0f} s, {bad} mismatches against the batch pipeline; "
        f"launches of the resident kernels {stats['launches']}, requests answered {stats['requests']}")
print(line)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "pool_soak.txt"), "w").write(line + "\n")
ctx.set_persistent_latency(False)
ctx.close()
sys.exit(1 if bad else 0)
