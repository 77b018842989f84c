"""Phase cycles of conv345_kernel (timing build: make -C art_planner_amd/csrc timing).
usage: ARTP_LIB=art_planner_amd/csrc/libartp_timing.so python scripts/cnn_timing.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import numpy as np, torch
import convert_weights
from art_planner_amd import _capi
from art_planner_amd.context import Context
from synthetic import raw_map
L = _capi.load()
ctx = Context(0, "yaml")
ctx.cost_load_weights(convert_weights.to_blob(convert_weights.random_params(0)))
names = ["patch load+barrier", "conv3 mfma", "conv3 store", "barrier", "conv4 mfma", "conv4 store", "barrier", "pool+barrier",
         "conv5 mfma", "conv5 store+barrier", "tile store", "(fused) window -> LDS + barrier", "(fused) conv1 o conv2 -> patch"]
for n, seed in ((400, 1234), (800, 77)):
    g = raw_map(n, 0.04, seed=seed)
    elv = np.ascontiguousarray(g["elevation"][::-1, ::-1]).astype(np.float32)
    ctx.cost_update_map(elv, g.res, g.len_x, g.len_y)
    out = (C.c_ulonglong * 20)()
    L.artp_debug_stage_cycles(out, 4)
    ctx.cost_update_map(elv, g.res, g.len_x, g.len_y)
    L.artp_debug_stage_cycles(out, 4)
    a = np.array(list(out)[:16], dtype=np.float64)
    wg = a[15]
    print(f"map {n}: {int(wg)} workgroups, cycles per workgroup (wavefront 0): total {a[:13].sum() / wg:.0f}")
    for k, nm in enumerate(names):
        print(f"   {nm:24s} {a[k] / wg:9.0f}")
    out8 = (C.c_ulonglong * 48)()
    L.artp_debug_stage_cycles(out8, 5)
    ctx.cost_update_map(elv, g.res, g.len_x, g.len_y)
    L.artp_debug_stage_cycles(out8, 5)
    k = np.array(list(out8), dtype=np.float64).reshape(12, 4)
    nwg = 242 if n == 400 else 256
    print(f"   the 15 x 15 kernel (conv_ksplit_kernel: patch / main loop / reduction / tile store; conv_kwalk_kernel: wave = nt + 3 kq,"
          f" sums over its tiles): cycles per workgroup and wavefront, {nwg} workgroup slots")
    print("      wave   patch loads   main loop   [2]               [3]            total")
    for w in range(12):
        r = k[w] / nwg
        print(f"      {w:4d} {r[0]:12.0f} {r[1]:11.0f} {r[2]:17.0f} {r[3]:11.0f} {r.sum():10.0f}")
ctx.close()
