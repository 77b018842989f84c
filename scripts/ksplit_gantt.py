"""The 15 x 15 kernel's launch as a Gantt chart per CU (timing build).  One record per workgroup: where it ran (XCC, SE, CU), when
it started, when its first MFMA step could issue, when its main loop ended, when it ended (s_memrealtime, 100 MHz).
usage: ARTP_LIB=art_planner_amd/csrc/libartp_timing.so python scripts/ksplit_gantt.py [400|800]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import numpy as np
import convert_weights
from art_planner_amd import _capi
from art_planner_amd.context import Context
from synthetic import raw_map
L = _capi.load()
ctx = Context(0, "yaml")
ctx.cost_load_weights(convert_weights.to_blob(convert_weights.random_params(0)))
for n in [int(a) for a in sys.argv[1:]] or [400, 800]:
    g = raw_map(n, 0.04, seed=77)
    elv = np.ascontiguousarray(g["elevation"][::-1, ::-1]).astype(np.float32)
    for _ in range(3):
        ctx.cost_update_map(elv, g.res, g.len_x, g.len_y)
    out = (C.c_ulonglong * 6144)()
    assert L.artp_debug_stage_cycles(out, 6) == 0
    r = np.array(list(out), dtype=np.uint64).reshape(1024, 6)
    fh, fw = ctx.cost_features().shape[:2]
    bidx = np.flatnonzero(r[:, 2] > 0)
    r = r[r[:, 2] > 0]
    hw, xcc = r[:, 0].astype(np.int64), r[:, 1].astype(np.int64) & 0xf
    cu, sh, se = (hw >> 8) & 0xf, (hw >> 12) & 1, (hw >> 13) & 0x7
    key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    t = r[:, 2:].astype(np.float64) * 0.01  # us
    t0 = t[:, 0].min()
    t -= t0
    print(f"map {n}: feature map {fh} x {fw}, {len(r)} workgroup records, {len(np.unique(key))} distinct (xcc, se, sh, cu); launch spans {t[:, 3].max():.1f} us")
    print(f"   start: first {t[:, 0].min():.2f} last {t[:, 0].max():.2f};  patch phase mean {np.mean(t[:, 1] - t[:, 0]):.2f} us;  main loop mean "
          f"{np.mean(t[:, 2] - t[:, 1]):.2f} (min {np.min(t[:, 2] - t[:, 1]):.2f} max {np.max(t[:, 2] - t[:, 1]):.2f});  reduction + store mean {np.mean(t[:, 3] - t[:, 2]):.2f}")
    per = {}
    for i, k in enumerate(key):
        per.setdefault(int(k), []).append(np.append(t[i], bidx[i]))
    cnt = np.array([len(v) for v in per.values()])
    print(f"   workgroups per CU: min {cnt.min()} max {cnt.max()} histogram {np.bincount(cnt).tolist()}")
    ends = np.array([max(x[3] for x in v) for v in per.values()])
    print(f"   a CU's last workgroup ends at: min {ends.min():.1f} median {np.median(ends):.1f} max {ends.max():.1f} us")
    # MFMA-phase coverage per CU: union of [main0, main1] intervals
    cov = []
    for v in per.values():
        iv = sorted((x[1], x[2]) for x in v)
        tot, cur_s, cur_e = 0.0, iv[0][0], iv[0][1]
        for s_, e_ in iv[1:]:
            if s_ > cur_e:
                tot += cur_e - cur_s
                cur_s, cur_e = s_, e_
            else:
                cur_e = max(cur_e, e_)
        tot += cur_e - cur_s
        cov.append(tot)
    print(f"   time a CU has at least one workgroup in its main loop: mean {np.mean(cov):.1f} us of {t[:, 3].max():.1f}")
    for k in list(per)[:6]:
        print(f"   CU key {k}: " + "  ".join(f"#{int(x[4])} [{x[0]:.1f} | {x[1]:.1f} .. {x[2]:.1f} | {x[3]:.1f}]" for x in sorted(per[k], key=lambda x: x[0])))
ctx.close()
