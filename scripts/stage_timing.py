"""Per-stage cycle totals of resolve_boxes_kernel (build with `make -C art_planner_amd/csrc timing`,
which writes libartp_timing.so; the shipped libartp.so carries no instrumentation).
usage: ARTP_LIB=art_planner_amd/csrc/libartp_timing.so python scripts/stage_timing.py"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from art_planner_amd import _capi
from art_planner_amd.context import Context
from synthetic import make_map

gm = make_map(400, 0.04, seed=1234)
ctx = Context(0, "yaml")
ctx.upload_map(gm)
L = _capi.load()
n = 1 << 22
se3 = torch.empty((n, 7), dtype=torch.float64, device="cuda:0")
va = torch.empty(n, dtype=torch.uint8, device="cuda:0")
ctx.use_torch_stream()
ctx.sample_and_validate_dev(1234, 0, n, se3, va)
torch.cuda.synchronize()
out = (C.c_ulonglong * 20)()
L.artp_debug_stage_cycles(out, 1)
ctx.sample_and_validate_dev(1234, n, n, se3, va)
torch.cuda.synchronize()
L.artp_debug_stage_cycles(out, 0)
a = np.array(list(out), dtype=np.float64).reshape(2, 10)
names = ["record", "scan", "vertex(f)", "compact", "corners"]
cnt = ctx.pipeline_counters()
print("counters", cnt)
ff = (C.c_ulonglong * 20)()
L.artp_debug_stage_cycles(ff, 3)  # reset >= 3: the feet_stream counters
print("feet_stream: stream ticks", ff[0], "corner ticks", ff[1], "boxes reaching the corner stage", ff[2])
ctx.sample_and_validate_dev(1234, n, n, se3, va)
torch.cuda.synchronize()
cc = (C.c_ulonglong * 20)()
L.artp_debug_stage_cycles(cc, 2)  # reset >= 2: read the classify phase counters instead (and reset all)
c = np.array(list(cc)[:16], dtype=np.float64).reshape(2, 8)
ph = ["PoseRec load+barrier", "head", "2 barriers", "stage+barrier", "tail: record + exact statistics", "tail: exits, probe, label", "queue+copy"]
for g, nm in ((0, "classify list waves"), (1, "classify early-exit waves")):
    tot = c[g, :7].sum()
    print(nm, int(c[g, 7]), "waves, ticks/wave", round(tot / max(c[g, 7], 1)),
          {k: round(float(c[g, j] / max(c[g, 7], 1))) for j, k in enumerate(ph)})
for g, nm in ((0, "torso G=64"), (1, "feet G=16")):
    tot = a[g, :5].sum()
    print(nm, "stage ticks/lifetime", round(tot / a[g, 8], 3), "groups", int(a[g, 9]), "ticks per group", a[g, 8] / a[g, 9],
          {names[k]: round(float(a[g, k] / tot), 3) for k in range(5)},
          "corners split", {n2: round(float(a[g, k] / tot), 3) for k, n2 in ((5, "candidates"), (6, "partners"), (7, "contacts"))})
