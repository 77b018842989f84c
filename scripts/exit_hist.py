"""Histogram of the dCollideHeightfieldZone exit that decides each box (bench workload C2).
Diagnostic for kernel tuning: which stages the undecided boxes end in."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from art_planner_amd.context import Context
from synthetic import make_map

gm = make_map(400, 0.04, seed=1234)
ctx = Context(0, "yaml")
ctx.upload_map(gm)
se3 = ctx.sample_states(1234, 0, 1 << 18)
valid, det = ctx.validate_states(se3, want_detail=True)
names = ["aabb_off", "above", "under", "spans", "flat", "vertex", "plane", "vertex2", "none"]
print("valid frac", valid.mean())
for col, nm in ((0, "torso"), (1, "foot1"), (2, "foot2")):
    d = det[:, col]
    ev = d >= 0
    print(nm, "evaluated", int(ev.sum()), {names[k]: int((d == k).sum()) for k in range(9) if (d == k).any()},
          "other", {int(k): int((d == k).sum()) for k in np.unique(d) if k < 0 or k > 8})
print("result col", {int(k): int((det[:, 5] == k).sum()) for k in np.unique(det[:, 5])})
