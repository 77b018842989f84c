"""Tuning aid: a few sample + validate batches on the C2 map with a contiguous unknown band (for a kernel trace)."""
import os, sys, copy
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests")): sys.path.insert(0, p)
import numpy as np, torch
from art_planner_amd.context import Context
from synthetic import make_map
dev = torch.device("cuda", 0)
gm = make_map(400, 0.04, seed=1234)
mode = sys.argv[1] if len(sys.argv) > 1 else "band"
e, m = gm["elevation"].copy(), gm["elevation_masked"].copy()
if mode == "band":
    e[150:250, :] = np.nan; m[150:250, :] = np.nan
elif mode == "border":
    for a in (e, m):
        a[:60, :] = np.nan; a[-60:, :] = np.nan; a[:, :60] = np.nan; a[:, -60:] = np.nan
else:
    rng = np.random.default_rng(0)
    e[rng.random(e.shape) < 0.01] = np.nan; m[rng.random(e.shape) < 0.01] = np.nan
gm.layers["elevation"] = np.asfortranarray(e); gm.layers["elevation_masked"] = np.asfortranarray(m)
ctx = Context(0, "yaml"); ctx.upload_map(gm); ctx.use_torch_stream()
S = 1 << 22
se3 = torch.empty((S, 7), dtype=torch.float64, device=dev); valid = torch.empty(S, dtype=torch.uint8, device=dev)
for i in range(4): ctx.sample_and_validate_dev(42, i * S, S, se3, valid)
torch.cuda.synchronize()
print("done", ctx.pipeline_counters())
