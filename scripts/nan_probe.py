"""Tuning aid: step time of sample + validate on the C2 map with unknown (NaN) cells: 1 % scattered, and a contiguous
unknown band -- how far the fallback kernels (ordered scans with ODE's running-dMAX quirk) fall behind the table path."""
import os, sys, copy, statistics
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests")): sys.path.insert(0, p)
import numpy as np, torch
from art_planner_amd.context import Context
from synthetic import make_map
dev = torch.device("cuda", 0)
base = make_map(400, 0.04, seed=1234)
rng = np.random.default_rng(0)
S = 1 << 22
for name in ("plain", "nan_1pct", "nan_band", "nan_border"):
    gm = copy.deepcopy(base)
    e, m = gm["elevation"].copy(), gm["elevation_masked"].copy()
    if name == "nan_1pct":
        e[rng.random(e.shape) < 0.01] = np.nan; m[rng.random(e.shape) < 0.01] = np.nan
    if name == "nan_band":
        e[150:250, :] = np.nan; m[150:250, :] = np.nan
    if name == "nan_border":
        for a in (e, m):
            a[:60, :] = np.nan; a[-60:, :] = np.nan; a[:, :60] = np.nan; a[:, -60:] = np.nan
    gm.layers["elevation"] = np.asfortranarray(e); gm.layers["elevation_masked"] = np.asfortranarray(m)
    ctx = Context(0, "yaml"); ctx.upload_map(gm); ctx.use_torch_stream()
    se3 = torch.empty((S, 7), dtype=torch.float64, device=dev); valid = torch.empty(S, dtype=torch.uint8, device=dev)
    for i in range(20): ctx.sample_and_validate_dev(42, i * S, S, se3, valid)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(20): ctx.sample_and_validate_dev(42, i * S, S, se3, valid)
    e1.record(); torch.cuda.synchronize()
    print(f"{name:12s} {e0.elapsed_time(e1) / 20:8.3f} ms per 2^22 states  valid {float(valid.float().mean()):.3f}  counters {ctx.pipeline_counters()}", flush=True)
    ctx.close()
