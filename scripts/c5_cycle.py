"""BASELINE config 5 cycle alone (what bench.py reports as replan_cycle_c5), with per-stage device times."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import numpy as np, torch
import convert_weights
from art_planner_amd.context import Context
from synthetic import map_from_device, raw_map
dev = torch.device("cuda", 0)
ctx = Context(0, "yaml")
gm = map_from_device(ctx, raw_map(400, 0.04, seed=1234, with_upper_bound=True), body_layer="upper_bound")
ctx.use_torch_stream()
ctx.cost_load_weights(convert_weights.to_blob(convert_weights.random_params(0)))
rng = np.random.default_rng(55)
ub, mk = gm["upper_bound"].copy(order="F"), gm["elevation_masked"].copy(order="F")
n5 = 1 << 18
se = torch.empty((n5, 7), dtype=torch.float64, device=dev)
va = torch.empty(n5, dtype=torch.uint8, device=dev)
rows = torch.rand((50000, 6), device=dev, dtype=torch.float32) * 4 - 2
cost = torch.empty((50000, 3), dtype=torch.float32, device=dev)
side = int(round(np.sqrt(0.05 * gm.rows * gm.cols / 3)))
elv_dev = torch.from_numpy(np.ascontiguousarray(ub[::-1, ::-1])).to(dev)
ctx.cost_update_map_dev(elv_dev, gm.res, gm.len_x, gm.len_y)
ctx.sample_and_validate_dev(42, 0, n5, se, va)
torch.cuda.synchronize()
cyc, host_rects, dev_rects = [], [], []
for it in range(100):
    t0 = time.perf_counter()
    org = []
    for _ in range(3):
        r0, c0 = int(rng.integers(0, gm.rows - side)), int(rng.integers(0, gm.cols - side))
        ub[r0:r0 + side, c0:c0 + side] += np.float32(rng.normal(0, 0.01))
        m_ = mk[r0:r0 + side, c0:c0 + side]
        mk[r0:r0 + side, c0:c0 + side] = np.where(np.isfinite(m_), ub[r0:r0 + side, c0:c0 + side], m_)
        org.append((r0, c0))
    ta = time.perf_counter()
    ctx.update_layer_rects(0, [ub[r0:r0 + side, c0:c0 + side] for r0, c0 in org], org)
    ctx.update_layer_rects(1, [mk[r0:r0 + side, c0:c0 + side] for r0, c0 in org], org)
    tb = time.perf_counter()
    torch.cuda.synchronize()
    tc = time.perf_counter()
    ctx.cost_update_map(np.ascontiguousarray(ub[::-1, ::-1]), gm.res, gm.len_x, gm.len_y)
    ctx.sample_and_validate_dev(42, 7_000_000 + it * n5, n5, se, va)
    ctx.cost_query_dev(rows, cost)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    cyc.append((t1 - ta) * 1e3)
    host_rects.append((tb - ta) * 1e3)
    dev_rects.append((tc - ta) * 1e3)
print(f"cycle median {np.median(cyc):.3f} ms (max {np.max(cyc):.3f}); rect updates: host {np.median(host_rects):.3f} ms, "
      f"until the device is done {np.median(dev_rects):.3f} ms")
ctx.close()
