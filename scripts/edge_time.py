import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import bench
from art_planner_amd.context import Context
from synthetic import map_from_device, raw_map
dev = torch.device("cuda", 0)
ctx = Context(0, "yaml")
map_from_device(ctx, raw_map(400, 0.04, seed=1234))
for use_side in (False, True):
    if use_side:
        st_ = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st_)
    ctx.use_torch_stream()
    n0 = 1 << 20
    se3 = torch.empty((n0, 7), dtype=torch.float64, device=dev)
    valid = torch.empty(n0, dtype=torch.uint8, device=dev)
    ctx.sample_and_validate_dev(42, 0, n0, se3, valid)
    torch.cuda.synchronize()
    st = se3.cpu().numpy(); acc = st[valid.cpu().numpy() != 0]
    ii, jj = bench.pair_edges(acc, 1 << 18)
    s1 = torch.from_numpy(np.ascontiguousarray(acc[ii])).to(dev); s2 = torch.from_numpy(np.ascontiguousarray(acc[jj])).to(dev)
    ev = torch.empty(len(ii), dtype=torch.uint8, device=dev)
    ctx.check_motions_dev(s1, s2, ev); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(5): ctx.check_motions_dev(s1, s2, ev)
    e1.record(); torch.cuda.synchronize(); t1 = time.perf_counter()
    print("side stream" if use_side else "default stream", "events ms/call", e0.elapsed_time(e1) / 5, "wall ms/call", (t1 - t0) / 5 * 1e3)
    # per-phase host timing of one call
    t0 = time.perf_counter(); ctx.check_motions_dev(s1, s2, ev); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("   host call ms", (t1 - t0) * 1e3, "then sync ms", (t2 - t1) * 1e3)
