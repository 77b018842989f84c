#!/usr/bin/env python3
"""Edge entry points on the bench's edge batch: ms per call (HIP events) and a hash of the verdicts (A/B of kernel changes)."""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import bench
from art_planner_amd.context import Context
from synthetic import map_from_device, raw_map
dev = torch.device("cuda", 0)
ctx = Context(0, "yaml")
map_from_device(ctx, raw_map(400, 0.04, seed=1234))
ctx.use_torch_stream()
S = 1 << 22
se3 = torch.empty((S, 7), dtype=torch.float64, device=dev); valid = torch.empty(S, dtype=torch.uint8, device=dev)
ctx.sample_and_validate_dev(42, 0, S, se3, valid); torch.cuda.synchronize()
acc = se3.cpu().numpy()[valid.cpu().numpy() != 0]
ii, jj = bench.pair_edges(acc, 1 << 18)
s1 = torch.from_numpy(np.ascontiguousarray(acc[ii])).to(dev); s2 = torch.from_numpy(np.ascontiguousarray(acc[jj])).to(dev)
E = len(ii)
ev = torch.empty(E, dtype=torch.uint8, device=dev); lt = torch.empty(E, dtype=torch.float64, device=dev)
ls = torch.empty((E, 7), dtype=torch.float64, device=dev); ni = torch.empty(E, dtype=torch.int32, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for name, fn, extra in (("check_motion", lambda: ctx.check_motions_dev(s1, s2, ev), None),
                        ("interp_0p5m", lambda: ctx.check_edges_interp_dev(s1, s2, ev, ni), ni),
                        ("last_valid", lambda: ctx.check_motions_last_valid_dev(s1, s2, ev, lt, ls), lt)):
    fn(); torch.cuda.synchronize()
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    h = hashlib.sha1(ev.cpu().numpy().tobytes() + (extra.cpu().numpy().tobytes() if extra is not None else b"")).hexdigest()[:16]
    ms = e0.elapsed_time(e1) / 5
    print(f"{name:14s} {ms:8.4f} ms  {E / ms / 1e3:9.1f} M edges/s  valid {float(ev.float().mean()):.6f}  hash {h}")
