"""Tuning aid: the 0.5 m interpolation rule alone on the bench's edge batch (for a kernel trace of that path:
   rocprofv3 --kernel-trace --stats -d out/trace -o trace -- python scripts/interp_only.py)."""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in (ROOT, os.path.join(ROOT, "tests")): sys.path.insert(0, p)
import numpy as np, torch, time
import bench
from art_planner_amd.context import Context
from synthetic import map_from_device, raw_map
dev = torch.device("cuda", 0)
ctx = Context(0, "yaml"); map_from_device(ctx, raw_map(400, 0.04, seed=1234)); ctx.use_torch_stream()
S = 1 << 22
se3 = torch.empty((S, 7), dtype=torch.float64, device=dev); valid = torch.empty(S, dtype=torch.uint8, device=dev)
ctx.sample_and_validate_dev(42, 0, S, se3, valid); torch.cuda.synchronize()
acc = se3.cpu().numpy()[valid.cpu().numpy() != 0]
ii, jj = bench.pair_edges(acc, 1 << 18)
s1 = torch.from_numpy(np.ascontiguousarray(acc[ii])).to(dev); s2 = torch.from_numpy(np.ascontiguousarray(acc[jj])).to(dev)
ev = torch.empty(len(ii), dtype=torch.uint8, device=dev); ni = torch.empty(len(ii), dtype=torch.int32, device=dev)
for _ in range(3): ctx.check_edges_interp_dev(s1, s2, ev, ni)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): ctx.check_edges_interp_dev(s1, s2, ev, ni)
torch.cuda.synchronize()
print("INTERP wall ms/call", (time.perf_counter() - t0) / 20 * 1e3, "states", int(ni.sum().item()))
