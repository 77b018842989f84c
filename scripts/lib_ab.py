"""Tuning aid: warm step time (median of batches 41-80), label hash and the edge batch of the bench for a build variant:
ARTP_LIB=<variant .so> python scripts/lib_ab.py"""
import hashlib, os, sys, statistics
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests")): sys.path.insert(0, p)
import numpy as np, torch
import bench
from art_planner_amd.context import Context
from synthetic import map_from_device, raw_map
dev = torch.device("cuda", 0)
ctx = Context(0, "yaml"); map_from_device(ctx, raw_map(400, 0.04, seed=1234)); ctx.use_torch_stream()
S = 1 << 22
se3 = torch.empty((S, 7), dtype=torch.float64, device=dev); valid = torch.empty(S, dtype=torch.uint8, device=dev)
ctx.sample_and_validate_dev(42, 0, S, se3, valid); torch.cuda.synchronize()
h = hashlib.sha1(valid.cpu().numpy().tobytes()).hexdigest()[:16]
acc = se3.cpu().numpy()[valid.cpu().numpy() != 0]
n = 80
evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
evs[0].record()
for i in range(n):
    ctx.sample_and_validate_dev(42, i * S, S, se3, valid); evs[i + 1].record()
torch.cuda.synchronize()
ms = [evs[i].elapsed_time(evs[i + 1]) for i in range(n)]
ii, jj = bench.pair_edges(acc, 1 << 18)
s1 = torch.from_numpy(np.ascontiguousarray(acc[ii])).to(dev); s2 = torch.from_numpy(np.ascontiguousarray(acc[jj])).to(dev)
ev = torch.empty(len(ii), dtype=torch.uint8, device=dev)
for _ in range(3): ctx.check_motions_dev(s1, s2, ev)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(8): ctx.check_motions_dev(s1, s2, ev)
e1.record(); torch.cuda.synchronize()
he = hashlib.sha1(ev.cpu().numpy().tobytes()).hexdigest()[:12]
print(f"{os.environ.get('ARTP_LIB', 'default'):45s} warm step {statistics.median(ms[40:]):.4f} ms  labels {h}  checkMotion {e0.elapsed_time(e1) / 8:.4f} ms  verdicts {he}")
