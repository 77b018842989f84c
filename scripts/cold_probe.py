"""Tuning aid: per-batch time of the fused sample + validate step from process start (HIP events around every batch), to
see how the first batches of a process differ from the steady state."""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests")): sys.path.insert(0, p)
import torch
from art_planner_amd.context import Context
from synthetic import map_from_device, raw_map
dev = torch.device("cuda", 0)
ctx = Context(0, "yaml"); map_from_device(ctx, raw_map(400, 0.04, seed=1234)); ctx.use_torch_stream()
S = 1 << 22
se3 = torch.empty((S, 7), dtype=torch.float64, device=dev); valid = torch.empty(S, dtype=torch.uint8, device=dev)
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
if mode == "spin":   # keep the GPU busy with something else for a while first
    a = torch.randn(8192, 8192, device=dev, dtype=torch.float16)
    for _ in range(200): b = a @ a
    torch.cuda.synchronize()
if mode == "idle":
    import time; ctx.sample_and_validate_dev(42, 0, S, se3, valid); torch.cuda.synchronize(); time.sleep(2.0)
n = 80
evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
evs[0].record()
for i in range(n):
    ctx.sample_and_validate_dev(42, i * S, S, se3, valid)
    evs[i + 1].record()
torch.cuda.synchronize()
ms = [evs[i].elapsed_time(evs[i + 1]) for i in range(n)]
print(mode, " ".join(f"{x:.3f}" for x in ms))
