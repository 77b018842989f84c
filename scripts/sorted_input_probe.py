"""How much faster is the validity pipeline on spatially ordered states?  (tuning probe: tile binning pays off
only if the gap is large).  Times artp_validate_states_dev on the bench batch as sampled, fully sorted by cell,
and binned by square tiles of a few sizes."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from art_planner_amd.context import Context
from synthetic import make_map

gm = make_map(400, 0.04, seed=1234)
ctx = Context(0, "yaml")
ctx.upload_map(gm)
dev = torch.device("cuda:0")
n = 1 << 22
se3 = torch.empty((n, 7), dtype=torch.float64, device=dev)
va = torch.empty(n, dtype=torch.uint8, device=dev)
ctx.use_torch_stream()
ctx.sample_states_dev(42, 0, n, se3)
torch.cuda.synchronize()


def timed(states, reps=10):
    ctx.validate_states_dev(states, va)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ctx.validate_states_dev(states, va)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


ri = ((gm.pos_x + 0.5 * gm.len_x - se3[:, 0]) / gm.res).long().clamp(0, gm.rows - 1)
ci = ((gm.pos_y + 0.5 * gm.len_y - se3[:, 1]) / gm.res).long().clamp(0, gm.cols - 1)
print("as sampled      %.3f ms" % timed(se3), "valid", int(va.sum()))
ref = va.clone()
for name, key in (("cell sorted", ci * gm.rows + ri),
                  ("tiles 8x8 ", (ci // 8) * 64 + ri // 8),
                  ("tiles 16x16", (ci // 16) * 32 + ri // 16),
                  ("tiles 32x32", (ci // 32) * 16 + ri // 32),
                  ("tiles 64x64", (ci // 64) * 8 + ri // 64)):
    perm = torch.argsort(key, stable=True)
    s2 = se3[perm].contiguous()
    t = timed(s2)
    assert torch.equal(va, ref[perm])
    print("%-15s %.3f ms" % (name, t), ctx.pipeline_counters())
