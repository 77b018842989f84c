"""Tuning probe: does running the two halves of a batch on two streams (two contexts, same map) beat one stream?
The small serial kernels at the end of the pipeline (partner pass, exact grouping: ~75 us for a few dozen boxes)
and kernels with complementary bottlenecks (sampler: L1 miss queue, stream kernels: VALU) could then overlap."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from art_planner_amd.context import Context
from synthetic import make_map

gm = make_map(400, 0.04, seed=1234)
n = 1 << 22
dev = "cuda:0"
se3 = torch.empty((n, 7), dtype=torch.float64, device=dev)
va = torch.empty(n, dtype=torch.uint8, device=dev)


def run(parts, steps=20):
    ctxs = [Context(0, "yaml") for _ in range(parts)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(parts)]
    for c, s in zip(ctxs, streams):
        c.upload_map(gm)
        with torch.cuda.stream(s):
            c.use_torch_stream()
    torch.cuda.synchronize()
    m = n // parts

    def step(k):
        for p, (c, s) in enumerate(zip(ctxs, streams)):
            with torch.cuda.stream(s):
                c.sample_and_validate_dev(1234, k * n + p * m, m, se3[p * m:(p + 1) * m], va[p * m:(p + 1) * m])
    for k in range(3):
        step(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        step(3 + k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print("parts %d: %.3f ms per 2^22 states (%.4g states/s), valid %.4f" % (parts, dt * 1e3, n / dt, va.float().mean().item()))
    for c in ctxs:
        c.close()


for parts in (1, 2, 4):
    run(parts)
