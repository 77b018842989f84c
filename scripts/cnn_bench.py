#!/usr/bin/env python3
"""Feature-extractor timing (kernels only, HIP events on the context's stream) at C3 (400^2) and C4 (800^2).
   python scripts/cnn_bench.py [reps]      ARTP_CNN_UNFUSED=1 selects the round-2 one-kernel-per-layer path."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import numpy as np
import torch
import bench
import convert_weights
from art_planner_amd.context import Context
from synthetic import raw_map

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)
ctx = Context(0, "yaml")
ctx.cost_load_weights(convert_weights.to_blob(convert_weights.random_params(0)))
st = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(st)
ctx.use_torch_stream()
for n, seed in ((400, 1234), (800, 77)):
    g = raw_map(n, 0.04, seed=seed)
    elv = torch.from_numpy(np.ascontiguousarray(g["elevation"][::-1, ::-1]).astype(np.float32)).to(dev)
    if os.environ.get("ARTP_BENCH_FLAT") == "1":   # DVFS probe: a constant map (every activation of a layer the same value)
        elv.zero_()
    for _ in range(3):
        ctx.cost_update_map_dev(elv, g.res, g.len_x, g.len_y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ctx.cost_update_map_dev(elv, g.res, g.len_x, g.len_y)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    gf = bench.cnn_flops(n) / 1e9
    print(f"map {n}: {ms * 1e3:.1f} us per feature map, {gf / ms:.0f} TFLOP/s algorithmic = "
          f"{gf / ms / bench.MFMA_F16_PEAK_TFLOPS:.3f} of the dense f16 peak "
          f"({'unfused' if os.environ.get('ARTP_CNN_UNFUSED') == '1' else 'fused'})")
ctx.close()
