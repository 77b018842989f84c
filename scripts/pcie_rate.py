"""PCIe-inclusive rate of the hot path when the boundary hands over HOST buffers (artp_sample_and_validate: states and labels
come back over PCIe; DESIGN.md 6 -- never `value`)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from art_planner_amd.context import Context
from synthetic import map_from_device, raw_map
ctx = Context(0, "yaml"); gm = map_from_device(ctx, raw_map(400, 0.04, seed=1234))
n = 1 << 22
ctx.sample_and_validate(42, 0, n)
t0 = time.perf_counter()
for i in range(3):
    se3, valid = ctx.sample_and_validate(42, (i + 1) * n, n)
dt = (time.perf_counter() - t0) / 3
print(f"artp_sample_and_validate, host buffers, 2^22 states: {dt * 1e3:.2f} ms per batch = {n / dt:.3e} states/s "
      f"({(n * 57) / dt / 1e9:.1f} GB/s of results over PCIe into pageable numpy buffers)")
import numpy as np
lab = ctx.validate_states(se3)     # host states in, labels out: 235 MB up, 4 MB down
t0 = time.perf_counter()
for i in range(3):
    lab = ctx.validate_states(se3)
dt = (time.perf_counter() - t0) / 3
print(f"artp_validate_states, host states in / labels out, 2^22 states: {dt * 1e3:.2f} ms per batch = {n / dt:.3e} states/s")
ctx.close()
