#!/usr/bin/env python3
"""Golden vectors for the learned motion cost, produced by the REFERENCE network class.

Build container only (imports /root/reference/art_planner_motion_cost/.../network_light.py; the file is
imported where it lies, nothing is copied).  Stores data only: a 112x112 elevation crop, the reference
feature map (CNNpart, float32 on CPU) and FCpart outputs for 4096 seeded edges, for the seeded parameters
of oracle/motion_cost_oracle.random_params(0); and a second, 120x120 crop (-> 36x36 features: partial tiles in
the HIP kernels) with its reference feature map (motion_cost_120.npz).
"""
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, "/root/reference/art_planner_motion_cost/src/art_planner_motion_cost/predictor")
import motion_cost_oracle as mo  # noqa: E402
import network_light  # noqa: E402  (the reference)
from synthetic import make_map  # noqa: E402


def main():
    params = mo.random_params(0)
    net = network_light.network().eval()
    sd = net.state_dict()
    for k in sd:
        if k.endswith("num_batches_tracked"):
            continue
        sd[k] = torch.from_numpy(params[k].copy())
    net.load_state_dict(sd)

    gm = make_map(400, 0.04, seed=1234)
    layer = gm["elevation"]
    # server convention (cost_query_server.py:66-74): elvMap[a, b] = layer(N-1-a, N-1-b)
    elv = np.ascontiguousarray(layer[::-1, ::-1]).astype(np.float32)
    crop = elv[140:252, 60:172].astype(np.float16).astype(np.float32)  # exactly representable in fp16
    n, res = 112, 0.04
    with torch.no_grad():
        feats = net.CNNpart(torch.from_numpy(crop).view(1, 1, n, n)).numpy()[0]  # [48,32,32] float32

    rng = np.random.default_rng(5)
    B = 4096
    L = n * res
    s = rng.uniform(-L / 2, L / 2, (B, 2))
    d = rng.uniform(-0.6, 0.6, (B, 2))
    syaw = rng.uniform(-np.pi, np.pi, B)
    tyaw = rng.uniform(-np.pi, np.pi, B)
    edges = np.stack([s[:, 0] + d[:, 0], s[:, 1] + d[:, 1], tyaw, s[:, 0], s[:, 1], syaw], 1).astype(np.float32)

    # CostQuery.__call__ (cost_query.py:39-69) on CPU, float32
    F = feats.shape[1]
    row, col = mo.query_cells(edges, res, L, L, F)
    ti = torch.from_numpy(edges.astype(np.float64))
    ti[:, :3] = ti[:, :3] - ti[:, 3:]
    f_t = torch.from_numpy(feats)[None][:, :, torch.from_numpy(row), torch.from_numpy(col)]
    f_t = f_t.squeeze(0).t().unsqueeze(-1).unsqueeze(-1)
    tgt = torch.cat((ti[:, :3], ti[:, 5:6]), dim=1).unsqueeze(-1).unsqueeze(-1).float()
    real_ones = torch.ones
    torch.ones = lambda *a, **k: real_ones(*a, **{kk: vv for kk, vv in k.items() if kk != "device"})  # FCpart hard-codes cuda
    with torch.no_grad():
        power, tim, _, omp = net.FCpart(f_t.float(), tgt)
    torch.ones = real_ones
    costs = torch.stack([power[:, 0, 0, 0], tim[:, 0, 0, 0], omp[:, 0, 0, 0]], 1).numpy().astype(np.float32)

    # cross-check the numpy restatement right here
    f_o = mo.cnn_features(params, crop)
    c_o = mo.fc_costs(params, feats, edges, res, L, L)
    print("features: ref range", feats.min(), feats.max(), "max |oracle-ref|", np.abs(f_o - feats).max())
    print("costs: max |oracle-ref|", np.abs(c_o - costs).max(), "mean", costs.mean(0))
    assert np.abs(f_o - feats).max() < 2e-3 * max(1.0, np.abs(feats).max())
    assert np.abs(c_o - costs).max() < 1e-3
    np.savez_compressed(os.path.join(HERE, "motion_cost.npz"), crop=crop.astype(np.float16), res=res,
                        features=feats.astype(np.float32), edges=edges, costs=costs)

    # a size whose feature map has PARTIAL tiles in the HIP kernels (8-row x 16-pixel output tiles of the 15x15
    # layer, 32-pixel wavefront tiles of the 3x3 layers): 120 x 120 -> 36 x 36 features
    n2 = 120
    crop2 = elv[30:30 + n2, 250:250 + n2].astype(np.float16).astype(np.float32)
    with torch.no_grad():
        feats2 = net.CNNpart(torch.from_numpy(crop2).view(1, 1, n2, n2)).numpy()[0]  # [48,36,36]
    assert feats2.shape == (48, 36, 36)
    f2_o = mo.cnn_features(params, crop2)
    print("120x120: max |oracle-ref|", np.abs(f2_o - feats2).max())
    assert np.abs(f2_o - feats2).max() < 2e-3 * max(1.0, np.abs(feats2).max())
    np.savez_compressed(os.path.join(HERE, "motion_cost_120.npz"), crop=crop2.astype(np.float16), res=res,
                        features=feats2.astype(np.float32))


if __name__ == "__main__":
    main()
