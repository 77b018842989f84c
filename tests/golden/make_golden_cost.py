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
import types

sys.dont_write_bytecode = True  # the reference tree is read-only: no __pycache__ next to its sources

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
import cost_query as ref_cost_query  # noqa: E402  (the reference's CostQuery: a stand-alone file, imported where it lies)
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


class _StubPredictor:
    """What CostQuery needs of predictor.CostPredictor (predictor.py:10-44; that class itself wants rospkg and a CUDA
    device): .network (featureResDownsampleFactor, mapClip), .features [1, C, Fh, Fw], .getPathCost -> FCpart.  float32 on
    the CPU instead of half on CUDA: the fixture pins the GATHER and the float32 costs, the half anchor is separate
    (make_golden_cost_anchor.py)."""

    def __init__(self, net, features):
        self.network = net
        self.features = features
        self.seen = None

    def getPathCost(self, features, tarInfo):
        self.seen = (features.clone(), tarInfo.clone())
        real_ones = torch.ones
        torch.ones = lambda *a, **k: real_ones(*a, **{kk: vv for kk, vv in k.items() if kk != "device"})  # FCpart hard-codes cuda
        try:
            with torch.no_grad():
                return self.network.FCpart(features.float(), tarInfo.float())
        finally:
            torch.ones = real_ones


def cost_query_fixture():
    """tests/golden/cost_query_ref.npz: rows / cols / costs straight out of the reference's CostQuery.setMapParams +
    __call__ (cost_query.py:26-35,39-69) -- float64 edge arrays as the server passes them (np.array of the request's
    Python floats, cost_query_server.py:131-136), starts far outside the map (both clamps), starts that land exactly on
    cell borders, and a NON-SQUARE map (length_x != length_y, features 36 x 32)."""
    params = mo.random_params(0)
    net = network_light.network().eval()
    sd = net.state_dict()
    for k in sd:
        if not k.endswith("num_batches_tracked"):
            sd[k] = torch.from_numpy(params[k].copy())
    net.load_state_dict(sd)
    torch.Tensor.cuda = lambda self, *a, **k: self   # CostQuery.__call__ moves its input to the GPU: there is none here

    gm = make_map(400, 0.04, seed=1234)
    elv = np.ascontiguousarray(gm["elevation"][::-1, ::-1]).astype(np.float32)
    res = 0.04
    out = {}
    rng = np.random.default_rng(17)
    for tag, (r0, c0, nr, nc) in (("sq", (140, 60, 112, 112)), ("rect", (30, 250, 120, 112))):
        crop = elv[r0:r0 + nr, c0:c0 + nc].astype(np.float16).astype(np.float32)
        with torch.no_grad():
            feats = net.CNNpart(torch.from_numpy(crop).view(1, 1, nr, nc))      # [1, 48, Fh, Fw] float32
        Fh, Fw = feats.shape[2], feats.shape[3]
        Lx, Ly = nr * res, nc * res
        B = 3000
        s = np.stack([rng.uniform(-Lx / 2, Lx / 2, B), rng.uniform(-Ly / 2, Ly / 2, B)], 1)
        s[:300] *= 3.0                                            # far outside the map: both ends of both clamps
        # starts exactly on feature-cell borders (and one float32 ulp either side): the truncation decides
        k = np.arange(300, 700)
        rb = int((Lx / res - 48) / 2 * 0.5)
        cell = rng.integers(1, Fh - 1, len(k))
        s[k, 0] = (cell - rb) * (2 * res)
        s[k[1::3], 0] = np.nextafter(s[k[1::3], 0].astype(np.float32), np.float32(np.inf))
        s[k[2::3], 0] = np.nextafter(s[k[2::3], 0].astype(np.float32), np.float32(-np.inf))
        d = rng.uniform(-0.6, 0.6, (B, 2))
        syaw, tyaw = rng.uniform(-np.pi, np.pi, B), rng.uniform(-np.pi, np.pi, B)
        edges32 = np.stack([s[:, 0] + d[:, 0], s[:, 1] + d[:, 1], tyaw, s[:, 0], s[:, 1], syaw], 1).astype(np.float32)
        edges64 = edges32.astype(np.float64)     # what the server's np.array(request.query_poses) holds
        # (1) the costs: the reference class end to end
        pred = _StubPredictor(net, feats)
        cq = ref_cost_query.CostQuery(pred, None)
        cq.setMapParams(res, Lx, Ly)
        energy, tim, risk = cq(edges64.copy())                   # __call__ subtracts in place
        costs = np.stack([energy, tim, risk], 1).astype(np.float32)
        # (2) the gather itself: a feature map that holds its own indices, through the same reference code
        code = torch.zeros((1, 2, Fh, Fw))
        code[0, 0] = torch.arange(Fh, dtype=torch.float32)[:, None].expand(Fh, Fw)
        code[0, 1] = torch.arange(Fw, dtype=torch.float32)[None, :].expand(Fh, Fw)
        spy = _StubPredictor(net, code)
        spy.getPathCost = lambda f, t: (f[:, 0:1], f[:, 1:2], None, f[:, 0:1])
        cq2 = ref_cost_query.CostQuery(spy, None)
        cq2.setMapParams(res, Lx, Ly)
        rr, cc, _ = cq2(edges64.copy())
        rows, cols = rr.astype(np.int64), cc.astype(np.int64)
        assert rows.min() == 1 and rows.max() == Fh - 2 and cols.min() == 1 and cols.max() == Fw - 2
        # the restatement against the reference, right here
        ro, co = mo.query_cells(edges32, res, Lx, Ly, (Fh, Fw))
        assert np.array_equal(ro, rows) and np.array_equal(co, cols), "oracle.query_cells != reference CostQuery"
        c_o = mo.fc_costs(params, feats[0].numpy(), edges32, res, Lx, Ly)
        print(tag, "features", (Fh, Fw), "row/col bias", cq.rowBias, cq.colBias, "costs max |oracle-ref|",
              np.abs(c_o - costs).max())
        assert np.abs(c_o - costs).max() < 1e-3
        out.update({f"{tag}_crop": crop.astype(np.float16), f"{tag}_edges": edges32, f"{tag}_rows": rows.astype(np.int32),
                    f"{tag}_cols": cols.astype(np.int32), f"{tag}_costs": costs,
                    f"{tag}_bias": np.array([cq.rowBias, cq.colBias], np.int32)})
    np.savez_compressed(os.path.join(HERE, "cost_query_ref.npz"), res=res, **out)


if __name__ == "__main__":
    main()
    cost_query_fixture()
