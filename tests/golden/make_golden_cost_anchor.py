#!/usr/bin/env python3
"""The reference network's OWN fp16-vs-fp32 error, as the anchor of the motion-cost tolerances.

Build container only (imports /root/reference/art_planner_motion_cost/.../network_light.py where it lies; nothing
is copied).  The reference runs the network in torch.half (predictor.py:22,34,44: `network.half()`, map and query
tensors `.half()`); the survey's bar (8c) compares against float32 torch.  This script measures, with the reference
class itself on CPU, how far the reference's half-precision evaluation is from its float32 evaluation -- per feature
and per edge cost, on exactly the inputs tests/test_motion_cost.py uses -- and stores the error STATISTICS (numbers
only) in motion_cost_fp16_anchor.json.  The GPU tests assert the HIP path (fp16 activations, fp32 accumulation) is
within 1.5x of these figures against the float32 result.
"""
import copy
import json
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


def fc(net, feats, edges, res, L, half):
    """CostQuery.__call__ (cost_query.py:39-69) on CPU; half=True follows predictor.getPathCost (.half() inputs)."""
    F = feats.shape[2]
    row, col = mo.query_cells(edges, res, L, L, F)
    ti = torch.from_numpy(edges.astype(np.float64))
    ti[:, :3] = ti[:, :3] - ti[:, 3:]
    f_t = feats[:, :, torch.from_numpy(row), torch.from_numpy(col)].squeeze(0).t().unsqueeze(-1).unsqueeze(-1)
    tgt = torch.cat((ti[:, :3], ti[:, 5:6]), dim=1).unsqueeze(-1).unsqueeze(-1)
    real_ones = torch.ones
    torch.ones = lambda *a, **k: real_ones(*a, **{kk: vv for kk, vv in k.items() if kk != "device"})  # FCpart hard-codes cuda
    try:
        with torch.no_grad():
            if half:
                power, tim, _, omp = net.FCpart(f_t.half(), tgt.half())
            else:
                power, tim, _, omp = net.FCpart(f_t.float(), tgt.float())
    finally:
        torch.ones = real_ones
    return torch.stack([power[:, 0, 0, 0], tim[:, 0, 0, 0], omp[:, 0, 0, 0]], 1).float().numpy()


def stats(net32, net16, elv, res, edges):
    n = elv.shape[0]
    x = torch.from_numpy(elv).view(1, 1, n, n)
    with torch.no_grad():
        f32 = net32.CNNpart(x)
        f16 = net16.CNNpart(x.half())
    fe = (f16.float() - f32).abs().numpy()
    c32 = fc(net32, f32, edges, res, n * res, False)
    c16 = fc(net16, f16, edges, res, n * res, True)
    ce = np.abs(c16 - c32)
    cex = (ce - 2e-3 * np.abs(c32)).max(axis=1)  # the excess over the survey's 2e-3 relative part, per edge
    return {"n": n, "feature_map": int(f32.shape[2]), "edges": int(len(edges)),
            "feat_abs_max": float(f32.abs().max()), "feat_err_max": float(fe.max()), "feat_err_mean": float(fe.mean()),
            "feat_err_q999": float(np.quantile(fe, 0.999)),
            "cost_err_max": float(ce.max()), "cost_err_mean": float(ce.mean()),
            "cost_err_per_output_max": [float(v) for v in ce.max(axis=0)],
            "cost_excess_over_2e-3_rel_q99": float(np.quantile(cex, 0.99)), "cost_excess_over_2e-3_rel_max": float(cex.max())}


def main():
    params = mo.random_params(0)
    net = network_light.network().eval()
    sd = net.state_dict()
    for k in sd:
        if not k.endswith("num_batches_tracked"):
            sd[k] = torch.from_numpy(params[k].copy())
    net.load_state_dict(sd)
    net16 = copy.deepcopy(net).half()
    out = {"what": "reference network_light.network on CPU: torch.half evaluation (predictor.py:22,34,44) vs its own float32 "
                   "evaluation, same weights (convert_weights.random_params(0)), same inputs as tests/test_motion_cost.py",
           "torch": torch.__version__, "cases": {}}
    g = np.load(os.path.join(HERE, "motion_cost.npz"))
    out["cases"]["golden_112"] = stats(net, net16, g["crop"].astype(np.float32), float(g["res"]), g["edges"])
    for n in (400, 800, 141, 97):   # test_gpu_features_match_oracle_at_c3_c4_and_odd_sizes: same maps, same edges
        gm = make_map(n, 0.04, seed=1234 if n == 400 else 77)
        elv = np.ascontiguousarray(gm["elevation"][::-1, ::-1]).astype(np.float16).astype(np.float32)
        rng = np.random.default_rng(n)
        B = 20000
        s = rng.uniform(-0.55 * gm.len_x, 0.55 * gm.len_x, (B, 2))
        d = rng.uniform(-0.6, 0.6, (B, 2))
        e = np.stack([s[:, 0] + d[:, 0], s[:, 1] + d[:, 1], rng.uniform(-np.pi, np.pi, B), s[:, 0], s[:, 1],
                      rng.uniform(-np.pi, np.pi, B)], 1).astype(np.float32)
        out["cases"][f"map_{n}"] = stats(net, net16, elv, gm.res, e)
        print(n, json.dumps(out["cases"][f"map_{n}"]))
    print("golden_112", json.dumps(out["cases"]["golden_112"]))
    json.dump(out, open(os.path.join(HERE, "motion_cost_fp16_anchor.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
