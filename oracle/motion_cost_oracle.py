"""CPU ORACLE (test infrastructure only) for the learned motion cost: numpy float32 restatement of
network.CNNpart / network.FCpart (art_planner_motion_cost/src/art_planner_motion_cost/predictor/
network_light.py:78-165) and CostQuery (cost_query.py:26-69).

Parity pin: tests/golden/make_golden_cost.py imports the REFERENCE network class in the build container,
loads the seeded parameters of random_params() into it and stores its outputs; tests compare this
restatement (CPU suite) and the HIP path (GPU suite) with those vectors.  The trained weights are git-LFS
stubs in the reference checkout, so parity is on seeded random weights only.
"""
import numpy as np

import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
from convert_weights import SHAPES, WITH_BIAS, random_params  # noqa: E402,F401  (seeded test weights)


def _conv(x, w):
    """Un-padded cross-correlation, x [C,H,W], w [O,C,kh,kw] -> [O,H-kh+1,W-kw+1] (float32)."""
    O, C, kh, kw = w.shape
    H, W = x.shape[1] - kh + 1, x.shape[2] - kw + 1
    out = np.zeros((O, H, W), np.float32)
    for dy in range(kh):
        for dx in range(kw):
            out += np.tensordot(w[:, :, dy, dx], x[:, dy:dy + H, dx:dx + W], axes=([1], [0])).astype(np.float32)
    return out


def _bn(x, p, name):
    g, b = p[name + "_bn.weight"], p[name + "_bn.bias"]
    m, v = p[name + "_bn.running_mean"], p[name + "_bn.running_var"]
    s = (g / np.sqrt(v + np.float32(1e-5))).astype(np.float32)
    return (x - m[:, None, None]) * s[:, None, None] + b[:, None, None]


def _lrelu(x):
    return np.where(x > 0, x, np.float32(0.3) * x).astype(np.float32)


def _maxpool(x, k, s):
    H, W = (x.shape[1] - k) // s + 1, (x.shape[2] - k) // s + 1
    out = np.full((x.shape[0], H, W), -np.inf, np.float32)
    for dy in range(k):
        for dx in range(k):
            out = np.maximum(out, x[:, dy:dy + s * (H - 1) + 1:s, dx:dx + s * (W - 1) + 1:s])
    return out


def cnn_features(p, elev):
    """network.CNNpart: elev [H,W] (index a along x, b along y) -> features [48, F, F]."""
    t = elev.astype(np.float32)[None]
    t = _bn(_conv(t, p["init_conv1.weight"]), p, "init_conv1")
    t = _lrelu(_bn(_conv(t, p["init_conv2.weight"]), p, "init_conv2"))
    t = _maxpool(t, 2, 2)
    t = _lrelu(_bn(_conv(t, p["init_conv3.weight"]), p, "init_conv3"))
    t = _lrelu(_bn(_conv(t, p["init_conv4.weight"]), p, "init_conv4"))
    t = _maxpool(t, 3, 1)
    t = _lrelu(_bn(_conv(t, p["init_conv5.weight"]), p, "init_conv5"))
    t = _lrelu(_bn(_conv(t, p["init_flatten.weight"]), p, "init_flatten"))
    return t


def query_cells(edges, res, len_x, len_y, F, cx=0.0, cy=0.0):
    """CostQuery.setMapParams + the index arithmetic of __call__ (cost_query.py:26-35,54-55): float64 arithmetic (the
    server hands float64 numpy to torch, cost_query_server.py:131-136), clamp to [1, shape - 2], .long() truncation.
    F = features.shape[2] for a square feature map, or (shape[2], shape[3]).  Pinned on the reference's own CostQuery
    by tests/golden/cost_query_ref.npz (make_golden_cost.py imports cost_query.py where it lies)."""
    Fh, Fw = (F, F) if np.isscalar(F) else F
    feat_res = res * 2
    row_bias = int((len_x / res - 2 * 24) / 2 * 0.5)
    col_bias = int((len_y / res - 2 * 24) / 2 * 0.5)
    e = np.asarray(edges, np.float64)
    row = np.clip((e[:, 3] - cx) / feat_res + row_bias, 1, Fh - 2).astype(np.int64)
    col = np.clip((e[:, 4] - cy) / feat_res + col_bias, 1, Fw - 2).astype(np.int64)
    return row, col


def fc_costs(p, feats, edges, res, len_x, len_y, cx=0.0, cy=0.0):
    """CostQuery.__call__ + network.FCpart: edges [B,6] -> [B,3] = energy, time, 1 - prob."""
    row, col = query_cells(edges, res, len_x, len_y, (feats.shape[1], feats.shape[2]), cx, cy)
    f = feats[:, row, col].T.astype(np.float32)  # [B,48]
    e = np.asarray(edges, np.float32)
    d = e[:, :3] - e[:, 3:]
    dx, dy, dyaw, syaw = d[:, 0], d[:, 1], d[:, 2].copy(), e[:, 5]
    pi = np.float32(np.pi)
    dyaw = np.where(dyaw > pi, dyaw - 2 * pi, dyaw)
    dyaw = np.where(dyaw < -pi, dyaw + 2 * pi, dyaw)
    tar = np.stack([dx, dy, np.sqrt(dx * dx + dy * dy), np.arctan2(dy, dx), dyaw, np.cos(dyaw), np.sin(dyaw),
                    syaw, np.cos(syaw), np.sin(syaw)], 1).astype(np.float32)

    def lin_bn(x, name):
        w = p[name + ".weight"].reshape(p[name + ".weight"].shape[0], -1)
        y = x @ w.T
        g, b = p[name + "_bn.weight"], p[name + "_bn.bias"]
        m, v = p[name + "_bn.running_mean"], p[name + "_bn.running_var"]
        return ((y - m) * (g / np.sqrt(v + np.float32(1e-5))) + b).astype(np.float32)

    def out(x, name):
        return x @ p[name + ".weight"].reshape(1, -1).T + p[name + ".bias"]

    t = lin_bn(tar, "tar0_conv1")
    h = _lrelu(lin_bn(np.concatenate([f, t], 1), "out0_conv1"))
    power = np.maximum(out(_lrelu(lin_bn(h, "out1_conv1")), "out2_conv1"), 0)
    tim = np.maximum(out(_lrelu(lin_bn(h, "out1_conv2")), "out2_conv2"), 0)
    prob = 1.0 / (1.0 + np.exp(-out(_lrelu(lin_bn(h, "out1_conv3")), "out2_conv3")))
    return np.concatenate([power, tim, 1.0 - prob], 1).astype(np.float32)
