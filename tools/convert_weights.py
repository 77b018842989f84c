#!/usr/bin/env python3
"""Convert a motion-cost network state_dict (art_planner_motion_cost/.../network_light.py:19-62) into the
flat float32 blob artp_cost_load_weights() expects (include/artp_c.h): eval-mode BatchNorm folded into
every convolution.

    python tools/convert_weights.py model.pt model.armc       # needs torch + the real (git-LFS) weights

`params` everywhere below is a dict name -> numpy array with the state_dict's names.
"""
import struct
import sys

import numpy as np

CONVS = ["init_conv1", "init_conv2", "init_conv3", "init_conv4", "init_conv5", "init_flatten"]
FC_BN = ["tar0_conv1", "out0_conv1", "out1_conv1", "out1_conv2", "out1_conv3"]
FC_OUT = ["out2_conv1", "out2_conv2", "out2_conv3"]
BN_EPS = 1e-5  # torch.nn.BatchNorm2d default

SHAPES = {  # network_light.py:19-62
    "init_conv1": (24, 1, 3, 3), "init_conv2": (24, 24, 3, 3), "init_conv3": (48, 24, 3, 3),
    "init_conv4": (48, 48, 3, 3), "init_conv5": (48, 48, 3, 3), "init_flatten": (48, 48, 15, 15),
    "tar0_conv1": (16, 10, 1, 1), "out0_conv1": (48, 64, 1, 1), "out1_conv1": (24, 48, 1, 1),
    "out1_conv2": (24, 48, 1, 1), "out1_conv3": (36, 48, 1, 1),
    "out2_conv1": (1, 24, 1, 1), "out2_conv2": (1, 24, 1, 1), "out2_conv3": (1, 36, 1, 1),
}
WITH_BIAS = ("out2_conv1", "out2_conv2", "out2_conv3")


def random_params(seed=0):
    """Seeded, well-conditioned parameters with the reference state_dict's names and shapes."""
    rng = np.random.default_rng(seed)
    p = {}
    for name, shp in SHAPES.items():
        fan_in = shp[1] * shp[2] * shp[3]
        p[name + ".weight"] = (rng.standard_normal(shp) * np.sqrt(2.0 / fan_in)).astype(np.float32)
        if name in WITH_BIAS:
            p[name + ".bias"] = (rng.standard_normal(shp[0]) * 0.1).astype(np.float32)
        else:
            c = shp[0]
            p[name + "_bn.weight"] = rng.uniform(0.5, 1.5, c).astype(np.float32)
            p[name + "_bn.bias"] = (rng.standard_normal(c) * 0.1).astype(np.float32)
            p[name + "_bn.running_mean"] = (rng.standard_normal(c) * 0.1).astype(np.float32)
            p[name + "_bn.running_var"] = rng.uniform(0.5, 1.5, c).astype(np.float32)
    return p




def fold(params, name):
    """(W', b') with y = conv(x, W') + b'  ==  BN(conv(x, W)) in eval mode."""
    w = np.asarray(params[name + ".weight"], np.float64)
    g = np.asarray(params[name + "_bn.weight"], np.float64)
    b = np.asarray(params[name + "_bn.bias"], np.float64)
    m = np.asarray(params[name + "_bn.running_mean"], np.float64)
    v = np.asarray(params[name + "_bn.running_var"], np.float64)
    s = g / np.sqrt(v + BN_EPS)
    return (w * s.reshape(-1, 1, 1, 1)).astype(np.float32), (b - m * s).astype(np.float32)


def to_blob(params) -> bytes:
    parts = []
    for n in CONVS:
        w, b = fold(params, n)
        parts += [w.ravel(), b.ravel()]
    for n in FC_BN:
        w, b = fold(params, n)
        parts += [w.reshape(w.shape[0], w.shape[1]).ravel(), b.ravel()]
    for n in FC_OUT:
        parts += [np.asarray(params[n + ".weight"], np.float32).ravel(),
                  np.asarray(params[n + ".bias"], np.float32).ravel()]
    body = np.concatenate(parts).astype("<f4").tobytes()
    return b"ARMC" + struct.pack("<B3x", 1) + body


def main():
    import torch
    sd = torch.load(sys.argv[1], map_location="cpu")
    params = {k: v.detach().float().numpy() for k, v in sd.items()}
    open(sys.argv[2], "wb").write(to_blob(params))


if __name__ == "__main__":
    main()
