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
