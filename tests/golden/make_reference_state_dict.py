#!/usr/bin/env python3
"""Build container only: torch.save()s the state_dict of the REFERENCE network class
(art_planner_motion_cost/.../predictor/network_light.py, imported where it lies) loaded with the seeded
parameters of tools/convert_weights.random_params(seed) -- the file format predictor.py:20 torch.load()s
(`self.network.load_state_dict(torch.load(modelFile))`).  Used by tests/test_motion_cost.py to exercise
tools/convert_weights.py main() on a real torch-saved checkpoint (the repository's trained checkpoints are
git-LFS stubs).  Usage: make_reference_state_dict.py out.pt [seed]"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tools"))
REF = "/root/reference/art_planner_motion_cost/src/art_planner_motion_cost/predictor"


def build_state_dict(seed=0):
    sys.path.insert(0, REF)
    import network_light  # the reference
    import convert_weights
    params = convert_weights.random_params(seed)
    net = network_light.network().eval()
    sd = net.state_dict()
    missing = [k for k in sd if not k.endswith("num_batches_tracked") and k not in params]
    assert not missing, missing
    for k in sd:
        if not k.endswith("num_batches_tracked"):
            assert tuple(sd[k].shape) == params[k].shape, (k, sd[k].shape, params[k].shape)
            sd[k] = torch.from_numpy(params[k].copy())
    net.load_state_dict(sd)
    return net.state_dict()


if __name__ == "__main__":
    torch.save(build_state_dict(int(sys.argv[2]) if len(sys.argv) > 2 else 0), sys.argv[1])
