"""The C++ host mirror (art_planner_amd/host: reference class names/signatures over the C ABI)."""
import os
import struct
import subprocess

import numpy as np
import pytest

import common
import golden_io

HOST = os.path.join(common.ROOT, "art_planner_amd", "host")
BIN = os.path.join(HOST, "test_host")


def _build():
    subprocess.check_call(["make", "-s", "-C", HOST])
    assert os.path.exists(BIN)


def test_host_mirror_builds_and_refuses_without_gpu():
    _build()
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    r = subprocess.run([BIN], capture_output=True, text=True)
    assert r.returncode == 3, r.stdout + r.stderr   # context creation throws: no CPU fallback


@pytest.mark.gpu
def test_host_mirror_labels_match_golden(tmp_path):
    _build()
    gm, _ = golden_io.load_boxes("slab120")
    s = golden_io.load_states("slab120")["yaml"]
    path = tmp_path / "fixture.bin"
    with open(path, "wb") as f:
        f.write(struct.pack("<ii", gm.rows, gm.cols))
        f.write(struct.pack("<dddd", gm.len_x, gm.len_y, gm.pos_x, gm.pos_y))
        f.write(np.asfortranarray(gm["elevation"], np.float32).tobytes(order="F"))
        f.write(np.asfortranarray(gm["elevation_masked"], np.float32).tobytes(order="F"))
        f.write(struct.pack("<i", len(s["se3"])))
        f.write(np.ascontiguousarray(s["se3"], np.float64).tobytes())
        f.write(np.ascontiguousarray(s["valid"], np.uint8).tobytes())
    r = subprocess.run([BIN, str(path)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
