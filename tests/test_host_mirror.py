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
BIN_PLANNER = os.path.join(HOST, "test_planner")


def _build():
    subprocess.check_call(["make", "-s", "-C", HOST])
    assert os.path.exists(BIN) and os.path.exists(BIN_PLANNER)


def test_host_mirror_builds_and_refuses_without_gpu():
    _build()
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    r = subprocess.run([BIN], capture_output=True, text=True)
    assert r.returncode == 3, r.stdout + r.stderr   # context creation throws: no CPU fallback
    r = subprocess.run([BIN_PLANNER], capture_output=True, text=True)
    assert r.returncode == 3, r.stdout + r.stderr


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


@pytest.mark.gpu
def test_planner_mirror_plans_on_a_perlin_map(tmp_path):
    """art_planner::Planner of the host mirror (setMap / plan / getSolutionPath, planner.h:31-71): NO_MAP, the
    ignored map without elevation layer, SOLVED with valid states and motions (original and simplified path),
    re-query on the kept roadmap after another setMap, INVALID_START / INVALID_GOAL."""
    _build()
    from art_planner_amd.context import Context
    from art_planner_amd.synthetic import make_map
    gm = make_map(250, 0.04, seed=77)
    # start / goal: two valid samples of the same map under the same preprocessing (Params defaults), far apart
    ctx = Context(0, "yaml")
    pm = ctx.preprocess_map(gm["elevation"], gm.len_x, gm.len_y, traversability=gm["traversability"], kind="defaults")
    pm.install()
    se3 = ctx.sample_states(5, 0, 4000)
    acc = se3[ctx.validate_states(se3) != 0]
    assert len(acc) > 100
    d = np.hypot(acc[:, None, 0] - acc[None, :200, 0], acc[:, None, 1] - acc[None, :200, 1])
    i, j = np.unravel_index(np.argmax(d), d.shape)
    assert d[i, j] > 5.0
    pm.close()
    ctx.close()
    path = tmp_path / "planner.bin"
    with open(path, "wb") as f:
        f.write(struct.pack("<ii", gm.rows, gm.cols))
        f.write(struct.pack("<dddd", gm.len_x, gm.len_y, gm.pos_x, gm.pos_y))
        f.write(np.asfortranarray(gm["elevation"], np.float32).tobytes(order="F"))
        f.write(np.asfortranarray(gm["traversability"], np.float32).tobytes(order="F"))
        f.write(np.ascontiguousarray(np.stack([acc[i], acc[j]]), np.float64).tobytes())
    r = subprocess.run([BIN_PLANNER, str(path)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
