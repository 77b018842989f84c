"""Shared test helpers: seeded pose / state generators and small synthetic maps."""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.dirname(os.path.abspath(__file__)) not in sys.path:
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from synthetic import GridMap, make_map  # noqa: E402

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def quat_to_R_f32(w, x, y, z):
    """Eigen::Quaternionf::toRotationMatrix in float32 (vectorised)."""
    w, x, y, z = (np.asarray(a, np.float32) for a in (w, x, y, z))
    tx, ty, tz = np.float32(2) * x, np.float32(2) * y, np.float32(2) * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    one = np.float32(1)
    R = np.empty(w.shape + (3, 3), np.float32)
    R[..., 0, 0] = one - (tyy + tzz)
    R[..., 0, 1] = txy - twz
    R[..., 0, 2] = txz + twy
    R[..., 1, 0] = txy + twz
    R[..., 1, 1] = one - (txx + tzz)
    R[..., 1, 2] = tyz - twx
    R[..., 2, 0] = txz - twy
    R[..., 2, 1] = tyz + twx
    R[..., 2, 2] = one - (txx + tyy)
    return R


def rpy_to_quat(roll, pitch, yaw):
    cr, sr = np.cos(roll / 2), np.sin(roll / 2)
    cp, sp = np.cos(pitch / 2), np.sin(pitch / 2)
    cy, sy = np.cos(yaw / 2), np.sin(yaw / 2)
    w = cy * cp * cr + sy * sp * sr
    x = cy * cp * sr - sy * sp * cr
    y = sy * cp * sr + cy * sp * cr
    z = sy * cp * cr - cy * sp * sr
    return w, x, y, z


def elevation_at(gm: GridMap, x, y, layer="elevation"):
    i = np.clip(((gm.pos_x + gm.len_x / 2 - x) / gm.res).astype(int), 0, gm.rows - 1)
    j = np.clip(((gm.pos_y + gm.len_y / 2 - y) / gm.res).astype(int), 0, gm.cols - 1)
    h = gm[layer][i, j]
    return np.where(np.isfinite(h), h, 0.0)


def make_dposes(x, y, z, roll, pitch, yaw):
    n = len(x)
    w, qx, qy, qz = rpy_to_quat(roll, pitch, yaw)
    R = quat_to_R_f32(w, qx, qy, qz)
    P = np.zeros((n, 16), np.float32)
    P[:, 0], P[:, 1], P[:, 2] = x, y, z
    P[:, 4:7] = R[:, 0]
    P[:, 8:11] = R[:, 1]
    P[:, 12:15] = R[:, 2]
    return P


def random_dposes(gm: GridMap, n, rng, z_off=(0.3, 0.15), tilt=0.45, spread=0.55):
    """Random box poses around the terrain surface (and a bit beyond the map border)."""
    x = gm.pos_x + rng.uniform(-gm.len_x * spread, gm.len_x * spread, n)
    y = gm.pos_y + rng.uniform(-gm.len_y * spread, gm.len_y * spread, n)
    z = elevation_at(gm, x, y) + z_off[0] + rng.normal(0, 1, n) * z_off[1]
    return make_dposes(x, y, z, rng.uniform(-tilt, tilt, n), rng.uniform(-tilt, tilt, n),
                       rng.uniform(-np.pi, np.pi, n))


def random_states(gm: GridMap, n, rng, z_off=(0.0, 0.05), tilt=0.2, spread=0.52):
    """Random SE3 states (x y z qx qy qz qw), feet plane near the terrain surface."""
    x = gm.pos_x + rng.uniform(-gm.len_x * spread, gm.len_x * spread, n)
    y = gm.pos_y + rng.uniform(-gm.len_y * spread, gm.len_y * spread, n)
    z = elevation_at(gm, x, y) + z_off[0] + rng.normal(0, 1, n) * z_off[1]
    w, qx, qy, qz = rpy_to_quat(rng.uniform(-tilt, tilt, n), rng.uniform(-tilt, tilt, n),
                                rng.uniform(-np.pi, np.pi, n))
    return np.stack([x, y, z, qx, qy, qz, w], axis=1).astype(np.float64)


def crop_map(gm: GridMap, i0, j0, n) -> GridMap:
    """n x n crop (keeps resolution; position recentred on the crop)."""
    out = GridMap(n, n, gm.res)
    x_c = gm.cell_x()[i0:i0 + n].mean()
    y_c = gm.cell_y()[j0:j0 + n].mean()
    out.pos_x, out.pos_y = float(x_c), float(y_c)
    for k, v in gm.layers.items():
        if v.ndim == 2:
            out.layers[k] = np.asfortranarray(v[i0:i0 + n, j0:j0 + n])
    return out


def slab_slit_map(n=120, res=0.05) -> GridMap:
    """Slabs, slits and non-finite cells in the spirit of art_planner/src/ode_test.cpp:24-84
    (a map RECIPE, not its code): flat ground, a raised slab, a thin wall, a trench and a 2x2 NaN block."""
    gm = GridMap(n, n, res)
    h = np.zeros((n, n), np.float32)
    h[20:50, 30:70] = 0.35           # slab
    h[70:72, 10:110] = 0.8           # thin wall
    h[85:100, 40:44] = -0.4          # trench (slit)
    h[60:64, 80:100] = 0.15          # low step
    gm.add("elevation", h)
    hm = h.copy()
    hm[70:72, 10:110] = -np.inf      # wall is untraversable
    hm[0:6, :] = -np.inf             # a masked border strip
    hn = hm.copy()
    hn[30:32, 90:92] = np.nan        # 2x2 NaN block (cf. ode_test.cpp:71-74)
    gm.add("elevation_masked", hm)
    gm.add("elevation_nan", hn)
    return gm


def oracle_states_valid_threaded(gm, rob, se3, n_threads=None):
    """The CPU oracle's labels of a LARGE state list: one private checker per thread (the C calls release the GIL), static
    partition -- the full 2^20-state batches of the GPU suite take a second or two on the GPU box's host cores instead of
    20 s on one (the driver's suite checks every label of its full-size batches, not a sample)."""
    import threading
    import oracle_py as O
    se3 = np.ascontiguousarray(se3, np.float64).reshape(-1, 7)
    if n_threads is None:
        try:
            n_threads = max(1, min(16, len(os.sched_getaffinity(0))))
        except AttributeError:
            n_threads = 4
    out = np.empty(len(se3), np.uint8)
    chunks = np.array_split(np.arange(len(se3)), n_threads)
    maps = [O.OracleMap(gm) for _ in range(n_threads)]

    def work(k):
        if len(chunks[k]):
            out[chunks[k]] = maps[k].states_valid(rob, se3[chunks[k]])

    th = [threading.Thread(target=work, args=(k,)) for k in range(n_threads)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    return out
