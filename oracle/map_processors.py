"""CPU ORACLE (test infrastructure only): numpy restatement of the reference's per-map preprocessing --
processors::Basic (art_planner/src/map/processors/basic.cpp:42-125), estimateNormals
(art_planner/src/utils.cpp:213-326) and computeCumulativeProbabilityDistribution
(processors/probability_distribution.cpp:20-46) -- the checker of the device chain artp_preprocess_map
 and the source of the derived layers of the CPU-side test maps (tests/synthetic.py).

OpenCV is not installed here: the morphology uses scipy.ndimage with a disk footprint x^2 + y^2 <= (size/2)^2 where
the reference rasterises cv::circle -- parity with OpenCV is UNPINNED (single boundary pixels of the footprint may
differ); everything else follows the reference's arithmetic in float32.
"""
from __future__ import annotations

import numpy as np


def _disk(size: int) -> np.ndarray:
    """getCircularKernel(size) (utils.cpp:114-119): cv::circle(centre (size/2, size/2), radius size/2, filled) on a
    size x size image -- OpenCV's midpoint circle fill (drawing.cpp Circle()), restated from the published source
    (OpenCV is not installed: unpinned).  size <= 0: an empty cv::Mat, for which cv::erode / cv::dilate use a
    3 x 3 rectangle."""
    if size <= 0:
        return np.ones((3, 3), bool)
    size = int(size)
    k = np.zeros((size, size), bool)
    radius = size // 2
    cx = cy = radius

    def hline(y, x0, x1):
        if 0 <= y < size:
            k[y, max(x0, 0):min(x1, size - 1) + 1] = True

    err, dx, dy, plus, minus = 0, radius, 0, 1, (radius << 1) - 1
    while dx >= dy:
        hline(cy - dy, cx - dx, cx + dx)
        hline(cy + dy, cx - dx, cx + dx)
        hline(cy - dx, cx - dy, cx + dy)
        hline(cy + dx, cx - dy, cx + dy)
        dy += 1
        err += plus
        plus += 2
        mask = -1 if err > 0 else 0          # (err <= 0) - 1
        err -= minus & mask
        dx += mask
        minus -= mask & 2
    return k


def _morph(m, size, dilate):
    """cv::dilate / cv::erode with the kernel anchored at (size/2, size/2): the extreme over the footprint offsets
    (x - r along the rows index, y - r along the columns index), out-of-image cells replicated (which for these
    footprints equals OpenCV's ignored border)."""
    k = _disk(size)
    n = k.shape[0]
    r = n // 2
    pad = np.pad(m, ((r, n - r), (r, n - r)), mode="edge")
    R, C = m.shape
    out = None
    for y in range(n):
        for x in range(n):
            if not k[y, x]:
                continue
            v = pad[x:x + R, y:y + C]
            out = v.copy() if out is None else (np.maximum(out, v) if dilate else np.minimum(out, v))
    return out


def _dilate(m, size):
    return _morph(m, size, True)


def _erode(m, size):
    return _morph(m, size, False)


def estimate_normals(gm: GridMap, elevation: np.ndarray, radius_m: float):
    """Vectorised form of estimateNormals (art_planner/src/utils.cpp:213-326)."""
    n_r = int(radius_m / gm.res)
    n_d = int(radius_m * 0.70710678118 / gm.res)
    rows, cols = elevation.shape
    X = np.broadcast_to(gm.cell_x()[:, None], (rows, cols)).astype(np.float32)
    Y = np.broadcast_to(gm.cell_y()[None, :], (rows, cols)).astype(np.float32)
    Z = elevation.astype(np.float32)
    P = np.stack([X, Y, Z], axis=-1)
    vec_sum = np.zeros((rows, cols, 3), np.float32)
    n_vec = np.zeros((rows, cols), np.int32)
    max_dz = np.zeros((rows, cols), np.float32)

    def shifted(di, dj):
        """P[i+di, j+dj] with validity mask."""
        out = np.zeros_like(P)
        ok = np.zeros((rows, cols), bool)
        i_src = slice(max(di, 0), rows + min(di, 0))
        i_dst = slice(max(-di, 0), rows + min(-di, 0))
        j_src = slice(max(dj, 0), cols + min(dj, 0))
        j_dst = slice(max(-dj, 0), cols + min(-dj, 0))
        out[i_dst, j_dst] = P[i_src, j_src]
        ok[i_dst, j_dst] = True
        return out, ok

    def accumulate(a, b):
        nonlocal vec_sum, n_vec, max_dz
        (pa, oka), (pb, okb) = a, b
        ok = oka & okb
        vx, vy = pa - P, pb - P
        c = np.cross(vx, vy)
        nrm = np.linalg.norm(c, axis=-1, keepdims=True)
        c = c / np.where(nrm > 0, nrm, 1)
        vec_sum += np.where(ok[..., None], c, 0).astype(np.float32)
        n_vec += ok
        dz = np.maximum(np.abs(vx[..., 2]), np.abs(vy[..., 2]))
        max_dz = np.where(ok, np.maximum(max_dz, dz), max_dz)

    for o in range(1, n_r):
        # both i+o and j+o must be in range (the reference skips the pair otherwise)
        a, b = shifted(o, 0), shifted(0, o)
        both = shifted(o, o)[1]
        accumulate((a[0], a[1] & both), (b[0], b[1] & both))
    for o in range(1, n_r):
        a, b = shifted(-o, 0), shifted(0, -o)
        both = shifted(-o, -o)[1]
        accumulate((a[0], a[1] & both), (b[0], b[1] & both))
    for o in range(1, n_d):
        accumulate(shifted(o, o), shifted(-o, o))
    for o in range(1, n_d):
        accumulate(shifted(-o, -o), shifted(o, -o))
    vec_sum = vec_sum / np.maximum(n_vec, 1)[..., None]
    nrm = np.linalg.norm(vec_sum, axis=-1, keepdims=True)
    nvec = vec_sum / np.where(nrm > 0, nrm, 1)
    return (nvec[..., 0].astype(np.float32), nvec[..., 1].astype(np.float32),
            nvec[..., 2].astype(np.float32), max_dz.astype(np.float32))


def cumulative_distribution(prob: np.ndarray):
    """computeCumulativeProbabilityDistribution (probability_distribution.cpp:20-46), float32."""
    prob = prob.astype(np.float32)
    prob_rowwise = prob.sum(axis=1, dtype=np.float32)
    prob_rowwise = prob_rowwise / prob_rowwise.sum(dtype=np.float32)
    with np.errstate(invalid="ignore", divide="ignore"):
        cum_prob = prob / prob.sum(axis=1, dtype=np.float32)[:, None]
    cum_prob_rowwise = np.cumsum(prob_rowwise, dtype=np.float32)
    cum_prob = np.cumsum(cum_prob, axis=1, dtype=np.float32)
    return cum_prob.astype(np.float32), cum_prob_rowwise.astype(np.float32)


def add_derived_layers(gm, robot, trav_thres: float = 0.15, foothold_margin: float = 0.3, hole_size_m: float = 0.3,
                       max_drop: float = 0.3, drop_search_radius: float = 0.16, min_step: float = 0.3,
                       foothold_size: float = 0.1, elevation_layer: str = "elevation"):
    """processors::Basic + the CDF on a map that carries `elevation` and `traversability`: adds normal_{x,y,z},
    plane_fit_std_dev, traversability_thresholded, elevation_masked, sample_probability, cum_prob and
    cum_prob_rowwise (column 0 of cum_prob_rowwise_hack).  robot: an object with torso_length, torso_width,
    reach_x, reach_y.  elevation_layer = params.planner.elevation_layer: the layer every derived layer comes from
    (basic.cpp:45-47,75,104; "upper_bound" in BASELINE config 5)."""
    res = gm.res
    elev, trav = gm[elevation_layer], gm["traversability"]
    nx, ny, nz, std = estimate_normals(gm, elev, (robot.torso_length + robot.torso_width) * 0.25)
    gm.add("normal_x", nx)
    gm.add("normal_y", ny)
    gm.add("normal_z", nz)
    gm.add("plane_fit_std_dev", std)

    # basic.cpp:57-106 safety morphology (disk footprints)
    trav_filter = (trav > trav_thres).astype(np.float32)
    fh = int(np.ceil(foothold_size / res))
    margin = int(np.ceil(2 * foothold_margin / res))
    hole = int(np.floor(hole_size_m / res))
    safety = _erode(_dilate(trav_filter, hole), hole)
    search = int(np.ceil(2 * drop_search_radius / res))
    hole_mask = (elev - _erode(elev, search)) > max_drop
    safety = np.where(hole_mask, trav_filter, safety)
    wall_mask = (_dilate(elev, margin) - elev) > min_step
    safety = np.where(wall_mask, 1.0, safety).astype(np.float32)
    safety = _erode(safety, margin)
    safety = np.where((trav_filter < 0.5) | wall_mask, trav_filter, safety)
    safety = _dilate(_erode(safety, fh), fh)
    safety = np.where(trav_filter < 0.5, trav_filter, safety).astype(np.float32)
    gm.add("traversability_thresholded", safety)
    masked = np.where(safety > 0.5, elev, -np.inf).astype(np.float32)
    gm.add("elevation_masked", masked)

    # basic.cpp:110-125 sample filter, probability_distribution.cpp
    total_reach = np.sqrt(robot.reach_x ** 2 + robot.reach_y ** 2)
    sf = _erode(_dilate(safety, int(total_reach / res)), int(total_reach / res))
    min_wall = min((robot.torso_length - robot.reach_x) * 0.5, (robot.torso_width - robot.reach_y) * 0.5)
    sf = _erode(sf, int(min_wall / res))
    if sf.sum() == 0:  # degenerate synthetic map: fall back to uniform
        sf = np.ones_like(sf)
    gm.add("sample_probability", sf)
    cum_prob, cum_rowwise = cumulative_distribution(sf)
    gm.add("cum_prob", cum_prob)
    gm.layers["cum_prob_rowwise"] = np.ascontiguousarray(cum_rowwise, dtype=np.float32)
    return gm
