"""Synthetic grid_map inputs for benchmarks and tests (SURVEY.md 8d "Synthetic inputs").

This is INPUT GENERATION, not part of the hot path: it produces the layers the reference's map
preprocessing (art_planner/src/map/processors/basic.cpp:42-125, probability_distribution.cpp:20-46,
utils.cpp:213-326 estimateNormals) would hand to the sampler / validity checker, for a seeded Perlin
terrain with box obstacles.  The morphology uses scipy.ndimage with a disk footprint instead of
OpenCV's drawn circle kernel -- the layers are statistically like the reference's, not bit-equal
(map preprocessing is "next" row N2 of SURVEY.md 8f, out of scope this round).

All layers are numpy float32 arrays of shape (rows, cols) in Fortran (column-major) order, i.e. the
memory layout of grid_map::Matrix (Eigen::MatrixXf).  Row index i grows with DEcreasing world x,
column index j with DEcreasing world y (grid_map convention).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict

import numpy as np
from scipy import ndimage


@dataclass
class GridMap:
    """Minimal stand-in for grid_map::GridMap: geometry + named float32 col-major layers."""

    rows: int
    cols: int
    res: float
    pos_x: float = 0.0
    pos_y: float = 0.0
    layers: Dict[str, np.ndarray] = field(default_factory=dict)

    @property
    def len_x(self) -> float:
        return self.rows * self.res

    @property
    def len_y(self) -> float:
        return self.cols * self.res

    def add(self, name: str, data: np.ndarray) -> None:
        assert data.shape == (self.rows, self.cols)
        self.layers[name] = np.asfortranarray(data, dtype=np.float32)

    def __getitem__(self, name: str) -> np.ndarray:
        return self.layers[name]

    def cell_x(self) -> np.ndarray:
        """World x of every row (grid_map getPosition)."""
        return (self.pos_x + (0.5 * self.len_x - 0.5 * self.res)) - self.res * np.arange(self.rows)

    def cell_y(self) -> np.ndarray:
        return (self.pos_y + (0.5 * self.len_y - 0.5 * self.res)) - self.res * np.arange(self.cols)


def _perlin(n: int, period_cells: float, rng: np.random.Generator) -> np.ndarray:
    """One octave of 2-D gradient noise on an n x n lattice, values roughly in [-1, 1]."""
    g = int(np.ceil(n / period_cells)) + 2
    ang = rng.uniform(0.0, 2.0 * np.pi, size=(g, g))
    gx, gy = np.cos(ang), np.sin(ang)
    u = np.arange(n) / period_cells
    i0 = np.floor(u).astype(int)
    f = u - i0
    fx, fy = np.meshgrid(f, f, indexing="ij")
    ix, iy = np.meshgrid(i0, i0, indexing="ij")

    def dot(di, dj):
        return gx[ix + di, iy + dj] * (fx - di) + gy[ix + di, iy + dj] * (fy - dj)

    def fade(t):
        return t * t * t * (t * (t * 6 - 15) + 10)

    sx, sy = fade(fx), fade(fy)
    n00, n10, n01, n11 = dot(0, 0), dot(1, 0), dot(0, 1), dot(1, 1)
    nx0 = n00 + sx * (n10 - n00)
    nx1 = n01 + sx * (n11 - n01)
    return (nx0 + sy * (nx1 - nx0)) * np.sqrt(2.0)


def perlin_terrain(n: int, res: float, seed: int = 1234, amplitude: float = 0.5, octaves: int = 4,
                   base_period_m: float = 4.0, persistence: float = 0.5, n_boxes: int = 12,
                   box_height: float = 0.6) -> np.ndarray:
    """4-octave Perlin terrain + axis-aligned raised boxes (obstacles)."""
    rng = np.random.default_rng(seed)
    h = np.zeros((n, n))
    amp, period, norm = 1.0, base_period_m / res, 0.0
    for _ in range(octaves):
        h += amp * _perlin(n, period, rng)
        norm += amp
        amp *= persistence
        period /= 2.0
    h *= amplitude / norm
    for _ in range(n_boxes):
        w = rng.uniform(0.8, 2.0, size=2) / res
        c = rng.uniform(0.1 * n, 0.9 * n, size=2)
        i0, i1 = int(c[0] - w[0] / 2), int(c[0] + w[0] / 2)
        j0, j1 = int(c[1] - w[1] / 2), int(c[1] + w[1] / 2)
        h[max(i0, 0):min(i1, n), max(j0, 0):min(j1, n)] += box_height
    return h.astype(np.float32)


def _disk(size: int) -> np.ndarray:
    size = max(int(size), 1)
    r = size // 2
    yy, xx = np.mgrid[-r:size - r, -r:size - r]
    return (xx * xx + yy * yy) <= r * r


def _dilate(m, size):
    # cv::dilate: max over the kernel placed with its anchor (size // 2) on the pixel, no reflection of the
    # kernel (scipy's grey_dilation would reflect it, which differs for even sizes)
    return ndimage.maximum_filter(m, footprint=_disk(size), mode="nearest") if size > 0 else m


def _erode(m, size):
    return ndimage.minimum_filter(m, footprint=_disk(size), mode="nearest") if size > 0 else m


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


@dataclass
class RobotDims:
    """Robot numbers used by the preprocessing (params.yaml:55-71 by default)."""
    torso_length: float = 1.31
    torso_width: float = 0.65
    reach_x: float = 0.2
    reach_y: float = 0.2


def make_map(n: int = 400, res: float = 0.04, seed: int = 1234, flat: bool = False,
             trav_thres: float = 0.15, robot: RobotDims = RobotDims(),
             foothold_margin: float = 0.3, hole_size_m: float = 0.3, max_drop: float = 0.3,
             drop_search_radius: float = 0.16, min_step: float = 0.3, foothold_size: float = 0.1,
             with_upper_bound: bool = False) -> GridMap:
    """Build a GridMap with every layer the hot path reads.

    Layers: elevation, traversability, elevation_masked, normal_{x,y,z}, plane_fit_std_dev,
    sample_probability, cum_prob, cum_prob_rowwise (column 0 of cum_prob_rowwise_hack) and optionally
    upper_bound.
    """
    gm = GridMap(n, n, res)
    elev = np.zeros((n, n), np.float32) if flat else perlin_terrain(n, res, seed)
    gm.add("elevation", elev)
    if with_upper_bound:
        rng = np.random.default_rng(seed + 1)
        extra = np.maximum(0.0, 0.2 * _perlin(n, 2.0 / res, rng)).astype(np.float32)
        gm.add("upper_bound", elev + extra)
    # traversability = 1 - clamp(slope / 0.6, 0, 1) from central differences
    gx, gy = np.gradient(elev.astype(np.float64), res)
    slope = np.sqrt(gx * gx + gy * gy)
    trav = (1.0 - np.clip(slope / 0.6, 0.0, 1.0)).astype(np.float32)
    gm.add("traversability", trav)

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
