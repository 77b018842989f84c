"""Synthetic grid_map inputs for tests, fixtures and benchmarks (SURVEY.md 8d "Synthetic inputs").

INPUT GENERATION, not part of the product: a seeded Perlin terrain with box obstacles and a slope-derived
traversability layer (raw_map); make_map adds the layers the reference's map preprocessing would hand to the
sampler / validity checker through the CPU oracle's restatement of that preprocessing (oracle/map_processors.py)
-- which is why this module lives under tests/ and not in the product package.  bench.py / smoke() take only
raw_map from here and derive the other layers with the PRODUCT's device preprocessing (map_from_device).

All layers are numpy float32 arrays of shape (rows, cols) in Fortran (column-major) order, i.e. the
memory layout of grid_map::Matrix (Eigen::MatrixXf).  Row index i grows with DEcreasing world x,
column index j with DEcreasing world y (grid_map convention).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict

import numpy as np


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


@dataclass
class RobotDims:
    """Robot numbers used by the preprocessing (params.yaml:55-71 by default)."""
    torso_length: float = 1.31
    torso_width: float = 0.65
    reach_x: float = 0.2
    reach_y: float = 0.2


def raw_map(n: int = 400, res: float = 0.04, seed: int = 1234, flat: bool = False,
             trav_thres: float = 0.15, robot: RobotDims = RobotDims(),
             foothold_margin: float = 0.3, hole_size_m: float = 0.3, max_drop: float = 0.3,
             drop_search_radius: float = 0.16, min_step: float = 0.3, foothold_size: float = 0.1,
             with_upper_bound: bool = False) -> GridMap:
    """The raw sensor-side layers: elevation, traversability (and optionally upper_bound)."""
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
    return gm


def make_map(n: int = 400, res: float = 0.04, seed: int = 1234, flat: bool = False,
             trav_thres: float = 0.15, robot: RobotDims = RobotDims(),
             foothold_margin: float = 0.3, hole_size_m: float = 0.3, max_drop: float = 0.3,
             drop_search_radius: float = 0.16, min_step: float = 0.3, foothold_size: float = 0.1,
             with_upper_bound: bool = False, elevation_layer: str = "elevation") -> GridMap:
    """Build a GridMap with every layer the hot path reads (CPU: derived layers from the oracle's restatement of
    the reference preprocessing).  elevation_layer = params.planner.elevation_layer (params.h:18): the layer the body
    checker, the sampler and the whole preprocessing read -- "upper_bound" for BASELINE config 5.

    Layers: elevation, traversability, elevation_masked, normal_{x,y,z}, plane_fit_std_dev,
    sample_probability, cum_prob, cum_prob_rowwise (column 0 of cum_prob_rowwise_hack) and optionally
    upper_bound.
    """
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import map_processors
    gm = raw_map(n, res, seed, flat, trav_thres, robot, foothold_margin, hole_size_m, max_drop, drop_search_radius,
                 min_step, foothold_size, with_upper_bound)
    return map_processors.add_derived_layers(gm, robot, trav_thres, foothold_margin, hole_size_m, max_drop,
                                             drop_search_radius, min_step, foothold_size, elevation_layer)


def cumulative_distribution(prob):
    """oracle/map_processors.cumulative_distribution (kept importable from here for the tests)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import map_processors
    return map_processors.cumulative_distribution(prob)


def map_from_device(ctx, raw: GridMap, kind: str = "yaml", body_layer: str = "elevation", **overrides) -> GridMap:
    """The same map with its derived layers computed by the PRODUCT (artp_preprocess_map on the context's GPU) and
    installed as the context's map; the layers come back for the CPU oracle legs of bench.py / smoke()."""
    pm = ctx.preprocess_map(raw[body_layer], raw.len_x, raw.len_y, raw.pos_x, raw.pos_y,
                            traversability=raw["traversability"], kind=kind, **overrides)
    pm.install()
    gm = GridMap(raw.rows, raw.cols, raw.res, raw.pos_x, raw.pos_y)
    for name, arr in raw.layers.items():
        gm.layers[name] = arr
    for name in ("elevation_masked", "normal_x", "normal_y", "normal_z", "plane_fit_std_dev", "sample_probability",
                 "cum_prob", "traversability_thresholded"):
        gm.add(name, pm.layer(name))
    gm.layers["cum_prob_rowwise"] = np.ascontiguousarray(pm.layer("cum_prob_rowwise"), dtype=np.float32)
    gm.preprocessed = pm
    return gm
