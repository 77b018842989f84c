"""ctypes wrapper of the batched roadmap front end (include/artp_c.h: artp_roadmap_*), "next" row N1.

Mirrors how the reference's planners are driven (Planner::plan -> PRMMotionCost::solve,
art_planner/src/planners/prm_motion_cost.cpp:289-297): build = sampleGraph + the connection loop of
addValidMilestone as batches, solve = baseSolve / constructSolution."""
import ctypes as C
from typing import Optional, Tuple

import numpy as np

from . import _capi


class Roadmap:
    def __init__(self, ctx, start, goal, n_milestones=10000, seed=42, first_index=0, k_neighbors=0,
                 objective=0, max_lon_vel=0.5, max_lat_vel=0.1, max_ang_vel=0.5, max_replans=1000,
                 cost_weights=None, risk_threshold=None, max_n_edges=0, recompute_density_after_n_samples=0,
                 max_sample_time=0.0, density_map=None, construction=0, max_query_edge_length=0.5):
        """construction: 0 = batched, 1 = PRMMotionCost::addValidMilestone order (the reference's own graph, chain
        vertices included), 2 = LazyPRMStarMinUpdate order (predecessor-only direct edges); include/artp_c.h.
        density_map: the PreprocessedMap of the installed map (Context.preprocess_map) -- needed for the in-build
        re-weighting of the sampling distribution (recompute_density_after_n_samples > 0)."""
        self.ctx = ctx
        self.L = ctx.L   # the library that made the context
        p = _capi.RoadmapParams()
        self.L.artp_roadmap_params_defaults(C.byref(p))
        p.seed, p.first_index, p.n_milestones, p.k_neighbors = seed, first_index, n_milestones, k_neighbors
        p.objective, p.max_replans = objective, max_replans
        p.max_lon_vel, p.max_lat_vel, p.max_ang_vel = max_lon_vel, max_lat_vel, max_ang_vel
        if cost_weights is not None:
            p.w_energy, p.w_time, p.w_risk = cost_weights
        if risk_threshold is not None:
            p.risk_threshold = risk_threshold
        p.max_n_edges, p.recompute_density_after_n_samples = max_n_edges, recompute_density_after_n_samples
        p.max_sample_time = max_sample_time
        p.construction = construction
        p.max_query_edge_length = max_query_edge_length
        self._density = density_map  # keeps the map (and its params) alive
        if density_map is not None:
            p.density_map = density_map.h
            p.density_params = C.cast(C.pointer(density_map.params), C.c_void_p)
        s = np.ascontiguousarray(start, np.float64).reshape(7)
        g = np.ascontiguousarray(goal, np.float64).reshape(7)
        h = C.c_void_p()
        ctx._chk(self.L.artp_roadmap_build(ctx.h, C.byref(p), s.ctypes.data, g.ctypes.data, C.byref(h)),
                 "artp_roadmap_build")
        self.h = h

    def close(self):
        if self.h:
            self.L.artp_roadmap_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def stats(self) -> dict:
        out = (C.c_uint64 * 8)()
        self.ctx._chk(self.L.artp_roadmap_stats(self.h, C.byref(out)), "artp_roadmap_stats")
        return {"vertices": out[0], "candidate_edges": out[1], "valid_edges": out[2], "removed_edges": out[3],
                "k": out[4], "samples_drawn": out[5], "reweightings": out[6],
                "time_budget_hit": bool(out[7] & 1), "edge_budget_hit": bool(out[7] & 2),
                "edge_budget_exceeded": bool(out[7] & 4)}

    def export(self) -> dict:
        st = self.stats()
        nv, ne, k = st["vertices"], st["candidate_edges"], st["k"]
        d = {"verts": np.empty((nv, 7), np.float64), "knn": np.empty((nv, k), np.uint32),
             "knn_dist": np.empty((nv, k), np.float64), "edges": np.empty((ne, 2), np.uint32),
             "edge_valid": np.empty(ne, np.uint8), "edge_interp": np.empty(ne, np.uint32),
             "edge_cost": np.empty(ne, np.float64), "edge_removed": np.empty(ne, np.uint8)}
        self.ctx._chk(self.L.artp_roadmap_export(self.h, *(d[k_].ctypes.data for k_ in
                                                          ("verts", "knn", "knn_dist", "edges", "edge_valid",
                                                           "edge_interp", "edge_cost", "edge_removed"))),
                      "artp_roadmap_export")
        return d

    def revalidate(self) -> dict:
        """Re-check every vertex and edge against the context's current map (after a map update)."""
        out = (C.c_uint64 * 4)()
        self.ctx._chk(self.L.artp_roadmap_revalidate(self.h, C.byref(out)), "artp_roadmap_revalidate")
        return {"invalid_vertices": out[0], "valid_edges_before": out[1], "valid_edges_after": out[2],
                "start_valid": bool(out[3] & 1), "goal_valid": bool(out[3] & 2)}

    def grow(self, n_more: int) -> dict:
        """Keep the milestones still valid on the current map, add n_more new ones, reconnect everything."""
        out = (C.c_uint64 * 2)()
        self.ctx._chk(self.L.artp_roadmap_grow(self.h, int(n_more), C.byref(out)), "artp_roadmap_grow")
        return {"kept": out[0], "dropped": out[1]}

    def set_query(self, start, goal):
        s = np.ascontiguousarray(start, np.float64).reshape(7)
        g = np.ascontiguousarray(goal, np.float64).reshape(7)
        self.ctx._chk(self.L.artp_roadmap_set_query(self.h, s.ctypes.data, g.ctypes.data), "artp_roadmap_set_query")

    def simplify(self, path):
        """Cheapest chain of valid shortcuts through the states of `path` (deterministic PathSimplifier)."""
        p = np.ascontiguousarray(path, np.float64).reshape(-1, 7)
        out = np.empty_like(p)
        n, cost = C.c_size_t(0), C.c_double(0.0)
        self.ctx._chk(self.L.artp_roadmap_simplify_path(self.h, p.ctypes.data, p.shape[0], out.ctypes.data,
                                                        C.byref(n), C.byref(cost)), "artp_roadmap_simplify_path")
        return out[:n.value].copy(), cost.value

    def solve_until(self, plan_time, grow_step, cap_states=4096):
        """LazyPRM*'s grow-while-planning loop: (path or None, cost, {"rounds", "vertices", "improved"})."""
        path = np.empty((cap_states, 7), np.float64)
        n, cost = C.c_size_t(0), C.c_double(0.0)
        st = (C.c_uint64 * 3)()
        self.ctx._chk(self.L.artp_roadmap_solve_until(self.h, plan_time, grow_step, path.ctypes.data, cap_states,
                                                      C.byref(n), C.byref(cost), C.byref(st)), "artp_roadmap_solve_until")
        info = {"rounds": st[0], "vertices": st[1], "improved": st[2]}
        if n.value == 0:
            return None, float("inf"), info
        return path[:n.value].copy(), cost.value, info

    def set_density_map(self, density_map):
        self._density = density_map
        if density_map is None:
            self.ctx._chk(self.L.artp_roadmap_set_density_map(self.h, None, None), "artp_roadmap_set_density_map")
        else:
            self.ctx._chk(self.L.artp_roadmap_set_density_map(self.h, density_map.h, C.byref(density_map.params)),
                          "artp_roadmap_set_density_map")

    def solve(self, cap_states=4096) -> Tuple[Optional[np.ndarray], float, int]:
        """(path n x 7 or None when start and goal are not connected, cost, lazy edge removals)."""
        path = np.empty((cap_states, 7), np.float64)
        n, cost, rep = C.c_size_t(0), C.c_double(0.0), C.c_int(0)
        self.ctx._chk(self.L.artp_roadmap_solve(self.h, path.ctypes.data, cap_states, C.byref(n), C.byref(cost),
                                                C.byref(rep)), "artp_roadmap_solve")
        if n.value == 0:
            return None, float("inf"), rep.value
        return path[:n.value].copy(), cost.value, rep.value
