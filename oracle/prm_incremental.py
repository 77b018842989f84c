"""CPU ORACLE (test infrastructure only): the reference's OWN roadmap construction, restated literally.

Unlike oracle/lazy_prm_cpu.py (which restates the batched front end of the product), this file follows the reference's
incremental planner step by step:

  PRMMotionCostMaintainer::sampleGraph   art_planner/src/planners/prm_motion_cost.cpp:145-219
      every accepted sample -> addValidMilestone, until max_n_vertices / max_n_edges (:171-172)
  PRMMotionCost::addValidMilestone       :325-390
      * the new milestone m is a graph vertex FIRST (:326) and enters the nearest-neighbour structure LAST (:387),
        so connectionStrategy_(m) (:334) sees its PREDECESSORS only;
      * PRMMotionCost derives from og::LazyPRMstar (prm_motion_cost.h:29): connectionStrategy_ is OMPL 1.4.2's
        KStarStrategy, k = ceil(e (1 + 1/d) ln n) with d = 6 and n = milestoneCount() = boost::num_vertices(g_) AT
        INSERTION TIME (m and every interpolated vertex included), neighbours = nn_->nearestK(m, k) by the SE3 distance;
      * per neighbour (connectionFilter_ of LazyPRM: always true): n_interp = floor(lateralDistance / 0.5) (:340-344);
        0 -> one direct edge; otherwise the interior states at t = step / (n_interp + 1) are checked IN ORDER, each
        valid one becomes a vertex chained to the previous one AND a nearest-neighbour target for every later
        milestone (:353-366), the first invalid one ends the chain (:367-371) -- the valid PREFIX stays in the graph --
        and only a complete chain gets its last edge to the neighbour (:373-377);
  PRMMotionCost::baseSolve               :432-533  (start and goal are added as milestones AFTER the roadmap, :447-470)
  PRMMotionCost::constructSolution       :536-673  (A*; the path's edges are checked with si_->checkMotion, the FIRST
                                                    invalid edge is removed and the search repeated, :640-661)
Edge weights: the reference fills them in one batch (updateEdges, :27-73) from the learned motion cost; for the
comparisons of tests/test_roadmap.py the weight is PathLengthObjective::motionCostHeuristic
(path_length_objective.cpp:58-70: Euclidean distance / max_lon_vel), objective 0 of artp_roadmap_build.

OMPL is not installed in this image: nearestK / KStarStrategy / SE3StateSpace::distance / interpolate are restated from
the published 1.4.2 sources (parity unpinned, see oracle/README.md); the validity checks are the C oracle's.
"""
import heapq
import math

import numpy as np

K_PRM = math.e * (1.0 + 1.0 / 6.0)   # KStarStrategy: kPRMConstant = e + e / dim, dim = 6 (SE3)
K_MAX_DIST = 0.5                      # prm_motion_cost.cpp:342


def se3_distance_to(verts, s):
    """OMPL SE3StateSpace::distance of every row of verts to s: |dp| + SO3 arc (subspace weights 1, 1)."""
    dp = np.linalg.norm(verts[:, :3] - s[:3], axis=1)
    dq = np.abs(verts[:, 3:] @ s[3:])
    arc = np.where(dq > 1.0 - 1e-9, 0.0, np.arccos(np.minimum(dq, 1.0)))
    return dp + arc


class IncrementalPRM:
    """The graph of PRMMotionCost: vertices = states, undirected edges with a weight, nn_ = the vertices that
    already went through nn_->add."""

    def __init__(self, om, rob, interpolate, max_lon_vel=0.5, cost_fn=None):
        """cost_fn(state_a, state_b): the objective's motionCost in the direction the edge is ADDED (the reference
        computes a weight once per edge: opt_->motionCost(m, n), lazy_prm_star_min_update.cpp:436; source -> target in
        updateEdges, prm_motion_cost.cpp:33-44).  None = PathLengthObjective without use_directional_cost (symmetric,
        Euclidean / max_lon_vel, also the A* heuristic); with a cost_fn the search runs without a heuristic."""
        self.om, self.rob, self.interpolate = om, rob, interpolate
        self.max_lon_vel = max_lon_vel
        self.cost_fn = cost_fn
        self.verts = np.empty((1024, 7), np.float64)
        self.nv = 0
        self.in_nn = np.zeros(1024, bool)
        self.is_milestone = []          # False for interpolated chain vertices
        self.edges = {}                 # (min, max) -> weight
        self.adj = {}
        self.states_checked = 0

    def _add_vertex(self, s, milestone):
        if self.nv == len(self.verts):
            self.verts = np.concatenate([self.verts, np.empty_like(self.verts)])
            self.in_nn = np.concatenate([self.in_nn, np.zeros_like(self.in_nn)])
        self.verts[self.nv] = s
        self.is_milestone.append(milestone)
        self.adj[self.nv] = []
        self.nv += 1
        return self.nv - 1

    def _add_edge(self, a, b):
        if self.cost_fn is not None:
            w = float(self.cost_fn(self.verts[a], self.verts[b]))
        else:
            w = float(np.linalg.norm(self.verts[a, :3] - self.verts[b, :3]) / self.max_lon_vel)
        self.edges[(min(a, b), max(a, b))] = w
        self.adj[a].append(b)
        self.adj[b].append(a)

    def add_valid_milestone(self, s):
        """prm_motion_cost.cpp:325-390."""
        m = self._add_vertex(s, True)
        cand = np.flatnonzero(self.in_nn[:self.nv])
        if len(cand):
            k = int(math.ceil(K_PRM * math.log(self.nv)))       # n = num_vertices, m included
            d = se3_distance_to(self.verts[cand], self.verts[m])
            order = np.argsort(d, kind="stable")[:k]             # nearestK: ascending distance
            for nb in cand[order]:
                dxy = self.verts[nb, :2] - self.verts[m, :2]
                n_interp = int(math.sqrt(dxy[0] * dxy[0] + dxy[1] * dxy[1]) / K_MAX_DIST)
                if n_interp == 0:
                    self._add_edge(m, int(nb))
                    continue
                prev, ok = m, True
                for step in range(1, n_interp + 1):
                    new_state = self.interpolate(self.verts[m], self.verts[nb], step * (1.0 / (n_interp + 1)))
                    self.states_checked += 1
                    ok = ok and bool(self.om.states_valid(self.rob, new_state[None])[0])
                    if not ok:
                        break
                    v = self._add_vertex(new_state, False)
                    self._add_edge(prev, v)
                    self.in_nn[v] = True                         # nn_->add(new_milestone), :364
                    prev = v
                if ok:
                    self._add_edge(prev, int(nb))
        self.in_nn[m] = True                                     # :387
        return m

    # ---- constructSolution (:536-673) ------------------------------------------------------------------------
    def _astar(self, start, goal):
        if self.cost_fn is None:
            h = lambda v: float(np.linalg.norm(self.verts[v, :3] - self.verts[goal, :3]) / self.max_lon_vel)
        else:
            h = lambda v: 0.0
        g = {start: 0.0}
        prev = {}
        closed = set()
        heap = [(h(start), start)]
        while heap:
            _, u = heapq.heappop(heap)
            if u in closed:
                continue
            closed.add(u)
            if u == goal:
                break
            for v in self.adj[u]:
                w = self.edges.get((min(u, v), max(u, v)))
                if w is None:
                    continue
                ng = g[u] + w
                if ng < g.get(v, math.inf):
                    g[v] = ng
                    prev[v] = u
                    heapq.heappush(heap, (ng + h(v), v))
        if goal not in closed:
            return None, math.inf
        p = [goal]
        while p[-1] != start:
            p.append(prev[p[-1]])
        return p[::-1], g[goal]

    def solve(self, start, goal, max_replans=1000):
        """Returns (path vertex ids | None, cost, lazy removals, edges checked)."""
        removed = checked = 0
        for _ in range(max_replans + 1):
            p, c = self._astar(start, goal)
            if p is None:
                return None, math.inf, removed, checked
            s1, s2 = self.verts[p[:-1]], self.verts[p[1:]]
            # the reference walks the path from the goal and removes the FIRST invalid edge it meets (:640-661); every
            # edge is checked in travel direction: checkMotion(*state, *prevState), state = the vertex nearer the start
            ok, _ = self.om.check_motions(self.rob, s1[::-1], s2[::-1])
            checked += len(ok)
            bad = np.flatnonzero(ok == 0)
            if len(bad) == 0:
                return p, c, removed, checked
            i = len(p) - 2 - int(bad[0])
            a, b = p[i], p[i + 1]
            del self.edges[(min(a, b), max(a, b))]
            removed += 1
        return None, math.inf, removed, checked


def build_and_solve(om, rob, interpolate, accepted, start, goal, max_n_vertices=10000, max_n_edges=50000,
                    max_lon_vel=0.5, cost_fn=None, max_replans=1000):
    """sampleGraph over the accepted-state stream, then baseSolve: start and goal join as milestones, A* + lazy edge
    check.  Returns a dict of counts, the path and its cost."""
    g = IncrementalPRM(om, rob, interpolate, max_lon_vel, cost_fn)
    used = 0
    for s in accepted:
        if not (g.nv < max_n_vertices and len(g.edges) < max_n_edges):   # :171-172
            break
        g.add_valid_milestone(s)
        used += 1
    vs = g.add_valid_milestone(np.asarray(start, np.float64))
    vg = g.add_valid_milestone(np.asarray(goal, np.float64))
    p, c, removed, checked = g.solve(vs, vg, max_replans)
    return {"milestones_used": used, "vertices": g.nv, "chain_vertices": int(g.nv - sum(g.is_milestone)),
            "edges": len(g.edges) + removed, "interior_states_checked": g.states_checked, "lazy_removals": removed,
            "lazy_edges_checked": checked, "path_cost": c, "path": None if p is None else g.verts[p].copy(), "graph": g}


def lazy_prm_star_min_update(om, rob, accepted, start, goal, n_milestones, max_lon_vel=0.5, cost_fn=None,
                             max_replans=1000):
    """BASELINE config 1's planner, literally: LazyPRMStarMinUpdate (lazy_prm_star_min_update.cpp).
      baseSolve (:496-615): start and goal become milestones FIRST (:507-535), then one accepted sample at a time
      (`do sampleUniform while !isValid`, :552-555) through addValidMilestone (:424-446): the k = ceil(e (1 + 1/6) ln n)
      nearest PREDECESSORS (n = num_vertices at insertion, m included; m enters nn_ last), one direct edge each,
      weight = opt_->motionCost (PathLengthObjective, non-directional default = Euclidean / max_lon_vel), validity
      unknown; constructSolution (:619-747) = A* + lazy checkMotion of the path's edges, first invalid edge removed.
    The termination condition is a planning time in the reference; here it is a milestone budget, and the solution is
    constructed once on the final graph (the graph only grows, so no intermediate solution can be cheaper than the
    final search's, lazy removals aside).  Returns a dict like build_and_solve."""
    g = IncrementalPRM(om, rob, None, max_lon_vel, cost_fn)

    def add(s):
        m = g._add_vertex(np.asarray(s, np.float64), True)
        cand = np.flatnonzero(g.in_nn[:g.nv])
        if len(cand):
            k = int(math.ceil(K_PRM * math.log(g.nv)))
            d = se3_distance_to(g.verts[cand], g.verts[m])
            for nb in cand[np.argsort(d, kind="stable")[:k]]:
                g._add_edge(m, int(nb))
        g.in_nn[m] = True
        return m

    vs, vg = add(start), add(goal)
    used = 0
    for s in accepted[:n_milestones]:
        add(s)
        used += 1
    p, c, removed, checked = g.solve(vs, vg, max_replans)
    return {"milestones_used": used, "vertices": g.nv, "edges": len(g.edges) + removed, "lazy_removals": removed,
            "lazy_edges_checked": checked, "path_cost": c, "path": None if p is None else g.verts[p].copy(), "graph": g}
