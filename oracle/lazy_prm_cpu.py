"""CPU ORACLE (test infrastructure / bench.py's cpu_baseline leg only): the reference's LazyPRM* front end for
BASELINE config C1, restated over the C oracle's sampler, validity checker and discrete motion validator.

What it follows (OMPL itself is not installed here; the loop structure is the reference's own code):
  LazyPRMStarMinUpdate::baseSolve      art_planner/src/planners/lazy_prm_star_min_update.cpp:496-615
      `do sampler_->sampleUniform(s) while (!si_->isValid(s)); addValidMilestone(s)` (:552-555)
  addValidMilestone                     :424-446  (k nearest by the SE3 distance, edge weight = opt_->motionCost)
  constructSolution                     :619-747  (A*; vertices already valid; edges checked lazily with
                                                   si_->checkMotion (:725), invalid ones removed, search repeated)
  PathLengthObjective::motionCostHeuristic  art_planner/src/objectives/path_length_objective.cpp:58-70
  KStarStrategy (OMPL 1.4.2): k = ceil(e (1 + 1/d) ln n), d = 6 for SE3
The roadmap here is built in one go (all milestones, then all k-NN connections) like the batched GPU front end,
so the two are comparable; the reference inserts milestones one at a time.
"""
import hashlib
import math
import time

import numpy as np
from scipy.sparse import csr_matrix
from scipy.sparse.csgraph import dijkstra


def se3_distance(a, b):
    """OMPL SE3StateSpace::distance: |dp| + SO3 arc (both subspace weights 1)."""
    dp = np.linalg.norm(a[..., :3] - b[..., :3], axis=-1)
    dq = np.abs(np.sum(a[..., 3:] * b[..., 3:], axis=-1))
    arc = np.where(dq > 1.0 - 1e-9, 0.0, np.arccos(np.clip(dq, -1.0, 1.0)))
    return dp + arc


def lazy_prm_star(om, smp, rob, start, goal, n_milestones, seed=42, max_lon_vel=0.5):
    """Returns a dict: label hash / counts / rates of the rejection loop, lazy edge checks, path and its cost."""
    out = {}
    # ---- rejection sampling (timed: states/s of the CPU validity path) ----
    t0 = time.perf_counter()
    acc, labels, drawn = [], [], 0
    while sum(len(a) for a in acc) < n_milestones:
        se3, _ = smp.sample(rob, seed, drawn, 4096)
        v = om.states_valid(rob, se3)
        labels.append(v)
        acc.append(se3[v != 0])
        drawn += 4096
        if drawn > 400 * n_milestones + 65536:
            break
    dt = time.perf_counter() - t0
    labels = np.concatenate(labels)
    out["states_drawn"] = int(drawn)
    out["states_per_s"] = drawn / dt
    out["label_hash"] = hashlib.sha1(labels.tobytes()).hexdigest()[:16]
    out["valid_fraction"] = float(labels.mean())
    verts = np.concatenate([np.stack([start, goal])] + acc)[:n_milestones + 2]
    nv = len(verts)
    # ---- k nearest (KStarStrategy), symmetrised ----
    k = min(nv - 1, max(1, int(math.ceil(math.e * (1.0 + 1.0 / 6.0) * math.log(nv)))))
    rows, cols = [], []
    for i0 in range(0, nv, 512):
        d = se3_distance(verts[i0:i0 + 512, None, :], verts[None, :, :])
        d[np.arange(len(d)), np.arange(i0, i0 + len(d))] = np.inf
        nn = np.argpartition(d, k, axis=1)[:, :k]
        rows.append(np.repeat(np.arange(i0, i0 + len(d)), k))
        cols.append(nn.ravel())
    u, v = np.concatenate(rows), np.concatenate(cols)
    lo, hi = np.minimum(u, v), np.maximum(u, v)
    key = np.unique(lo.astype(np.int64) * nv + hi)
    eu, ev = (key // nv).astype(np.int64), (key % nv).astype(np.int64)
    w = np.linalg.norm(verts[eu, :3] - verts[ev, :3], axis=1) / max_lon_vel  # motionCostHeuristic
    alive = np.ones(len(eu), bool)
    out["vertices"], out["k"], out["candidate_edges"] = int(nv), int(k), int(len(eu))
    # ---- constructSolution: search, lazy checkMotion of the path's edges, remove, repeat ----
    edge_id = {(int(a), int(b)): i for i, (a, b) in enumerate(zip(eu, ev))}
    checked, t_check, path = 0, 0.0, None
    for _ in range(1000):
        g = csr_matrix((w[alive], (eu[alive], ev[alive])), shape=(nv, nv))
        dist, pred = dijkstra(g, directed=False, indices=0, return_predecessors=True)
        if not np.isfinite(dist[1]):
            break
        p = [1]
        while p[-1] != 0:
            p.append(int(pred[p[-1]]))
        p = p[::-1]
        s1, s2 = verts[p[:-1]], verts[p[1:]]
        t1 = time.perf_counter()
        ok, _ = om.check_motions(rob, s1, s2)
        t_check += time.perf_counter() - t1
        checked += len(ok)
        bad = np.flatnonzero(ok == 0)
        if len(bad) == 0:
            path = verts[p]
            out["path_cost"] = float(dist[1])
            break
        for b in bad:
            a_, b_ = p[b], p[b + 1]
            alive[edge_id[(min(a_, b_), max(a_, b_))]] = False
    out["lazy_edges_checked"] = int(checked)
    out["lazy_edges_per_s"] = checked / t_check if t_check > 0 else None
    out["path"] = path
    return out


def shortcut(om, rob, path, max_lon_vel=0.5):
    """Deterministic stand-in for ss_->simplifySolution() (planner.cpp:266-280; OMPL's PathSimplifier is
    randomised): the cheapest chain of shortcuts (i -> j valid under the discrete motion validator) through the
    path's states -- what artp_roadmap_simplify_path computes on the GPU."""
    n = len(path)
    best = np.full(n, np.inf)
    prev = np.full(n, -1)
    best[0] = 0.0
    for i in range(n - 1):
        js = np.arange(i + 1, n)
        ok, _ = om.check_motions(rob, np.repeat(path[i][None], len(js), 0), path[js])
        c = np.linalg.norm(path[js, :3] - path[i, :3], axis=1) / max_lon_vel
        for j, o, cc in zip(js, ok, c):
            if o and best[i] + cc < best[j]:
                best[j], prev[j] = best[i] + cc, i
    p = [n - 1]
    while p[-1] != 0 and prev[p[-1]] >= 0:
        p.append(int(prev[p[-1]]))
    return path[p[::-1]], float(best[n - 1])
