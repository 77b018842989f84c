"""CPU suite: invariants of oracle/prm_incremental.py, the literal restatement of the reference's incremental roadmap
construction (PRMMotionCost::addValidMilestone, prm_motion_cost.cpp:325-390; LazyPRMStarMinUpdate, C1's planner)."""
import math
import os
import sys

import numpy as np

import common
import oracle_py as O

sys.path.insert(0, os.path.join(common.ROOT, "oracle"))
import prm_incremental as PI  # noqa: E402


def _accepted(gm, rob, n_draw, seed=42):
    om, smp = O.OracleMap(gm), O.OracleSampler(gm)
    se3, _ = smp.sample(rob, seed, 0, n_draw)
    return om, se3[om.states_valid(rob, se3) != 0]


def test_flat_map_chains_are_complete_and_budgets_count_chain_vertices():
    rob = O.robot("yaml")
    gm = common.make_map(100, 0.1, flat=True)
    om, acc = _accepted(gm, rob, 2048)
    g = PI.IncrementalPRM(om, rob, O.interpolate)
    n_interp_sum = 0
    for i, s in enumerate(acc[:60]):
        nv0 = g.nv
        m = g.add_valid_milestone(s)
        assert m == nv0 and g.is_milestone[m]
        # neighbours: the k = ceil(e (1 + 1/6) ln n) nearest PREDECESSORS, n counted with m itself
        if i:
            k = int(math.ceil(PI.K_PRM * math.log(nv0 + 1)))
            first_hops = [v for v in g.adj[m]]
            assert len(first_hops) == min(k, nv0)           # one chain (or direct edge) per neighbour, all valid on a flat map
            assert all(v < m or not g.is_milestone[v] for v in first_hops)
    # flat map: every interior state is valid, so every chain is complete: no dangling tip (degree >= 2; more when a
    # later milestone picked the chain vertex as a neighbour)
    chain = [v for v in range(g.nv) if not g.is_milestone[v]]
    assert chain and all(len(g.adj[v]) >= 2 for v in chain) and any(len(g.adj[v]) > 2 for v in chain)
    # chain vertices are nearest-neighbour targets of later milestones (prm_motion_cost.cpp:364)
    assert g.in_nn[:g.nv].all()
    # sampleGraph's vertex budget counts chain vertices (num_vertices, :171): far fewer milestones than the budget
    om2, acc2 = _accepted(gm, rob, 4096)
    s = np.array([-4.0, -4.0, acc2[0, 2], 0, 0, 0, 1.0])
    t = np.array([4.0, 4.0, acc2[0, 2], 0, 0, 0, 1.0])
    r = PI.build_and_solve(om2, rob, O.interpolate, acc2, s, t, max_n_vertices=3000)
    assert r["vertices"] >= 3000 and r["milestones_used"] < 600 and r["chain_vertices"] > 2000
    assert r["path"] is not None and r["path_cost"] >= 8 * math.sqrt(2) / 0.5 - 1e-9


def test_invalid_interior_state_keeps_the_valid_prefix():
    """A chain that meets an invalid interior state keeps its valid prefix as dangling vertices (degree 1 at the
    tip) and gets no edge to the neighbour (prm_motion_cost.cpp:353-377)."""
    rob = O.robot("yaml")
    gm = common.make_map(160, 0.04, seed=1234)
    om, acc = _accepted(gm, rob, 1 << 14)
    g = PI.IncrementalPRM(om, rob, O.interpolate)
    for s in acc[:400]:
        g.add_valid_milestone(s)
    tips = [v for v in range(g.nv) if not g.is_milestone[v] and len(g.adj[v]) == 1]
    assert tips, "no truncated chain on an obstacle map?"
    for v in tips[:20]:
        assert om.states_valid(rob, g.verts[v][None])[0] == 1      # the prefix itself is valid
    # every vertex in the graph is a valid state; edges never connect two milestones further apart than one 0.5 m step
    assert om.states_valid(rob, g.verts[:g.nv]).all()
    for (a, b) in list(g.edges)[:2000]:
        dxy = g.verts[a, :2] - g.verts[b, :2]
        assert math.hypot(dxy[0], dxy[1]) < 0.5 + 1e-9


def test_c1_planner_connects_goal_to_start_directly():
    """LazyPRMStarMinUpdate adds start and goal first (lazy_prm_star_min_update.cpp:507-535): the goal's only
    predecessor is the start, so the direct edge is in the graph and, on the flat C1 map, it is the answer."""
    rob = O.robot("yaml")
    gm = common.make_map(100, 0.1, flat=True)
    om, acc = _accepted(gm, rob, 2048)
    s = np.array([-4.0, -4.0, acc[0, 2], 0, 0, 0, 1.0])
    t = np.array([4.0, 4.0, acc[0, 2], 0, 0, 0, 1.0])
    r = PI.lazy_prm_star_min_update(om, rob, acc, s, t, 500)
    assert (0, 1) in r["graph"].edges and len(r["path"]) == 2
    assert abs(r["path_cost"] - 8 * math.sqrt(2) / 0.5) < 1e-9
