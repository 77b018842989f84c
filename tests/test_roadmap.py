"""'Next' row N1: the batched roadmap front end.  OMPL is not available here, so there is
no reference run to compare plans with (parity unpinned); every stage is checked against an
independent restatement instead: k-NN vs numpy brute force, edge verdicts and the final path vs the CPU
oracle, edge costs vs a numpy restatement of PathLengthObjective, the search vs scipy's Dijkstra."""
import numpy as np
import pytest

import common
import oracle_py as O

pytestmark = pytest.mark.gpu


def _se3_distance(a, b):
    """OMPL SE3StateSpace::distance: R^3 L2 + SO3 arc length (SO3StateSpace.cpp arcLength)."""
    dp = np.sqrt(((a[:, None, :3] - b[None, :, :3]) ** 2).sum(-1))
    dq = np.abs((a[:, None, 3:] * b[None, :, 3:]).sum(-1))
    arc = np.where(dq > 1.0 - 1e-9, 0.0, np.arccos(np.minimum(dq, 1.0)))
    return dp + arc


def _valid_state_near(ctx, gm, xy, rng, tries=4000):
    """A valid state near (x, y): sampler states filtered by distance."""
    se3 = ctx.sample_states(99, 0, 1 << 16)
    ok = ctx.validate_states(se3) != 0
    cand = se3[ok]
    d = np.hypot(cand[:, 0] - xy[0], cand[:, 1] - xy[1])
    return cand[np.argmin(d)]


@pytest.fixture(scope="module")
def planning_setup():
    from art_planner_amd.context import Context
    from synthetic import make_map
    gm = make_map(200, 0.04, seed=5)
    ctx = Context(0, "yaml")
    ctx.upload_map(gm)
    rng = np.random.default_rng(3)
    start = _valid_state_near(ctx, gm, (gm.pos_x - 2.6, gm.pos_y - 2.6), rng)
    goal = _valid_state_near(ctx, gm, (gm.pos_x + 2.6, gm.pos_y + 2.6), rng)
    yield gm, ctx, start, goal
    ctx.close()


def test_knn_matches_bruteforce(planning_setup):
    from art_planner_amd.roadmap import Roadmap
    gm, ctx, start, goal = planning_setup
    rm = Roadmap(ctx, start, goal, n_milestones=3000, seed=7)
    st, d = rm.stats(), rm.export()
    nv, k = st["vertices"], st["k"]
    assert nv == 3002 and k == int(np.ceil(np.e * (1 + 1 / 6) * np.log(nv)))
    V = d["verts"]
    assert np.array_equal(V[0], start) and np.array_equal(V[1], goal)
    # milestones are exactly the first accepted states of the sample stream
    se3 = ctx.sample_states(7, 0, int(st["samples_drawn"]))
    acc = se3[ctx.validate_states(se3) != 0]
    assert np.array_equal(V[2:], acc[:3000])
    D = _se3_distance(V, V)
    np.fill_diagonal(D, np.inf)
    order = np.argsort(D, axis=1, kind="stable")[:, :k]
    ref_d = np.take_along_axis(D, order, axis=1)
    assert np.abs(d["knn_dist"] - ref_d).max() < 1e-9
    same = d["knn"] == order
    # a different neighbour is only acceptable on a (near-)tie of the distances
    assert np.abs(np.take_along_axis(D, d["knn"].astype(np.int64), axis=1) - ref_d)[~same].max(initial=0.0) < 1e-9
    assert same.mean() > 0.999
    # candidate edges = symmetrised k-NN pairs, unique, sorted
    pairs = set()
    for i in range(nv):
        for j in d["knn"][i]:
            pairs.add((min(i, int(j)), max(i, int(j))))
    assert sorted(pairs) == [tuple(e) for e in d["edges"].tolist()]
    rm.close()


def test_edges_costs_and_path_against_oracle_and_scipy(planning_setup):
    from scipy.sparse import csr_matrix
    from scipy.sparse.csgraph import dijkstra
    from art_planner_amd.roadmap import Roadmap
    gm, ctx, start, goal = planning_setup
    rob = O.robot("yaml")
    om = O.OracleMap(gm)
    rm = Roadmap(ctx, start, goal, n_milestones=4000, seed=11)
    d0 = rm.export()
    V, E = d0["verts"], d0["edges"].astype(np.int64)
    # edge verdicts: the 0.5 m interpolation rule of addValidMilestone, by the CPU oracle
    sub = np.random.default_rng(0).choice(len(E), 20000, replace=False)
    vo, no = om.edges_interp_valid(rob, V[E[sub, 0]], V[E[sub, 1]])
    assert np.array_equal(d0["edge_valid"][sub], vo) and np.array_equal(d0["edge_interp"][sub], no)
    assert 0.05 < d0["edge_valid"].mean() < 0.999
    # Euclidean objective: a chain of interpolated sub-edges on a straight segment sums to its length / v
    eu = np.sqrt(((V[E[:, 0], :3] - V[E[:, 1], :3]) ** 2).sum(-1)) / 0.5
    assert np.abs(d0["edge_cost"] - eu).max() < 1e-9
    path, cost, replans = rm.solve()
    assert path is not None and np.array_equal(path[0], start) and np.array_equal(path[-1], goal)
    d1 = rm.export()
    keep = (d1["edge_valid"] != 0) & (d1["edge_removed"] == 0)
    n = len(V)
    W = csr_matrix((d1["edge_cost"][keep], (E[keep, 0], E[keep, 1])), shape=(n, n))
    ref = dijkstra(W, directed=False, indices=0)[1]
    assert abs(cost - ref) < 1e-9 * max(1.0, ref)
    # the path: consecutive vertices are roadmap edges; cost = sum of their costs; every edge passes the
    # oracle's discrete motion validator (constructSolution's final check)
    seg = np.sqrt(((path[1:, :3] - path[:-1, :3]) ** 2).sum(-1)) / 0.5
    assert abs(seg.sum() - cost) < 1e-9 * max(1.0, cost)
    assert om.check_motions(rob, path[:-1], path[1:])[0].all()
    assert cost >= np.linalg.norm(goal[:3] - start[:3]) / 0.5 - 1e-12
    assert replans == int(d1["edge_removed"].sum())
    rm.close()


def test_directional_cost_and_invalid_endpoints(planning_setup):
    from art_planner_amd.roadmap import Roadmap
    from art_planner_amd._capi import ArtpError
    gm, ctx, start, goal = planning_setup
    rm = Roadmap(ctx, start, goal, n_milestones=1500, seed=3, objective=1)
    d = rm.export()
    V, E = d["verts"], d["edges"].astype(np.int64)

    def yaw(q):
        return np.float32(np.arctan2(2 * (q[:, 3] * q[:, 2] + q[:, 0] * q[:, 1]),
                                     1 - 2 * (q[:, 1] ** 2 + q[:, 2] ** 2))).astype(np.float64)

    def cost(a, b):  # PathLengthObjective::motionCost with use_directional_cost
        dx, dy = b[:, 0] - a[:, 0], b[:, 1] - a[:, 1]
        y1, y2 = yaw(a[:, 3:]), yaw(b[:, 3:])
        dd = np.abs(y1 - y2)
        dyaw = np.where(dd > np.pi, 2 * np.pi - dd, dd)
        lon = np.cos(y1) * dx + np.sin(y1) * dy
        lat = -np.sin(y1) * dx + np.cos(y1) * dy
        return np.maximum(np.maximum(np.abs(lon) / 0.5, np.abs(lat) / 0.1), np.abs(dyaw) / 0.5)

    direct = d["edge_interp"] == 0  # no interior states: the chain is the edge itself
    assert direct.sum() > 100
    assert np.abs(d["edge_cost"][direct] - cost(V[E[direct, 0]], V[E[direct, 1]])).max() < 1e-9
    # chains: interpolate like OMPL and add the sub-edge costs
    idx = np.nonzero(~direct)[0][:300]
    for e in idx:
        a, b, ni = V[E[e, 0]], V[E[e, 1]], int(d["edge_interp"][e])
        pts = [a] + [O.interpolate(a, b, s / (ni + 1)) for s in range(1, ni + 1)] + [b]
        pts = np.array(pts)
        assert abs(cost(pts[:-1], pts[1:]).sum() - d["edge_cost"][e]) < 1e-8
    path, c, _ = rm.solve()
    assert path is not None and c > 0
    rm.close()
    bad = start.copy()
    bad[2] -= 5.0  # under the terrain: not a valid state
    with pytest.raises(ArtpError):
        Roadmap(ctx, bad, goal, n_milestones=100)


def test_flat_map_path_is_near_the_straight_line():
    """BASELINE config C1 (flat 100 x 100 @ 0.1 m): with no obstacles the PRM* path converges to the
    straight segment; the cost is bounded below by it and, at this density, within 10 %."""
    from art_planner_amd.context import Context
    from art_planner_amd.roadmap import Roadmap
    from synthetic import make_map
    gm = make_map(100, 0.1, flat=True)
    ctx = Context(0, "yaml")
    ctx.upload_map(gm)
    se3 = ctx.sample_states(1, 0, 4096)
    ok = se3[ctx.validate_states(se3) != 0]
    start = ok[np.argmin(np.hypot(ok[:, 0] + 3.0, ok[:, 1] + 3.0))]
    goal = ok[np.argmin(np.hypot(ok[:, 0] - 3.0, ok[:, 1] - 3.0))]
    rm = Roadmap(ctx, start, goal, n_milestones=5000, seed=2)
    path, cost, _ = rm.solve()
    lb = np.linalg.norm(goal[:3] - start[:3]) / 0.5
    assert path is not None and lb - 1e-12 <= cost < 1.10 * lb
    # simplify_solution: on an obstacle-free map the cheapest chain of shortcuts is the segment itself
    simp, scost = rm.simplify(path)
    assert len(simp) == 2 and abs(scost - lb) < 1e-12
    rm.close()
    # the C1 query of SURVEY.md 8d: start (-4, -4, yaw 0) -> goal (4, 4); optimum = 8*sqrt(2) m / 0.5 m/s
    z0 = float(ok[0, 2])
    s1 = np.array([-4.0, -4.0, z0, 0, 0, 0, 1.0])
    g1 = np.array([4.0, 4.0, z0, 0, 0, 0, 1.0])
    assert ctx.validate_states(np.stack([s1, g1])).all()
    rm = Roadmap(ctx, s1, g1, n_milestones=10000, seed=42)
    p1, c1, _ = rm.solve()
    q1, d1 = rm.simplify(p1)
    assert abs(d1 - 8 * np.sqrt(2) / 0.5) < 1e-4 and c1 < 1.05 * d1
    rm.close()
    ctx.close()


def test_roadmap_priced_through_an_external_cost_function(planning_setup):
    """artp_cost_set_external_query = the MotionCostFunc seam of PRMMotionCostMaintainer (prm_motion_cost.cpp:27-73): the
    roadmap's learned-cost batches go [B x 6] -> the caller's function -> [B x 3].  (a) a function that forwards to
    artp_cost_query reproduces device pricing bit for bit; (b) an analytic function prices the graph by its own numbers,
    without any weights in play; (c) a failing function fails the build with ARTP_ERR_COST_FUNC = "Motion cost call
    failed" (motion_cost_objective.cpp:78-83) and the context stays usable."""
    import os
    import sys
    sys.path.insert(0, os.path.join(common.ROOT, "oracle"))
    sys.path.insert(0, os.path.join(common.ROOT, "tools"))
    import motion_cost_oracle as mo
    import convert_weights
    from art_planner_amd._capi import ArtpError
    from art_planner_amd.roadmap import Roadmap
    gm, ctx, start, goal = planning_setup
    ctx.cost_load_weights(convert_weights.to_blob(mo.random_params(0)))
    elv = np.ascontiguousarray(gm["elevation"][::-1, ::-1]).astype(np.float32)
    ctx.cost_update_map(elv, gm.res, gm.len_x, gm.len_y, gm.pos_x, gm.pos_y)
    kw = dict(n_milestones=1200, seed=5, k_neighbors=40, objective=2, cost_weights=(0.25, 1.0, 5.0), risk_threshold=0.55)
    rm = Roadmap(ctx, start, goal, **kw)
    dev = rm.export()
    rm.close()
    seen = []

    def forward(edges):
        seen.append(len(edges))
        return ctx.cost_query(edges)

    try:
        ctx.cost_set_external_query(forward)
        rm = Roadmap(ctx, start, goal, **kw)
        ext = rm.export()
        rm.close()
        assert seen and sum(seen) >= len(dev["edge_cost"])
        assert np.array_equal(ext["edges"], dev["edges"]) and np.array_equal(ext["edge_cost"], dev["edge_cost"])

        # (b) energy 1 per query, time = lateral length, no risk: an edge of n_interp interior states costs
        # 0.25 (n_interp + 1) + its sub-edges' lengths
        def analytic(edges):
            out = np.zeros((len(edges), 3), np.float32)
            out[:, 0] = 1.0
            out[:, 1] = np.hypot(edges[:, 0] - edges[:, 3], edges[:, 1] - edges[:, 4])
            return out

        ctx.cost_set_external_query(analytic)
        rm = Roadmap(ctx, start, goal, **kw)
        d = rm.export()
        V, E = d["verts"], d["edges"].astype(np.int64)
        length = np.hypot(V[E[:, 0], 0] - V[E[:, 1], 0], V[E[:, 0], 1] - V[E[:, 1], 1])
        want = 0.25 * (d["edge_interp"] + 1.0) + length
        assert np.isfinite(d["edge_cost"]).all() and np.abs(d["edge_cost"] - want).max() < 1e-4
        path, cost, _ = rm.solve()
        assert path is not None and cost >= np.hypot(*(start[:2] - goal[:2])) - 1e-6
        rm.close()
        # (c)
        ctx.cost_set_external_query(lambda edges: None)
        with pytest.raises(ArtpError) as ei:
            Roadmap(ctx, start, goal, **kw)
        assert ei.value.status == -9 and "Motion cost call failed" in str(ei.value)
    finally:
        ctx.cost_set_external_query(None)
    rm = Roadmap(ctx, start, goal, **kw)
    assert np.array_equal(rm.export()["edge_cost"], dev["edge_cost"])   # device pricing again
    rm.close()


def test_learned_cost_objective_matches_per_subedge_queries(planning_setup):
    """objective 2 = PRMMotionCostMaintainer::updateEdges: every sub-edge of a chain is one EdgeMatrix row
    (target x y yaw, start x y yaw); chain cost = sum of getCost over its rows, infinite as soon as one
    row's risk exceeds the threshold.  Checked against artp_cost_query on rows rebuilt in numpy."""
    import os
    import sys
    sys.path.insert(0, os.path.join(common.ROOT, "oracle"))
    sys.path.insert(0, os.path.join(common.ROOT, "tools"))
    import motion_cost_oracle as mo
    import convert_weights
    from art_planner_amd.roadmap import Roadmap
    gm, ctx, start, goal = planning_setup
    ctx.cost_load_weights(convert_weights.to_blob(mo.random_params(0)))
    elv = np.ascontiguousarray(gm["elevation"][::-1, ::-1]).astype(np.float32)
    ctx.cost_update_map(elv, gm.res, gm.len_x, gm.len_y, gm.pos_x, gm.pos_y)
    w = (0.25, 1.0, 5.0)
    thr = 0.55  # random weights: risks spread around 0.5, so both feasible and infeasible chains occur
    rm = Roadmap(ctx, start, goal, n_milestones=1200, seed=5, k_neighbors=60, objective=2, cost_weights=w,
                 risk_threshold=thr)
    d = rm.export()
    V, E = d["verts"], d["edges"].astype(np.int64)

    def yaw(q):
        return np.float32(np.arctan2(2 * (q[3] * q[2] + q[0] * q[1]), 1 - 2 * (q[1] ** 2 + q[2] ** 2)))

    rows, owner = [], []
    sel = np.concatenate([np.nonzero(d["edge_interp"] > 0)[0][:400], np.nonzero(d["edge_interp"] == 0)[0][:400]])
    assert (d["edge_interp"][sel] > 0).sum() > 50
    for e in sel:
        a, b, ni = V[E[e, 0]], V[E[e, 1]], int(d["edge_interp"][e])
        pts = [a] + [O.interpolate(a, b, s / (ni + 1)) for s in range(1, ni + 1)] + [b]
        for s0, s1 in zip(pts[:-1], pts[1:]):
            rows.append([s1[0], s1[1], yaw(s1[3:]), s0[0], s0[1], yaw(s0[3:])])
            owner.append(e)
    c3 = ctx.cost_query(np.array(rows, np.float32)).astype(np.float64)
    owner = np.array(owner)
    for e in sel:
        r = c3[owner == e]
        ref = np.inf if (r[:, 2] > np.float32(thr)).any() else (r[:, 0] * np.float32(w[0]) + r[:, 1] * np.float32(w[1])
                                                                + r[:, 2] * np.float32(w[2])).sum()
        got = d["edge_cost"][e]
        assert (np.isinf(ref) and np.isinf(got)) or abs(got - ref) <= 1e-5 * max(1.0, abs(ref)), (e, got, ref)
    fin = np.isfinite(d["edge_cost"])
    assert 0.02 < fin.mean() < 0.98
    path, cost, _ = rm.solve()
    if path is not None:  # every edge of the plan is feasible and the cost adds up
        assert np.isfinite(cost) and cost > 0
    rm.close()


def test_revalidate_after_map_update_and_new_query(planning_setup):
    """The kept roadmap after a map change (LazyPRMStarMinUpdate's use case): a wall is raised across the
    plan; revalidate must equal a from-scratch evaluation of the same vertices / edges on the new map, the
    new plan avoids the wall; then a new start / goal are attached to the kept roadmap."""
    from scipy.sparse import csr_matrix
    from scipy.sparse.csgraph import dijkstra
    from art_planner_amd.roadmap import Roadmap
    gm, ctx, start, goal = planning_setup
    rob = O.robot("yaml")
    rm = Roadmap(ctx, start, goal, n_milestones=4000, seed=13)
    path0, cost0, _ = rm.solve()
    assert path0 is not None
    # raise a block of terrain under the middle of the plan (both layers, like a new obstacle)
    mid = path0[len(path0) // 2]
    ix = int((gm.pos_x + 0.5 * gm.len_x - mid[0]) / gm.res)
    iy = int((gm.pos_y + 0.5 * gm.len_y - mid[1]) / gm.res)
    r0, c0 = max(ix - 6, 0), max(iy - 6, 0)
    layers = {}
    for slot, name in ((0, "elevation"), (1, "elevation_masked")):
        lay = gm[name].copy()
        patch = lay[r0:r0 + 12, c0:c0 + 12]
        if slot == 0:
            patch = np.where(np.isfinite(patch), patch, np.float32(0)) + np.float32(0.6)
        else:
            patch = np.full_like(patch, -np.inf)
        lay[r0:r0 + 12, c0:c0 + 12] = patch
        layers[name] = lay
        ctx.update_layer_rect(slot, np.asfortranarray(patch), r0, c0)
    info = rm.revalidate()
    assert info["invalid_vertices"] > 0 and info["valid_edges_after"] < info["valid_edges_before"]
    d = rm.export()
    V, E = d["verts"], d["edges"].astype(np.int64)
    import copy
    gm2 = copy.deepcopy(gm)
    gm2.layers["elevation"] = np.asfortranarray(layers["elevation"])
    gm2.layers["elevation_masked"] = np.asfortranarray(layers["elevation_masked"])
    om2 = O.OracleMap(gm2)
    vok = om2.states_valid(rob, V) != 0
    sub = np.random.default_rng(1).choice(len(E), 15000, replace=False)
    eo, _ = om2.edges_interp_valid(rob, V[E[sub, 0]], V[E[sub, 1]])
    expect = (eo != 0) & vok[E[sub, 0]] & vok[E[sub, 1]]
    assert np.array_equal(d["edge_valid"][sub] != 0, expect)
    assert info["invalid_vertices"] == int((~vok).sum())
    assert info["start_valid"] and info["goal_valid"]
    path1, cost1, _ = rm.solve()
    d1 = rm.export()
    keep1 = (d1["edge_valid"] != 0) & (d1["edge_removed"] == 0)
    W1 = csr_matrix((d1["edge_cost"][keep1], (E[keep1, 0], E[keep1, 1])), shape=(len(V),) * 2)
    ref1 = dijkstra(W1, directed=False, indices=0)[1]
    assert (path1 is None) == bool(np.isinf(ref1))
    if path1 is not None:
        assert abs(cost1 - ref1) < 1e-9 * max(1.0, ref1) and cost1 >= cost0 - 1e-12
        assert om2.states_valid(rob, path1).all() and om2.check_motions(rob, path1[:-1], path1[1:])[0].all()
    # new query on the kept roadmap
    se3 = ctx.sample_states(77, 0, 1 << 14)
    okv = se3[ctx.validate_states(se3) != 0]
    s2 = okv[np.argmin(np.hypot(okv[:, 0] - (gm.pos_x + 2.5), okv[:, 1] - (gm.pos_y - 2.5)))]
    g2 = okv[np.argmin(np.hypot(okv[:, 0] - (gm.pos_x - 2.5), okv[:, 1] - (gm.pos_y + 2.5)))]
    rm.set_query(s2, g2)
    d2 = rm.export()
    V2, E2 = d2["verts"], d2["edges"].astype(np.int64)
    assert np.array_equal(V2[0], s2) and np.array_equal(V2[1], g2) and np.array_equal(V2[2:], V[2:])
    assert np.array_equal(E2[E2[:, 0] >= 2], E[E[:, 0] >= 2])           # the rest of the roadmap is untouched
    D = _se3_distance(V2[:2], V2)
    D[0, 0] = D[1, 1] = np.inf
    # vertices the updated map invalidated stay in the roadmap but are no neighbour targets (the reference's invalid
    # vertices are not in nn_)
    bad = np.flatnonzero(~vok)
    D[:, bad[bad >= 2]] = np.inf
    k = rm.stats()["k"]
    for q in (0, 1):
        ref = set(np.argsort(D[q], kind="stable")[:k].tolist())
        got = set(int(v) if u == q else int(u) for u, v in E2[(E2[:, 0] == q) | (E2[:, 1] == q)])
        assert ref <= got                                                # its own k nearest (plus the other's pick)
    head = E2[:, 0] < 2
    ho, hn = om2.edges_interp_valid(rob, V2[E2[head, 0]], V2[E2[head, 1]])
    assert np.array_equal(d2["edge_valid"][head], ho) and np.array_equal(d2["edge_interp"][head], hn)
    path2, cost2, _ = rm.solve()
    keep = (d2["edge_valid"] != 0)
    rem = rm.export()["edge_removed"] != 0
    W = csr_matrix((d2["edge_cost"][keep & ~rem], (E2[keep & ~rem, 0], E2[keep & ~rem, 1])), shape=(len(V2),) * 2)
    ref_cost = dijkstra(W, directed=False, indices=0)[1]
    assert (path2 is None and np.isinf(ref_cost)) or abs(cost2 - ref_cost) < 1e-9 * max(1.0, ref_cost)
    rm.close()
    ctx.upload_map(gm)  # restore the module fixture's map


def test_grow_keeps_valid_milestones_and_continues_the_sample_stream(planning_setup):
    """artp_roadmap_grow (sampleGraph between queries): on an unchanged map the grown roadmap equals a roadmap
    built with the larger milestone count in one go (same sample stream, same connection rule); after a map
    change the milestones the map invalidated are dropped, the others stay in order, and the plan is valid."""
    from art_planner_amd.roadmap import Roadmap
    gm, ctx, start, goal = planning_setup
    rob = O.robot("yaml")
    rm = Roadmap(ctx, start, goal, n_milestones=1500, seed=21)
    st0 = rm.stats()
    info = rm.grow(1000)
    assert info == {"kept": 1500, "dropped": 0}
    d = rm.export()
    assert rm.stats()["vertices"] == 2502 and rm.stats()["k"] >= st0["k"]
    ref = Roadmap(ctx, start, goal, n_milestones=2500, seed=21)
    dr = ref.export()
    # both draw accepted samples in index order; the grown one skipped the unused tail of its first batch, so
    # compare the common prefix exactly and the whole graph through its own invariants
    assert np.array_equal(d["verts"][:1502], dr["verts"][:1502])
    V, E = d["verts"], d["edges"].astype(np.int64)
    om = O.OracleMap(gm)
    assert om.states_valid(rob, V).all()
    sub = np.random.default_rng(4).choice(len(E), 6000, replace=False)
    eo, no = om.edges_interp_valid(rob, V[E[sub, 0]], V[E[sub, 1]])
    assert np.array_equal(d["edge_valid"][sub], eo) and np.array_equal(d["edge_interp"][sub], no)
    D = _se3_distance(V[:40], V)
    D[np.arange(40), np.arange(40)] = np.inf
    k = rm.stats()["k"]
    for q in range(40):
        want = set(np.argsort(D[q], kind="stable")[:k].tolist())
        got = set(int(v) if u == q else int(u) for u, v in E[(E[:, 0] == q) | (E[:, 1] == q)])
        assert want <= got
    p1, c1, _ = rm.solve()
    assert p1 is not None and om.check_motions(rob, p1[:-1], p1[1:])[0].all()
    ref.close()
    # map change: a block in the middle of the plan becomes an obstacle
    mid = p1[len(p1) // 2]
    ix = int((gm.pos_x + 0.5 * gm.len_x - mid[0]) / gm.res)
    iy = int((gm.pos_y + 0.5 * gm.len_y - mid[1]) / gm.res)
    r0, c0 = max(ix - 6, 0), max(iy - 6, 0)
    import copy
    gm2 = copy.deepcopy(gm)
    for slot, name in ((0, "elevation"), (1, "elevation_masked")):
        lay = gm[name].copy()
        patch = lay[r0:r0 + 12, c0:c0 + 12]
        patch = (np.where(np.isfinite(patch), patch, np.float32(0)) + np.float32(0.6)) if slot == 0 \
            else np.full_like(patch, -np.inf)
        lay[r0:r0 + 12, c0:c0 + 12] = patch
        gm2.layers[name] = np.asfortranarray(lay)
        ctx.update_layer_rect(slot, np.asfortranarray(patch), r0, c0)
    om2 = O.OracleMap(gm2)
    vok = om2.states_valid(rob, V) != 0
    info2 = rm.grow(500)
    assert info2["dropped"] == int((~vok[2:]).sum()) > 0 and info2["kept"] == int(vok[2:].sum())
    d2 = rm.export()
    V2, E2 = d2["verts"], d2["edges"].astype(np.int64)
    assert len(V2) == 2 + info2["kept"] + 500
    assert np.array_equal(V2[2:2 + info2["kept"]], V[2:][vok[2:]])        # kept milestones, order preserved
    assert om2.states_valid(rob, V2).all()
    sub = np.random.default_rng(5).choice(len(E2), 6000, replace=False)
    eo, _ = om2.edges_interp_valid(rob, V2[E2[sub, 0]], V2[E2[sub, 1]])
    assert np.array_equal(d2["edge_valid"][sub], eo)
    p2, c2, _ = rm.solve()
    if p2 is not None:
        assert om2.states_valid(rob, p2).all() and om2.check_motions(rob, p2[:-1], p2[1:])[0].all()
    rm.close()
    ctx.upload_map(gm)  # restore the module fixture's map


def test_simplify_path_is_valid_and_never_worse(planning_setup):
    """The batched shortcutting: the result is a subsequence of the plan from start to goal, every edge of
    it passes the oracle's interpolation rule and discrete motion validator, and its cost is <= the plan's
    (and equal to brute-force DP over the oracle-validated shortcut graph)."""
    from art_planner_amd.roadmap import Roadmap
    gm, ctx, start, goal = planning_setup
    rob = O.robot("yaml")
    om = O.OracleMap(gm)
    rm = Roadmap(ctx, start, goal, n_milestones=4000, seed=11)
    path, cost, _ = rm.solve()
    assert path is not None and len(path) > 4
    simp, scost = rm.simplify(path)
    assert np.array_equal(simp[0], path[0]) and np.array_equal(simp[-1], path[-1])
    idx = [int(np.nonzero((path == s).all(axis=1))[0][0]) for s in simp]
    assert idx == sorted(idx) and len(set(idx)) == len(idx)      # a subsequence
    assert scost <= cost + 1e-9 and len(simp) <= len(path)
    seg = np.sqrt(((simp[1:, :3] - simp[:-1, :3]) ** 2).sum(-1)) / 0.5
    assert abs(seg.sum() - scost) < 1e-9 * max(1.0, scost)
    assert om.edges_interp_valid(rob, simp[:-1], simp[1:])[0].all()
    assert om.check_motions(rob, simp[:-1], simp[1:])[0].all()
    # brute force: DP over all pairs validated by the oracle
    n = len(path)
    ii, jj = np.triu_indices(n, 1)
    ok = (om.edges_interp_valid(rob, path[ii], path[jj])[0] != 0) & (om.check_motions(rob, path[ii], path[jj])[0] != 0)
    w = np.sqrt(((path[jj, :3] - path[ii, :3]) ** 2).sum(-1)) / 0.5
    best = np.full(n, np.inf)
    best[0] = 0.0
    for a, b, o, ww in zip(ii, jj, ok, w):   # (i, j) in lexicographic order: all edges into j come before j's out-edges
        if o and best[a] + ww < best[b]:
            best[b] = best[a] + ww
    assert abs(best[-1] - scost) < 1e-9 * max(1.0, scost)
    rm.close()


def test_replan_loop_example_runs():
    """examples/replan_loop.py: raw map -> device preprocessing -> install -> roadmap upkeep -> plan, 5 cycles."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("replan_loop", os.path.join(common.ROOT, "examples", "replan_loop.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    stats = mod.main(5, verbose=False)
    assert len(stats) >= 3 and np.isfinite(stats).all()
    assert np.median(stats.sum(axis=1)) < 100.0   # the reference's 10 Hz budget, with two orders of margin


# ---- round 2: PRMMotionCostMaintainer::sampleGraph's budgets and in-build re-weighting, LazyPRM*'s growth loop --------
def _preprocessed_ctx(n=250, seed=77):
    from art_planner_amd.context import Context
    from synthetic import make_map
    gm = make_map(n, 0.04, seed=seed)
    ctx = Context(0, "yaml")
    pm = ctx.preprocess_map(gm["elevation"], gm.len_x, gm.len_y, gm.pos_x, gm.pos_y,
                            traversability=gm["traversability"], use_inverse_vertex_density=1)
    pm.install()
    se3 = ctx.sample_states(5, 0, 6000)
    acc = se3[ctx.validate_states(se3) != 0]
    d = np.hypot(acc[:, None, 0] - acc[None, :300, 0], acc[:, None, 1] - acc[None, :300, 1])
    i, j = np.unravel_index(np.argmax(d), d.shape)
    return gm, ctx, pm, acc[i], acc[j]


def _cell_counts(gm, verts, cell=10):
    """Vertices per (cell x cell)-cell block of the map, over the blocks the sampler can reach at all."""
    ri = ((gm.pos_x + 0.5 * gm.len_x - verts[:, 0]) / gm.res).astype(int).clip(0, gm.rows - 1) // cell
    ci = ((gm.pos_y + 0.5 * gm.len_y - verts[:, 1]) / gm.res).astype(int).clip(0, gm.cols - 1) // cell
    nb = (gm.rows + cell - 1) // cell
    return np.bincount(ri * nb + ci, minlength=nb * nb).reshape(nb, nb)


@pytest.mark.gpu
def test_in_build_density_reweighting_follows_the_reference_rounds():
    """prm_motion_cost.cpp:190-193: every recompute_density_after_n_samples vertices Map::reApplyPreprocessing()
    recomputes the sampling distribution from the inverse vertex density.  (1) the number of re-weightings is the
    reference's floor(vertices / R) (none at the very end); (2) the FIRST R milestones are those of the fixed
    distribution (same sample stream), later ones differ; (3) the distribution the last round sampled from equals the
    numpy restatement of computeInverseSampleDensity over the vertices known at that point; (4) the vertex density
    gets flatter than with a fixed distribution (that is the purpose of the re-weighting)."""
    from art_planner_amd.roadmap import Roadmap
    from synthetic import cumulative_distribution
    import test_preprocess as TP
    gm, ctx, pm, s, g = _preprocessed_ctx()
    R, nm = 400, 1500
    fixed = Roadmap(ctx, s, g, n_milestones=nm, seed=3)
    vf = fixed.export()["verts"]
    fixed.close()
    sf = pm.layer("traversability_sample_filter")
    rw = Roadmap(ctx, s, g, n_milestones=nm, seed=3, recompute_density_after_n_samples=R, density_map=pm)
    st = rw.stats()
    vr = rw.export()["verts"]
    assert st["vertices"] == nm + 2 and st["reweightings"] == nm // R
    assert np.array_equal(vr[:2 + R], vf[:2 + R]) and not np.array_equal(vr[2 + R:2 + 2 * R], vf[2 + R:2 + 2 * R])
    # (3) the map now carries the distribution of the last re-weighting: the first 3 R milestones
    known = vr[2:2 + nm // R * R]
    tx = -((known[:, 0] - gm.pos_x) - 0.5 * gm.len_x)
    ty = -((known[:, 1] - gm.pos_y) - 0.5 * gm.len_y)
    cnt = np.zeros((gm.rows, gm.cols), np.float32)
    np.add.at(cnt, ((tx / np.float64(np.float32(gm.res))).astype(int).clip(0, gm.rows - 1),
                    (ty / np.float64(np.float32(gm.res))).astype(int).clip(0, gm.cols - 1)), 1.0)
    prm = ctx.params
    radius = (prm.torso_length + prm.torso_width) * 0.25
    k = int(6 * radius / gm.res)
    k += 1 if k % 2 == 0 else 0
    blurred = TP._blur_reflect101(cnt, TP._gauss_taps(k, radius / gm.res))
    got_blur = pm.layer("n_samples")
    assert np.abs(got_blur - blurred).max() < 1e-5 * max(1.0, blurred.max())
    prob = pm.layer("sample_probability")
    base = (np.float32(got_blur.max()) - got_blur) * sf
    assert np.allclose(prob / max(prob.max(), 1e-30), base / max(base.max(), 1e-30), atol=2e-5)
    cp, _ = cumulative_distribution(prob)
    with np.errstate(invalid="ignore"):
        assert np.nanmax(np.abs(pm.layer("cum_prob") - cp)) < 1e-5
    # (4) flatter: dispersion of the block counts over the reachable blocks
    blocks = sf.reshape(gm.rows // 10, 10, gm.cols // 10, 10).sum((1, 3)) > 60   # blocks that are mostly samplable
    cf, cr = _cell_counts(gm, vf[2:])[blocks], _cell_counts(gm, vr[2:])[blocks]
    assert cr.std() / cr.mean() < 0.93 * cf.std() / cf.mean(), (cr.std() / cr.mean(), cf.std() / cf.mean())
    # the kept roadmap keeps re-weighting when it grows, and survives losing its density map
    rw.grow(500)
    assert rw.stats()["reweightings"] > st["reweightings"]
    rw.set_density_map(None)
    rw.grow(100)
    rw.close()
    # restore the fixed distribution for whoever comes next
    pm.reweight_dev(None)
    pm.close()
    ctx.close()


@pytest.mark.gpu
def test_sample_graph_budgets():
    """max_n_edges / max_sample_time of PRMMotionCostMaintainer::sampleGraph (prm_motion_cost.cpp:171-185)."""
    from art_planner_amd.roadmap import Roadmap
    gm, ctx, pm, s, g = _preprocessed_ctx()
    full = Roadmap(ctx, s, g, n_milestones=3000, seed=3)
    ne_full = full.stats()["candidate_edges"]
    full.close()
    cut = Roadmap(ctx, s, g, n_milestones=3000, seed=3, max_n_edges=ne_full // 3)
    st = cut.stats()
    assert st["edge_budget_hit"] and st["candidate_edges"] <= ne_full // 3 and 3 < st["vertices"] < 3002
    assert not st["edge_budget_exceeded"]
    e = cut.export()
    assert e["edges"].max() < st["vertices"]
    # the vertices are a prefix of the unconstrained build's (same stream, the budget only stops it earlier)
    full = Roadmap(ctx, s, g, n_milestones=3000, seed=3)
    assert np.array_equal(e["verts"], full.export()["verts"][:st["vertices"]])
    full.close()
    cut.close()
    timed = Roadmap(ctx, s, g, n_milestones=2_000_000, seed=3, max_sample_time=0.002)
    st = timed.stats()
    assert st["time_budget_hit"] and 10 < st["vertices"] < 2_000_000, st
    timed.close()
    pm.close()
    ctx.close()


@pytest.mark.gpu
def test_solve_until_grows_while_planning():
    """LazyPRMStarMinUpdate::baseSolve (lazy_prm_star_min_update.cpp:552-615): the roadmap grows until the planning
    time is over and the best solution found is returned -- never worse than the first one."""
    from art_planner_amd.roadmap import Roadmap
    gm, ctx, pm, s, g = _preprocessed_ctx()
    rm = Roadmap(ctx, s, g, n_milestones=300, seed=9)
    p0, c0, _ = rm.solve()
    path, cost, info = rm.solve_until(0.15, 400)
    assert info["rounds"] >= 2 and info["vertices"] >= 302 + 400 * info["rounds"] - 5
    assert path is not None and cost <= (c0 if p0 is not None else np.inf) + 1e-12
    assert np.array_equal(path[0], s) and np.array_equal(path[-1], g)
    assert ctx.validate_states(path).all() and ctx.check_motions(path[:-1], path[1:]).all()
    # plan_time 0: exactly one solve, no growth
    nv = rm.stats()["vertices"]
    p1, c1, info1 = rm.solve_until(0.0, 400)
    # (the grown graph's own optimum may differ from the best of all rounds: k-nearest connections change as
    # the roadmap gets denser)
    assert info1["rounds"] == 0 and rm.stats()["vertices"] == nv and p1 is not None and np.isfinite(c1)
    rm.close()
    pm.close()
    ctx.close()


@pytest.mark.gpu
def test_device_search_matches_host_astar_and_scipy():
    """Roadmaps of >= 30 000 vertices search on the device (label-correcting relaxation): same cost as the host
    A* of the small-roadmap path and as scipy's Dijkstra on the exported graph."""
    from scipy.sparse import csr_matrix
    from scipy.sparse.csgraph import dijkstra
    from art_planner_amd.roadmap import Roadmap
    from art_planner_amd.context import Context
    from synthetic import make_map
    gm = make_map(400, 0.04, seed=1234)
    ctx = Context(0, "yaml")
    ctx.upload_map(gm)
    se3 = ctx.sample_states(5, 0, 8000)
    acc = se3[ctx.validate_states(se3) != 0]
    s = acc[np.argmin(acc[:, 0] + acc[:, 1])]
    g = acc[np.argmax(acc[:, 0] + acc[:, 1])]
    rm = Roadmap(ctx, s, g, n_milestones=40000, seed=11)
    path, cost, rep = rm.solve()
    assert path is not None
    e = rm.export()
    ok = (e["edge_valid"] != 0) & (e["edge_removed"] == 0) & np.isfinite(e["edge_cost"])
    nv = len(e["verts"])
    gr = csr_matrix((e["edge_cost"][ok], (e["edges"][ok, 0], e["edges"][ok, 1])), shape=(nv, nv))
    d = dijkstra(gr, directed=False, indices=0)
    assert abs(d[1] - cost) <= 1e-9 * cost
    # the returned states are roadmap vertices joined by usable edges whose costs add up to the reported cost
    idx = [int(np.flatnonzero((e["verts"] == p).all(1))[0]) for p in path]
    cost_of = {(int(a), int(b)): c for (a, b), c, o in zip(e["edges"], e["edge_cost"], ok) if o}
    tot = sum(cost_of[(min(a, b), max(a, b))] for a, b in zip(idx[:-1], idx[1:]))
    assert abs(tot - cost) <= 1e-9 * cost
    rm.close()
    ctx.close()


def test_costs_against_the_reference_planners_own_graph_construction():
    """Expected values from oracle/prm_incremental.py -- the reference's INCREMENTAL construction restated literally
    (addValidMilestone: predecessors-only k-NN with k at insertion time, valid chain prefixes kept as vertices and
    nearest-neighbour targets, vertex / edge budgets counted the reference's way; constructSolution) -- not from a
    restatement of the batched front end (VERDICT r2 missing #1).  Same accepted-state stream on both sides.
      C1 (lazy_prm_star_min_update, flat 100 x 100): the reference adds start and goal first, so the goal's only
         predecessor is the start: its graph holds the direct edge and the answer is the straight line, 8 sqrt(2) / 0.5.
         The batched front end reaches that cost to 1e-4 after simplification (north star), its raw roadmap path to 1 %.
      Perlin 160 x 160 (prm_motion_cost construction, objective 0): the batched roadmap over the SAME milestones must
         give a path cost within 2 % of the incremental graph's (measured: 0.4 % cheaper), never below the straight line."""
    import json
    import os
    import sys
    sys.path.insert(0, os.path.join(common.ROOT, "oracle"))
    import prm_incremental as PI
    from art_planner_amd.context import Context
    from art_planner_amd.roadmap import Roadmap
    from synthetic import make_map
    rob = O.robot("yaml")
    report = {}
    # ---- C1 ----
    gm = make_map(100, 0.1, flat=True)
    om = O.OracleMap(gm)
    ctx = Context(0, "yaml")
    ctx.upload_map(gm)
    se3 = ctx.sample_states(42, 0, 4096)   # ONE stream for both sides (the device sampler equals the oracle's to 1e-12)
    lab = om.states_valid(rob, se3)
    assert np.array_equal(ctx.validate_states(se3), lab)
    acc = se3[lab != 0]
    z0 = float(acc[0, 2])
    s = np.array([-4.0, -4.0, z0, 0, 0, 0, 1.0])
    g = np.array([4.0, 4.0, z0, 0, 0, 0, 1.0])
    ref = PI.lazy_prm_star_min_update(om, rob, acc, s, g, 2000)
    optimum = 8.0 * np.sqrt(2.0) / 0.5
    assert abs(ref["path_cost"] - optimum) < 1e-9 and len(ref["path"]) == 2   # the direct start-goal edge
    rm = Roadmap(ctx, s, g, n_milestones=2000, seed=42)
    assert np.array_equal(rm.export()["verts"][2:], acc[:2000])                # the same milestones
    p, c, _ = rm.solve()
    q, d = rm.simplify(p)
    assert abs(d - ref["path_cost"]) < 1e-4, (d, ref["path_cost"])
    assert 0.0 <= c - ref["path_cost"] < 0.01 * ref["path_cost"], (c, ref["path_cost"])
    report["c1"] = {"reference_construction_cost": ref["path_cost"], "batched_cost": c, "batched_simplified_cost": d}
    rm.close()
    ctx.close()
    # ---- Perlin 160 x 160, PRMMotionCost's construction ----
    gm = make_map(160, 0.04, seed=1234)
    om = O.OracleMap(gm)
    ctx = Context(0, "yaml")
    ctx.upload_map(gm)
    se3 = ctx.sample_states(42, 0, 1 << 15)
    lab = om.states_valid(rob, se3)
    assert np.array_equal(ctx.validate_states(se3), lab)
    acc = se3[lab != 0]

    def near(xy):
        return acc[np.argmin(np.hypot(acc[:, 0] - xy[0], acc[:, 1] - xy[1]))]

    s, g = near((gm.pos_x - 2.4, gm.pos_y - 2.4)), near((gm.pos_x + 2.4, gm.pos_y + 2.4))
    ref = PI.build_and_solve(om, rob, O.interpolate, acc, s, g)                # budgets: 10 000 vertices / 50 000 edges
    assert ref["path"] is not None and ref["chain_vertices"] > 500 and ref["lazy_removals"] >= 0
    m_used = ref["milestones_used"]
    rm = Roadmap(ctx, s, g, n_milestones=m_used, seed=42)
    assert np.array_equal(rm.export()["verts"][2:], acc[:m_used])
    p, c, _ = rm.solve()
    straight = float(np.linalg.norm(g[:3] - s[:3]) / 0.5)
    assert p is not None and c >= straight - 1e-9
    assert abs(c - ref["path_cost"]) < 0.02 * ref["path_cost"], (c, ref["path_cost"])
    # the batched path is a valid plan under the reference's own checks
    assert om.states_valid(rob, p).all() and om.check_motions(rob, p[:-1], p[1:])[0].all()
    report["perlin160"] = {"reference_construction": {k: v for k, v in ref.items() if k not in ("path", "graph")},
                           "batched_cost": c, "batched_vertices": int(rm.stats()["vertices"]),
                           "batched_candidate_edges": int(rm.stats()["candidate_edges"]), "straight_line": straight}
    rm.close()
    ctx.close()
    out = os.path.join(common.ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    json.dump(report, open(os.path.join(out, "roadmap_vs_reference_construction.json"), "w"), indent=1)


def _edge_set(ex):
    return {(int(u), int(v)) for (u, v), ok in zip(ex["edges"], ex["edge_valid"]) if ok}


@pytest.mark.gpu
def test_construction_2_reproduces_the_lazy_prm_star_graph():
    """artp_roadmap_params::construction = 2 against oracle/prm_incremental.lazy_prm_star_min_update (BASELINE config 1's
    planner restated literally): the SAME EDGE SET -- start, goal, then the milestones, every vertex connected to the
    k = ceil(e (1 + 1/6) ln n) nearest of its predecessors with n counted at its own insertion -- the same weights, the
    same lazy removals and the same answer.  On the flat C1 map that answer is the direct start-goal edge."""
    import os
    import sys
    sys.path.insert(0, os.path.join(common.ROOT, "oracle"))
    import prm_incremental as PI
    from art_planner_amd.context import Context
    from art_planner_amd.roadmap import Roadmap
    from synthetic import make_map
    rob = O.robot("yaml")
    for name, gm, n_ms in (("c1", make_map(100, 0.1, flat=True), 2000), ("perlin", make_map(160, 0.04, seed=1234), 1500)):
        om = O.OracleMap(gm)
        ctx = Context(0, "yaml")
        ctx.upload_map(gm)
        se3 = ctx.sample_states(42, 0, 1 << (13 if name == "c1" else 15))
        lab = om.states_valid(rob, se3)
        assert np.array_equal(ctx.validate_states(se3), lab)
        acc = se3[lab != 0]
        assert len(acc) >= n_ms
        if name == "c1":
            z0 = float(acc[0, 2])
            s, g = np.array([-4.0, -4.0, z0, 0, 0, 0, 1.0]), np.array([4.0, 4.0, z0, 0, 0, 0, 1.0])
        else:
            near = lambda xy: acc[np.argmin(np.hypot(acc[:, 0] - xy[0], acc[:, 1] - xy[1]))]
            s, g = near((gm.pos_x - 2.4, gm.pos_y - 2.4)), near((gm.pos_x + 2.4, gm.pos_y + 2.4))
        ref = PI.lazy_prm_star_min_update(om, rob, acc, s, g, n_ms)
        rm = Roadmap(ctx, s, g, n_milestones=n_ms, seed=42, construction=2)
        ex = rm.export()
        assert np.array_equal(ex["verts"][2:], acc[:n_ms]) and ex["verts"].shape[0] == ref["vertices"]
        p, c, removed = rm.solve()
        # the oracle's graph after ITS solve lacks the removed edges: compare against edges + removals on both sides
        ref_edges = set(ref["graph"].edges.keys())
        ex2 = rm.export()
        mine_all = {(int(u), int(v)) for u, v in ex2["edges"]}
        mine_left = {(int(u), int(v)) for (u, v), r in zip(ex2["edges"], ex2["edge_removed"]) if not r}
        assert len(mine_all) == ref["edges"], (name, len(mine_all), ref["edges"])
        assert mine_left == ref_edges, (name, len(mine_left ^ ref_edges))
        assert (ex["edge_interp"] == 0).all() and ex["edge_valid"].all()
        w_ref = np.array([ref["graph"].edges[(int(u), int(v))] for (u, v), r in zip(ex2["edges"], ex2["edge_removed"]) if not r])
        assert np.allclose(ex2["edge_cost"][ex2["edge_removed"] == 0], w_ref, rtol=1e-12, atol=0)
        assert removed == ref["lazy_removals"], (name, removed, ref["lazy_removals"])
        assert (p is None) == (ref["path"] is None)
        if p is not None:
            assert abs(c - ref["path_cost"]) < 1e-9 * max(1.0, c), (name, c, ref["path_cost"])
            assert np.allclose(p, ref["path"], atol=1e-12)
        if name == "c1":
            assert len(p) == 2 and abs(c - 8.0 * np.sqrt(2.0) / 0.5) < 1e-9
        rm.close()
        ctx.close()


@pytest.mark.gpu
def test_construction_1_reproduces_the_prm_motion_cost_graph():
    """artp_roadmap_params::construction = 1 against oracle/prm_incremental.build_and_solve (PRMMotionCost::
    addValidMilestone / sampleGraph / baseSolve / constructSolution restated literally) on the 160 x 160 Perlin map with
    the reference's budgets (10 000 vertices, 50 000 edges): the same milestones consumed, the same chain vertices (the
    valid prefixes of failing chains included), the SAME EDGE SET, the same lazy removals, the same path."""
    import os
    import sys
    sys.path.insert(0, os.path.join(common.ROOT, "oracle"))
    import prm_incremental as PI
    from art_planner_amd.context import Context
    from art_planner_amd.roadmap import Roadmap
    from synthetic import make_map
    rob = O.robot("yaml")
    gm = make_map(160, 0.04, seed=1234)
    om = O.OracleMap(gm)
    ctx = Context(0, "yaml")
    ctx.upload_map(gm)
    se3 = ctx.sample_states(42, 0, 1 << 15)
    lab = om.states_valid(rob, se3)
    assert np.array_equal(ctx.validate_states(se3), lab)
    acc = se3[lab != 0]
    near = lambda xy: acc[np.argmin(np.hypot(acc[:, 0] - xy[0], acc[:, 1] - xy[1]))]
    s, g = near((gm.pos_x - 2.4, gm.pos_y - 2.4)), near((gm.pos_x + 2.4, gm.pos_y + 2.4))
    for budget_v, budget_e in ((10000, 50000), (1200, 50000)):   # the edge budget ends the first, the vertex budget the second
        ref = PI.build_and_solve(om, rob, O.interpolate, acc, s, g, max_n_vertices=budget_v, max_n_edges=budget_e)
        G = ref["graph"]
        rm = Roadmap(ctx, s, g, n_milestones=budget_v, max_n_edges=budget_e, seed=42, construction=1)
        ex = rm.export()
        st = rm.stats()
        assert st["vertices"] == ref["vertices"], (st["vertices"], ref["vertices"])
        # oracle ids: insertion order, start and goal inserted last (each followed by its own chain vertices); here start = 0,
        # goal = 1, the rest in insertion order
        ms = np.flatnonzero(np.array(G.is_milestone))
        vs, vg = int(ms[-2]), int(ms[-1])
        assert np.array_equal(G.verts[vs], s) and np.array_equal(G.verts[vg], g)
        to_mine = np.empty(G.nv, np.int64)
        nxt = 2
        for o in range(G.nv):
            if o == vs:
                to_mine[o] = 0
            elif o == vg:
                to_mine[o] = 1
            else:
                to_mine[o] = nxt
                nxt += 1
        assert np.allclose(ex["verts"][to_mine], G.verts[:G.nv], atol=1e-12)
        chain = ~np.array(G.is_milestone)
        assert int(chain.sum()) == ref["chain_vertices"] and ref["chain_vertices"] > 100
        p, c, removed = rm.solve()
        ex2 = rm.export()
        mine_all = {(int(u), int(v)) for u, v in ex2["edges"]}
        mine_left = {(int(u), int(v)) for (u, v), r in zip(ex2["edges"], ex2["edge_removed"]) if not r}
        ref_left = {tuple(sorted((int(to_mine[a]), int(to_mine[b])))) for (a, b) in G.edges.keys()}
        assert len(mine_all) == ref["edges"], (len(mine_all), ref["edges"])
        assert mine_left == ref_left, len(mine_left ^ ref_left)
        assert removed == ref["lazy_removals"], (removed, ref["lazy_removals"])
        assert (p is None) == (ref["path"] is None), budget_v
        if budget_v == 10000:
            assert p is not None
        if p is not None:
            assert abs(c - ref["path_cost"]) < 1e-9 * c, (c, ref["path_cost"])
            assert np.allclose(p, ref["path"], atol=1e-12)
        assert st["samples_drawn"] > 0 and (st["edge_budget_hit"] or budget_v != 10000)
        rm.close()
    ctx.close()


@pytest.mark.gpu
def test_reference_order_roadmaps_upkeep():
    """grow / revalidate / set_query on roadmaps built in the reference planners' insertion orders.
    construction 2: growing by n milestones gives the graph a fresh build over n more milestones gives (the insertion
    order is the stream order either way).  construction 1: sampleGraph continued on the kept graph -- vertex ids and
    coordinates of the old graph are a prefix of the new one, every old edge is still there, the new milestones connect
    into the old graph; nothing is invalid after a revalidate on the unchanged map; a new query solves."""
    from art_planner_amd.context import Context
    from art_planner_amd.roadmap import Roadmap
    from synthetic import make_map
    gm = make_map(160, 0.04, seed=1234)
    ctx = Context(0, "yaml")
    ctx.upload_map(gm)
    se3 = ctx.sample_states(42, 0, 1 << 15)
    acc = se3[ctx.validate_states(se3) != 0]
    near = lambda xy: acc[np.argmin(np.hypot(acc[:, 0] - xy[0], acc[:, 1] - xy[1]))]
    s, g = near((gm.pos_x - 2.4, gm.pos_y - 2.4)), near((gm.pos_x + 2.4, gm.pos_y + 2.4))
    # ---- construction 2 ----
    a = Roadmap(ctx, s, g, n_milestones=600, seed=42, construction=2)
    assert a.grow(300) == {"kept": 600, "dropped": 0}
    b = Roadmap(ctx, s, g, n_milestones=900, seed=42, construction=2)
    ea, eb = a.export(), b.export()
    assert np.array_equal(ea["verts"], eb["verts"]) and np.array_equal(ea["edges"], eb["edges"])
    assert np.allclose(ea["edge_cost"], eb["edge_cost"], rtol=1e-15)
    rv = a.revalidate()
    assert rv["invalid_vertices"] == 0 and rv["valid_edges_after"] == len(ea["edges"]) and rv["start_valid"] and rv["goal_valid"]
    pa, ca, _ = a.solve()
    pb, cb, _ = b.solve()
    assert (pa is None) == (pb is None) and (pa is None or abs(ca - cb) < 1e-12)
    a.close()
    b.close()
    # ---- construction 1 ----
    r = Roadmap(ctx, s, g, n_milestones=2000, seed=42, construction=1)
    e0, st0 = r.export(), r.stats()
    assert 2000 <= st0["vertices"] < 2000 + 2 + 200        # the last milestone's chains and the query vertices overshoot
    out = r.grow(2000)
    e1, st1 = r.export(), r.stats()
    assert out == {"kept": st0["vertices"], "dropped": 0}
    assert st0["vertices"] + 2000 <= st1["vertices"] < st0["vertices"] + 2000 + 200 and st1["samples_drawn"] > st0["samples_drawn"]
    assert np.array_equal(e1["verts"][:st0["vertices"]], e0["verts"])
    old = {(int(u), int(v)) for u, v in e0["edges"]}
    new = {(int(u), int(v)) for u, v in e1["edges"]}
    assert old <= new and len(new) > len(old)
    assert any(u < st0["vertices"] <= v for u, v in new - old)      # new vertices are wired into the old graph
    assert (e1["edge_interp"] == 0).all() and e1["edge_valid"].all()
    # sub-edges are at most 0.5 m long in the plane (or direct edges of dense neighbourhoods: n_interp = 0)
    d = np.hypot(*(e1["verts"][e1["edges"][:, 0], :2] - e1["verts"][e1["edges"][:, 1], :2]).T)
    assert d.max() < 0.5 + 1e-9
    rv = r.revalidate()
    assert rv["invalid_vertices"] == 0 and rv["valid_edges_after"] == len(new)
    p, c, _ = r.solve()
    assert p is not None and np.array_equal(p[0], s) and np.array_equal(p[-1], g)
    assert ctx.validate_states(p).all() and ctx.check_motions(p[:-1], p[1:]).all()
    s2, g2 = g, near(p[len(p) // 2, :2] + 0.2)     # a query inside the component the first one crossed
    r.set_query(s2, g2)
    p2, c2, _ = r.solve()
    assert p2 is not None and np.array_equal(p2[0], s2) and np.array_equal(p2[-1], g2)
    assert ctx.check_motions(p2[:-1], p2[1:]).all()
    r.close()
    ctx.close()


def _directional_cost(a, b, v_lon=0.5, v_lat=0.1, v_ang=0.5):
    """PathLengthObjective::motionCost with use_directional_cost (path_length_objective.cpp:26-56), a -> b."""
    def yaw(q):
        return float(np.float32(np.arctan2(2 * (q[3] * q[2] + q[0] * q[1]), 1 - 2 * (q[1] ** 2 + q[2] ** 2))))
    dx, dy = b[0] - a[0], b[1] - a[1]
    y1, y2 = yaw(a[3:]), yaw(b[3:])
    dd = abs(y1 - y2)
    dyaw = 2 * np.pi - dd if dd > np.pi else dd
    lon = np.cos(y1) * dx + np.sin(y1) * dy
    lat = -np.sin(y1) * dx + np.cos(y1) * dy
    return max(abs(lon) / v_lon, abs(lat) / v_lat, abs(dyaw) / v_ang)


@pytest.mark.gpu
def test_reference_order_constructions_weigh_edges_in_the_direction_they_were_added():
    """The reference computes an edge's weight once, in the direction the edge was ADDED to its undirected graph
    (opt_->motionCost(m, n): new vertex -> neighbour, lazy_prm_star_min_update.cpp:436; source -> target in updateEdges,
    prm_motion_cost.cpp:33-44: m -> first chain vertex -> ... -> neighbour).  With the directional objective that is
    visible: weights and path cost of construction 1 and 2 equal the oracle's, whose _add_edge(a, b) is called in the
    reference's order."""
    import os
    import sys
    sys.path.insert(0, os.path.join(common.ROOT, "oracle"))
    import prm_incremental as PI
    from art_planner_amd.context import Context
    from art_planner_amd.roadmap import Roadmap
    from synthetic import make_map
    rob = O.robot("yaml")
    gm = make_map(160, 0.04, seed=1234)
    om = O.OracleMap(gm)
    ctx = Context(0, "yaml")
    ctx.upload_map(gm)
    se3 = ctx.sample_states(42, 0, 1 << 15)
    lab = om.states_valid(rob, se3)
    acc = se3[lab != 0]
    near = lambda xy: acc[np.argmin(np.hypot(acc[:, 0] - xy[0], acc[:, 1] - xy[1]))]
    s, g = near((gm.pos_x - 2.4, gm.pos_y - 2.4)), near((gm.pos_x + 2.4, gm.pos_y + 2.4))
    # construction 2: every edge is (new -> predecessor) = (larger id -> smaller id)
    ref = PI.lazy_prm_star_min_update(om, rob, acc, s, g, 800, cost_fn=_directional_cost)
    rm = Roadmap(ctx, s, g, n_milestones=800, seed=42, construction=2, objective=1)
    ex = rm.export()
    w_ref = np.array([ref["graph"].edges.get((int(u), int(v)), np.nan) for u, v in ex["edges"]])
    p, c, removed = rm.solve()
    left = rm.export()["edge_removed"] == 0
    assert np.isfinite(w_ref[left]).all() and np.allclose(ex["edge_cost"][left], w_ref[left], rtol=1e-9, atol=1e-12)
    V = ex["verts"]
    asym = [abs(_directional_cost(V[v], V[u]) - _directional_cost(V[u], V[v])) for u, v in ex["edges"][:2000]]
    assert max(asym) > 0.1                                          # the direction matters for this objective
    assert removed == ref["lazy_removals"] and (p is None) == (ref["path"] is None)
    if p is not None:
        assert abs(c - ref["path_cost"]) < 1e-9 * c and np.allclose(p, ref["path"], atol=1e-12)
    rm.close()
    # construction 1
    ref = PI.build_and_solve(om, rob, O.interpolate, acc, s, g, max_n_vertices=2500, max_n_edges=50000,
                             cost_fn=_directional_cost)
    G = ref["graph"]
    rm = Roadmap(ctx, s, g, n_milestones=2500, max_n_edges=50000, seed=42, construction=1, objective=1)
    ex = rm.export()
    ms = np.flatnonzero(np.array(G.is_milestone))
    vs, vg = int(ms[-2]), int(ms[-1])
    to_mine = np.empty(G.nv, np.int64)
    nxt = 2
    for o in range(G.nv):
        if o == vs:
            to_mine[o] = 0
        elif o == vg:
            to_mine[o] = 1
        else:
            to_mine[o] = nxt
            nxt += 1
    p, c, removed = rm.solve()
    ex2 = rm.export()
    ref_w = {tuple(sorted((int(to_mine[a]), int(to_mine[b])))): w for (a, b), w in G.edges.items()}
    left = ex2["edge_removed"] == 0
    assert {(int(u), int(v)) for u, v in ex2["edges"][left]} == set(ref_w.keys())
    w_ref = np.array([ref_w[(int(u), int(v))] for u, v in ex2["edges"][left]])
    assert np.allclose(ex2["edge_cost"][left], w_ref, rtol=1e-9, atol=1e-12)
    assert removed == ref["lazy_removals"] and (p is None) == (ref["path"] is None)
    if p is not None:
        assert abs(c - ref["path_cost"]) < 1e-9 * c
    rm.close()
    ctx.close()


@pytest.mark.gpu
def test_construction_1_reweights_like_sample_graph():
    """sampleGraph's in-build re-weighting (prm_motion_cost.cpp:190-193) in the reference's insertion order: the check
    `num_vertices / R > n_proc` counts GRAPH vertices (chain vertices included) after every milestone, and the sample
    stream continues right behind the sample that gave the milestone.  Until the first re-weighting the graph is the
    fixed-distribution graph, vertex for vertex; afterwards the milestones differ."""
    from art_planner_amd.roadmap import Roadmap
    gm, ctx, pm, s, g = _preprocessed_ctx()
    R, budget = 300, 1500
    fixed = Roadmap(ctx, s, g, n_milestones=budget, seed=3, construction=1)
    ef, sf_ = fixed.export(), fixed.stats()
    fixed.close()
    rw = Roadmap(ctx, s, g, n_milestones=budget, seed=3, construction=1, recompute_density_after_n_samples=R,
                 density_map=pm)
    er, sr = rw.export(), rw.stats()
    # one re-weighting per milestone at most, every time the vertex count passed another multiple of R
    assert 1 <= sr["reweightings"] <= (sr["vertices"] - 2) // R and sf_["reweightings"] == 0
    # vertices 2 .. are in insertion order: identical until the milestone that triggered the first re-weighting
    n_same = 0
    while n_same < min(len(ef["verts"]), len(er["verts"])) - 2 and np.array_equal(ef["verts"][2 + n_same], er["verts"][2 + n_same]):
        n_same += 1
    assert R <= n_same < R + 200 and n_same < sr["vertices"] - 2, n_same
    assert sr["samples_drawn"] != sf_["samples_drawn"]
    p, c, _ = rw.solve()
    assert p is None or (ctx.validate_states(p).all() and ctx.check_motions(p[:-1], p[1:]).all())
    rw.close()
    pm.reweight_dev(None)
    pm.close()
    ctx.close()


@pytest.mark.gpu
def test_lazy_removals_survive_a_grow_and_invalid_vertices_are_no_query_neighbours():
    """ADVICE r3.  (1) construction 1: the reference removes an edge its lazy path check rejected from g_ for good
    (prm_motion_cost.cpp:652-660) -- after solve -> grow -> solve the edges removed by the first solve are still marked
    removed and the second solve does not pay for them again.  (2) after a map update invalidated vertices, a new query
    is connected to valid vertices only (the reference's invalid vertices are not in nn_)."""
    from art_planner_amd.context import Context
    from art_planner_amd.roadmap import Roadmap
    from synthetic import make_map
    gm = make_map(160, 0.04, seed=1234)
    ctx = Context(0, "yaml")
    ctx.upload_map(gm)
    se3 = ctx.sample_states(42, 0, 1 << 15)
    acc = se3[ctx.validate_states(se3) != 0]
    near = lambda xy: acc[np.argmin(np.hypot(acc[:, 0] - xy[0], acc[:, 1] - xy[1]))]
    s, g = near((gm.pos_x - 2.4, gm.pos_y - 2.4)), near((gm.pos_x + 2.4, gm.pos_y + 2.4))
    r = Roadmap(ctx, s, g, n_milestones=10000, max_n_edges=50000, seed=42, construction=1)
    p0, c0, rep0 = r.solve()
    e0 = r.export()
    rem0 = {(int(u), int(v)) for (u, v), x in zip(e0["edges"], e0["edge_removed"]) if x}
    assert p0 is not None and rep0 >= 1 and len(rem0) == rep0
    r.grow(300)
    e1 = r.export()
    rem1 = {(int(u), int(v)) for (u, v), x in zip(e1["edges"], e1["edge_removed"]) if x}
    assert rem0 <= rem1, "lazy removals must carry over a grow"
    p1, c1, rep1 = r.solve()
    assert p1 is not None and ctx.check_motions(p1[:-1], p1[1:]).all()
    e2 = r.export()
    rem2 = {(int(u), int(v)) for (u, v), x in zip(e2["edges"], e2["edge_removed"]) if x}
    assert rem0 <= rem2 and rep1 == len(rem2) - len(rem1)      # only NEW removals were paid for
    # (2) raise a block of terrain under part of the roadmap, revalidate, re-query next to it
    verts = e2["verts"]
    mid = verts[len(verts) // 2, :2]
    # grid_map index of a position: rows run against x, columns against y
    i0 = int((gm.pos_x + gm.len_x / 2 - mid[0]) / gm.res)
    j0 = int((gm.pos_y + gm.len_y / 2 - mid[1]) / gm.res)
    r0, c0_ = min(max(0, i0 - 20), gm.rows - 40), min(max(0, j0 - 20), gm.cols - 40)
    for slot, name in ((0, "elevation"), (1, "elevation_masked")):
        a = gm[name].copy(order="F")
        patch = a[r0:r0 + 40, c0_:c0_ + 40].copy()
        patch[::2, :] += np.float32(0.6)           # a washboard: nothing stands there any more
        ctx.update_layer_rect(slot, patch, r0, c0_)
    rv = r.revalidate()
    assert rv["invalid_vertices"] > 0
    vok = ctx.validate_states(r.export()["verts"])
    assert (vok == 0).sum() == rv["invalid_vertices"]
    good = r.export()["verts"][vok != 0]
    block_xy = verts[np.flatnonzero(vok == 0)[0], :2]
    cand = good[np.argsort(np.hypot(good[:, 0] - block_xy[0], good[:, 1] - block_xy[1]))]
    s2, g2 = cand[0], cand[5]                       # valid states right next to the invalidated region
    r.set_query(s2, g2)
    ex = r.export()
    for q in (0, 1):
        nb = ex["knn"][q]
        nb = nb[nb != 0xffffffff]
        assert len(nb) > 0 and (vok[nb[nb >= 2]] != 0).all(), "a query vertex was connected to an invalid vertex"
    r.close()
    ctx.close()


@pytest.mark.gpu
def test_path_segments_are_priced_with_max_query_edge_length(planning_setup):
    """Params::planner.prm_motion_cost.max_query_edge_length (params.h:54): MotionCostObjective::motionCost splits a motion
    into (unsigned)(lateral distance / it) + 1 cost queries (motion_cost_objective.cpp:41-77) -- the price of a path
    SEGMENT (shortcut candidates of the simplification), not of the graph's sub-edges.  With 0.25 instead of the default
    0.5 the simplified path's cost equals a DP over segments priced with rows rebuilt in numpy at 0.25; the graph's
    edge costs do not change."""
    import os
    import sys
    sys.path.insert(0, os.path.join(common.ROOT, "oracle"))
    sys.path.insert(0, os.path.join(common.ROOT, "tools"))
    import motion_cost_oracle as mo
    import convert_weights
    from art_planner_amd.roadmap import Roadmap
    gm, ctx, start, goal = planning_setup
    ctx.cost_load_weights(convert_weights.to_blob(mo.random_params(0)))
    elv = np.ascontiguousarray(gm["elevation"][::-1, ::-1]).astype(np.float32)
    ctx.cost_update_map(elv, gm.res, gm.len_x, gm.len_y, gm.pos_x, gm.pos_y)
    w, thr = (0.25, 1.0, 5.0), 2.0   # threshold above every risk: all segments feasible, the DP is about the sums

    def yaw(q):
        return np.float32(np.arctan2(2 * (q[3] * q[2] + q[0] * q[1]), 1 - 2 * (q[1] ** 2 + q[2] ** 2)))

    def segment_cost(a, b, step):
        ni = int(np.hypot(b[0] - a[0], b[1] - a[1]) / step)
        pts = [a] + [O.interpolate(a, b, s / (ni + 1)) for s in range(1, ni + 1)] + [b]
        rows = [[s1[0], s1[1], yaw(s1[3:]), s0[0], s0[1], yaw(s0[3:])] for s0, s1 in zip(pts[:-1], pts[1:])]
        r = ctx.cost_query(np.array(rows, np.float32)).astype(np.float64)
        return float((r[:, 0] * np.float32(w[0]) + r[:, 1] * np.float32(w[1]) + r[:, 2] * np.float32(w[2])).sum())

    costs = {}
    for mq in (0.5, 0.25):
        rm = Roadmap(ctx, start, goal, n_milestones=4000, seed=11, objective=2, cost_weights=w, risk_threshold=thr,
                     max_query_edge_length=mq)
        path, cost, _ = rm.solve()
        assert path is not None and len(path) >= 4
        costs[mq] = (cost, rm.export()["edge_cost"].copy())
        p = path[:6]
        simp, scost = rm.simplify(p)
        n = len(p)
        ii, jj = np.triu_indices(n, 1)
        ok = (ctx.check_edges_interp(p[ii], p[jj])[0] != 0) & (ctx.check_motions(p[ii], p[jj]) != 0)
        best = np.full(n, np.inf)
        best[0] = 0.0
        for a, b, o in zip(ii, jj, ok):
            if o and np.isfinite(best[a]):
                best[b] = min(best[b], best[a] + segment_cost(p[a], p[b], mq))
        assert np.isfinite(best[-1]) and abs(best[-1] - scost) <= 1e-5 * max(1.0, scost), (mq, best[-1], scost)
        rm.close()
    # the graph (and the plan found on it) is priced per sub-edge of the 0.5 m chain whatever max_query_edge_length is
    assert costs[0.5][0] == costs[0.25][0] and np.array_equal(costs[0.5][1], costs[0.25][1])
