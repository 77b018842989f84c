"""One-off differential campaign for artp_roadmap_params::construction = 1 / 2 against oracle/prm_incremental.py (the
reference planners' graph constructions restated literally): several maps, sample-stream seeds, budgets and both
objectives.  Reports vertices / edges / removals / path cost and whether the edge SETS are identical.
Usage (GPU box): python scripts/roadmap_campaign.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle_py as O  # noqa: E402
import prm_incremental as PI  # noqa: E402
from art_planner_amd.context import Context  # noqa: E402
from art_planner_amd.roadmap import Roadmap  # noqa: E402
from synthetic import make_map  # noqa: E402
from test_roadmap import _directional_cost  # noqa: E402

rob = O.robot("yaml")
bad = 0
for map_name, gm in (("perlin160/1234", make_map(160, 0.04, seed=1234)), ("perlin200/77", make_map(200, 0.04, seed=77)),
                     ("flat100", make_map(100, 0.1, flat=True))):
    om = O.OracleMap(gm)
    ctx = Context(0, "yaml")
    ctx.upload_map(gm)
    for seed in (42, 7, 2024):
        se3 = ctx.sample_states(seed, 0, 1 << 15)
        lab = om.states_valid(rob, se3)
        assert np.array_equal(ctx.validate_states(se3), lab)
        acc = se3[lab != 0]
        near = lambda xy: acc[np.argmin(np.hypot(acc[:, 0] - xy[0], acc[:, 1] - xy[1]))]
        q = 0.3 * gm.len_x
        s, g = near((gm.pos_x - q, gm.pos_y - q)), near((gm.pos_x + q, gm.pos_y + q))
        for objective, cost_fn in ((0, None), (1, _directional_cost)):
            # construction 2
            n_ms = min(1200, len(acc))
            t0 = time.perf_counter()
            ref = PI.lazy_prm_star_min_update(om, rob, acc, s, g, n_ms, cost_fn=cost_fn, max_replans=100000)
            t1 = time.perf_counter()
            rm = Roadmap(ctx, s, g, n_milestones=n_ms, seed=seed, construction=2, objective=objective, max_replans=100000)
            p, c, removed = rm.solve()
            t2 = time.perf_counter()
            ex = rm.export()
            left = {(int(u), int(v)) for (u, v), r in zip(ex["edges"], ex["edge_removed"]) if not r}
            same = left == set(ref["graph"].edges.keys()) and removed == ref["lazy_removals"] and \
                (p is None) == (ref["path"] is None) and (p is None or abs(c - ref["path_cost"]) < 1e-9 * max(1.0, c))
            bad += 0 if same else 1
            print(f"{map_name:15s} seed {seed:5d} obj {objective} construction 2: vertices {ex['verts'].shape[0]:5d} edges {len(ex['edges']):6d} "
                  f"removed {removed:3d} cost {c:9.4f} | oracle edges {ref['edges']:6d} removed {ref['lazy_removals']:3d} cost {ref['path_cost']:9.4f} "
                  f"| {'IDENTICAL' if same else 'DIFFERENT'} (oracle {t1 - t0:.1f} s, device {t2 - t1:.3f} s)", flush=True)
            rm.close()
            # construction 1
            for bv in (1500, 4000):
                t0 = time.perf_counter()
                ref = PI.build_and_solve(om, rob, O.interpolate, acc, s, g, max_n_vertices=bv, max_n_edges=50000, cost_fn=cost_fn, max_replans=100000)
                t1 = time.perf_counter()
                rm = Roadmap(ctx, s, g, n_milestones=bv, max_n_edges=50000, seed=seed, construction=1, objective=objective, max_replans=100000)
                p, c, removed = rm.solve()
                t2 = time.perf_counter()
                ex = rm.export()
                G = ref["graph"]
                ms = np.flatnonzero(np.array(G.is_milestone))
                vs, vg = int(ms[-2]), int(ms[-1])
                to_mine, nxt = np.empty(G.nv, np.int64), 2
                for o in range(G.nv):
                    if o == vs:
                        to_mine[o] = 0
                    elif o == vg:
                        to_mine[o] = 1
                    else:
                        to_mine[o] = nxt
                        nxt += 1
                left = {(int(u), int(v)) for (u, v), r in zip(ex["edges"], ex["edge_removed"]) if not r}
                ref_left = {tuple(sorted((int(to_mine[a]), int(to_mine[b])))) for (a, b) in G.edges.keys()}
                same = ex["verts"].shape[0] == G.nv and left == ref_left and removed == ref["lazy_removals"] and \
                    (p is None) == (ref["path"] is None) and (p is None or abs(c - ref["path_cost"]) < 1e-9 * max(1.0, c))
                bad += 0 if same else 1
                print(f"{map_name:15s} seed {seed:5d} obj {objective} construction 1 budget {bv:5d}: vertices {ex['verts'].shape[0]:5d} "
                      f"(chain {ref['chain_vertices']:5d}) edges {len(ex['edges']):6d} removed {removed:3d} cost {c:9.4f} | oracle vertices {G.nv:5d} "
                      f"edges {ref['edges']:6d} cost {ref['path_cost']:9.4f} | {'IDENTICAL' if same else 'DIFFERENT'} "
                      f"(oracle {t1 - t0:.1f} s, device {t2 - t1:.3f} s)", flush=True)
                rm.close()
    ctx.close()
print("GRAPHS THAT DIFFER FROM THE ORACLE'S:", bad)
