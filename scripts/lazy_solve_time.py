"""Solve time of the roadmaps the host searches (bench map, bench query): the batched default and the reference planners'
own constructions, with the shortest-path tree + verdict table (default) and with round 3's A* per round (ARTP_SOLVE_ASTAR=1
in a child process); paths and removal counts must agree."""
import json, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from art_planner_amd.context import Context
from art_planner_amd.roadmap import Roadmap
from synthetic import map_from_device, raw_map

def run():
    ctx = Context(0, "yaml"); gm = map_from_device(ctx, raw_map(400, 0.04, seed=1234))
    probe = ctx.sample_states(42, 9_000_000, 1 << 15); okp = probe[ctx.validate_states(probe) != 0]
    s = okp[np.argmin(np.hypot(okp[:, 0] - (gm.pos_x - 0.4 * gm.len_x), okp[:, 1] - (gm.pos_y - 0.4 * gm.len_y)))]
    g = okp[np.argmin(np.hypot(okp[:, 0] - (gm.pos_x + 0.4 * gm.len_x), okp[:, 1] - (gm.pos_y + 0.4 * gm.len_y)))]
    out = {}
    for name, kw in (("batched_10000", dict(n_milestones=10000)),
                     ("prm_motion_cost_order", dict(n_milestones=10000, max_n_edges=50000, construction=1)),
                     ("lazy_prm_star_order_10000", dict(n_milestones=10000, construction=2)),
                     ("lazy_prm_star_order_20000", dict(n_milestones=20000, construction=2))):
        Roadmap(ctx, s, g, n_milestones=500, seed=42).close()
        rm = Roadmap(ctx, s, g, seed=42, **kw)
        t0 = time.perf_counter(); p, c, r = rm.solve(); t1 = time.perf_counter()
        p2, c2, r2 = rm.solve(); t2 = time.perf_counter()
        e = rm.export()
        out[name] = {"solve_ms": (t1 - t0) * 1e3, "again_ms": (t2 - t1) * 1e3, "cost": c, "cost_again": c2, "removals": r,
                     "removals_again": r2, "path_states": None if p is None else len(p), "edges": int(len(e["edges"])),
                     "removed_edges": sorted(map(int, np.flatnonzero(e["edge_removed"]))),
                     "path": None if p is None else [list(map(float, x)) for x in p]}
        rm.close()
    ctx.close()
    return out

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        print("RESULT " + json.dumps(run()))
        sys.exit(0)
    new = run()
    # $ARTP_SOLVE_ASTAR is read by the variants build only
    vlib = os.path.join(ROOT, "art_planner_amd", "csrc", "libartp_variants.so")
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, ARTP_SOLVE_ASTAR="1", ARTP_LIB=vlib),
                       capture_output=True, text=True)
    old = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    for k in new:
        a, b = new[k], old[k]
        same = a["removed_edges"] == b["removed_edges"] and a["path"] == b["path"] and a["removals"] == b["removals"]
        print(f"{k:28s} edges {a['edges']:7d} removals {a['removals']:4d}  tree+table {a['solve_ms']:8.2f} ms (again {a['again_ms']:6.2f})"
              f"   A* per round {b['solve_ms']:8.2f} ms (again {b['again_ms']:6.2f})   same path / removals: {same}")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    for d in (new, old):
        for v in d.values():
            v.pop("path"); v.pop("removed_edges")
    json.dump({"tree_and_verdict_table": new, "astar_per_round_r03": old}, open(os.path.join(ROOT, "gpurun_out", "lazy_solve_time.json"), "w"), indent=1)
