"""End-to-end replanning loop on one MI355X, the way art_planner's PlannerRos drives its planner at map
rate (art_planner_ros/src/planner_ros.cpp: map callback -> Planner::setMap -> plan): every cycle a new raw
elevation map arrives, is preprocessed and installed on the device, the kept roadmap is re-validated,
the robot's new pose becomes the start, and a plan comes back.

    python examples/replan_loop.py [cycles]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))  # synthetic input maps live with the tests
from art_planner_amd.context import Context  # noqa: E402
from art_planner_amd.roadmap import Roadmap  # noqa: E402
from synthetic import make_map  # noqa: E402


def main(cycles=10, verbose=True):
    gm = make_map(400, 0.04, seed=1234)
    ctx = Context(0, "yaml")
    elev = gm["elevation"].copy()
    trav = gm["traversability"]
    prev = ctx.preprocess_map(elev, gm.len_x, gm.len_y, gm.pos_x, gm.pos_y, traversability=trav)
    prev.install()
    probe = ctx.sample_states(1, 0, 1 << 15)
    ok = probe[ctx.validate_states(probe) != 0]
    start = ok[np.argmin(np.hypot(ok[:, 0] + 6.0, ok[:, 1] + 6.0))]
    goal = ok[np.argmin(np.hypot(ok[:, 0] - 6.0, ok[:, 1] - 6.0))]
    rm = Roadmap(ctx, start, goal, n_milestones=10000, seed=7)
    path, cost, _ = rm.solve()
    assert path is not None
    rng = np.random.default_rng(0)
    stats = []
    for cyc in range(cycles):
        t0 = time.perf_counter()
        # a new map: a bump appears somewhere near the plan
        k = rng.integers(2, len(path) - 2)
        ix = int((gm.pos_x + 0.5 * gm.len_x - path[k, 0]) / gm.res) + rng.integers(-15, 15)
        iy = int((gm.pos_y + 0.5 * gm.len_y - path[k, 1]) / gm.res) + rng.integers(-15, 15)
        ix, iy = int(np.clip(ix, 0, gm.rows - 10)), int(np.clip(iy, 0, gm.cols - 10))
        elev[ix:ix + 10, iy:iy + 10] += np.float32(0.5)
        new = ctx.preprocess_map(elev, gm.len_x, gm.len_y, gm.pos_x, gm.pos_y, traversability=trav, vertices=path)
        _, rect, n_upd = new.change_from(prev, 0.05)
        new.install()
        t1 = time.perf_counter()
        info = rm.revalidate()
        # the robot moved one state along its plan; if that state died with the map change, stay put
        cand = path[min(1, len(path) - 2)]
        cur = cand if ctx.validate_states(cand[None])[0] else path[0]
        if not (ctx.validate_states(cur[None])[0] and ctx.validate_states(goal[None])[0]):
            print("start or goal buried by the map change; stopping")
            break
        rm.set_query(cur, goal)
        t2 = time.perf_counter()
        path2, cost2, lazy = rm.solve()
        if path2 is None:
            print(f"cycle {cyc}: goal unreachable on the kept roadmap -> resample")
            rm.close()
            rm = Roadmap(ctx, cur, goal, n_milestones=10000, seed=100 + cyc)
            path2, cost2, lazy = rm.solve()
            if path2 is None:
                break
        simp, scost = rm.simplify(path2)
        t3 = time.perf_counter()
        prev.close()
        prev, path = new, path2
        stats.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
        if verbose:
            print(f"cycle {cyc}: map {stats[-1][0]:.2f} ms (updated cells {n_upd}, rect {rect[2]}x{rect[3]}), "
                  f"roadmap {stats[-1][1]:.2f} ms ({info['invalid_vertices']} vertices invalid), "
                  f"plan+simplify {stats[-1][2]:.2f} ms: {len(path2)} -> {len(simp)} states, cost {cost2:.2f} -> {scost:.2f} s")
    rm.close()
    prev.close()
    ctx.close()
    return np.array(stats)


if __name__ == "__main__":
    s = main(int(sys.argv[1]) if len(sys.argv) > 1 else 10)
    if len(s):
        print("median per cycle: map %.2f ms, roadmap upkeep %.2f ms, plan %.2f ms" % tuple(np.median(s, axis=0)))
