"""Tuning aid: one no-path case of scripts/roadmap_campaign.py (perlin200/77, directional objective, construction 2)
under the solver's environment switches: which of them changes the number of lazy removals."""
import os
# the $ARTP_LAZY_* / $ARTP_SOLVE_ASTAR switches are read by the variants build only: run with
#   ARTP_LIB=art_planner_amd/csrc/libartp_variants.so python scripts/lazy_nopath_probe.py, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")): sys.path.insert(0, p)
    import numpy as np
    import oracle_py as O
    from art_planner_amd.context import Context
    from art_planner_amd.roadmap import Roadmap
    from synthetic import make_map
    gm = make_map(200, 0.04, seed=77)
    om = O.OracleMap(gm); rob = O.robot("yaml")
    ctx = Context(0, "yaml"); ctx.upload_map(gm)
    for seed in (42, 7):
        se3 = ctx.sample_states(seed, 0, 1 << 15)
        lab = ctx.validate_states(se3)
        acc = se3[lab != 0]
        near = lambda xy: acc[np.argmin(np.hypot(acc[:, 0] - xy[0], acc[:, 1] - xy[1]))]
        q = 0.3 * gm.len_x
        s, g = near((gm.pos_x - q, gm.pos_y - q)), near((gm.pos_x + q, gm.pos_y + q))
        for objective in (0, 1):
            rm = Roadmap(ctx, s, g, n_milestones=min(1200, len(acc)), seed=seed, construction=2, objective=objective, max_replans=100000)
            p, c, removed = rm.solve()
            ex = rm.export()
            import hashlib
            h = hashlib.sha1(np.asarray(ex["edge_removed"], np.uint8).tobytes()).hexdigest()[:12]
            print(f"  seed {seed} obj {objective}: removed {removed} cost {c} removed-set {h}")
            rm.close()
else:
    for env in ({}, {"ARTP_LAZY_INFORMED": "0"}, {"ARTP_LAZY_ROOT": "0"}, {"ARTP_LAZY_ROOT": "1"}, {"ARTP_LAZY_ROOT": "0", "ARTP_LAZY_INFORMED": "0"}, {"ARTP_SOLVE_ASTAR": "1"}):
        print(env, flush=True)
        subprocess.call([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, **env))
