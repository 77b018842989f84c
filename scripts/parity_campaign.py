"""One-off randomized label-parity campaign (GPU pipeline vs CPU oracle) over map families the unit tests
only sample: NaN-riddled, terraced, steps, tiny and huge robots, near-vertical walls.  Usage (GPU box):
python scripts/parity_campaign.py [states_per_case]"""
import copy
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import common  # noqa: E402
import oracle_py as O  # noqa: E402
from art_planner_amd.context import Context, make_params  # noqa: E402
from synthetic import make_map  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 150000
n_edges = int(sys.argv[2]) if len(sys.argv) > 2 else 0   # edges per case through the three edge entry points (0 = none)
rng = np.random.default_rng(2024)

_EDGE_JOB = {}


def _edge_chunk(lo_hi):
    """Oracle verdicts of one chunk of the current case's edges (forked worker: the case lives in _EDGE_JOB)."""
    lo, hi = lo_hi
    om, rob, a, b = _EDGE_JOB["om"], _EDGE_JOB["rob"], _EDGE_JOB["a"][lo:hi], _EDGE_JOB["b"][lo:hi]
    ok, t, st = om.check_motions_last_valid(rob, a, b)
    ok0, _ = om.check_motions(rob, a, b)
    oki, ni = om.edges_interp_valid(rob, a, b)
    return ok, t, st, ok0, oki, ni


def edge_campaign(ctx, gm, rob, se3, labels):
    """n_edges edges between states of the case -- a state and a neighbour up to 2 m away with another attitude; a
    third of them start at accepted states -- through artp_check_motions, artp_check_motions_last_valid and
    artp_check_edges_interp against the CPU oracle (32 forked workers)."""
    import multiprocessing as mp
    m = n_edges
    src = rng.integers(0, len(se3), m)
    acc = np.flatnonzero(labels)
    if len(acc):
        src[::2] = acc[rng.integers(0, len(acc), len(src[::2]))]
    a = se3[src].copy()
    b = se3[rng.integers(0, len(se3), m)].copy()
    d = rng.uniform(-1.4, 1.4, (m, 2))
    b[:, 0], b[:, 1] = a[:, 0] + d[:, 0], a[:, 1] + d[:, 1]
    b[:, 2] = a[:, 2] + rng.normal(0, 0.03, m)
    b[::7] = a[::7]                                  # identical states (nd = 0)
    b[1::11, :3] = a[1::11, :3]                        # rotation only
    # (upload_map set the z bounds the oracle's z_extent uses: min / max finite elevation -+ reach.z / 2)
    g_ok = ctx.check_motions(a, b)
    g_ok2, g_t, g_st = ctx.check_motions_last_valid(a, b)
    g_oki, g_ni = ctx.check_edges_interp(a, b)
    _EDGE_JOB.update(om=O.OracleMap(gm), rob=rob, a=a, b=b)
    chunks = [(i, min(i + 250, m)) for i in range(0, m, 250)]
    with mp.get_context("fork").Pool(32) as pool:
        parts = pool.map(_edge_chunk, chunks)
    ok, t, st, ok0, oki, ni = (np.concatenate([p_[k] for p_ in parts]) for k in range(6))
    bad = int((g_ok != ok0).sum()) + int((g_ok2 != ok).sum()) + int((g_t != t).sum()) + int((g_oki != oki).sum()) + \
        int((g_ni != ni).sum()) + int((np.abs(g_st - st).max(axis=1) > 1e-12).sum())
    # the latency form (<= 64 edges per host call: check_motions_few_kernel) on a slice of the same edges, chunks of 1 .. 64:
    # the same verdicts, lastValid pairs (states bit-equal to the batch pipeline's) and interpolation counts
    mf = min(m, 2048)
    i, k, bad_few = 0, 1, 0
    while i < mf:
        j = min(i + k, mf)
        f_ok = ctx.check_motions(a[i:j], b[i:j])
        f_ok2, f_t, f_st = ctx.check_motions_last_valid(a[i:j], b[i:j])
        f_oki, f_ni = ctx.check_edges_interp(a[i:j], b[i:j])
        bad_few += int((f_ok != ok0[i:j]).sum()) + int((f_ok2 != ok[i:j]).sum()) + int((f_t != t[i:j]).sum()) + \
            int((f_oki != oki[i:j]).sum()) + int((f_ni != ni[i:j]).sum()) + \
            int((~((f_st == g_st[i:j]) | (np.isnan(f_st) & np.isnan(g_st[i:j])))).any(axis=1).sum())   # (nd = 0 on an invalid
        #                                                                    state: t = -inf, the interpolated state is NaN in both)
        i, k = j, k % 64 + 1
    _EDGE_JOB["bad_few"] = (bad_few, mf)
    # ... and through the resident pool (artp_set_persistent_latency: calls of one and two edges, no launch)
    ctx.set_persistent_latency(True)
    i, k, bad_pool = 0, 1, 0
    while i < mf:
        j = min(i + k, mf)
        f_ok = ctx.check_motions(a[i:j], b[i:j])
        f_ok2, f_t, f_st = ctx.check_motions_last_valid(a[i:j], b[i:j])
        f_oki, f_ni = ctx.check_edges_interp(a[i:j], b[i:j])
        bad_pool += int((f_ok != ok0[i:j]).sum()) + int((f_ok2 != ok[i:j]).sum()) + int((f_t != t[i:j]).sum()) + \
            int((f_oki != oki[i:j]).sum()) + int((f_ni != ni[i:j]).sum()) + \
            int((~((f_st == g_st[i:j]) | (np.isnan(f_st) & np.isnan(g_st[i:j])))).any(axis=1).sum())
        i, k = j, k % 2 + 1
    ctx.set_persistent_latency(False)
    _EDGE_JOB["bad_pool"] = (bad_pool, mf)
    return bad + bad_few + bad_pool, float(ok0.mean()), float(oki.mean())

base = make_map(240, 0.04, seed=31)


def variant(name):
    gm = copy.deepcopy(base)
    e, m = gm["elevation"].copy(), gm["elevation_masked"].copy()
    if name == "nan":
        holes = rng.random(e.shape) < 0.01
        e[holes] = np.nan
        m[rng.random(e.shape) < 0.01] = np.nan
    elif name == "terraced":
        e = (np.round(e / 0.05) * 0.05).astype(np.float32)
        m = np.where(np.isfinite(m), e, m).astype(np.float32)
    elif name == "steps":
        e = (e + 0.4 * ((np.arange(e.shape[0])[:, None] // 17) % 2)).astype(np.float32)
        m = np.where(np.isfinite(m), e, m).astype(np.float32)
    elif name == "inf":
        m[rng.random(e.shape) < 0.05] = np.inf
        e[rng.random(e.shape) < 0.002] = -np.inf
    if name == "far":  # a map far from the world origin: float cancellation in every pose -> field transform
        gm.pos_x, gm.pos_y = 512.3, -77.7
    gm.layers["elevation"] = np.asfortranarray(e)
    gm.layers["elevation_masked"] = np.asfortranarray(m)
    return gm


cases = [("plain", "yaml"), ("nan", "yaml"), ("terraced", "yaml"), ("steps", "defaults"), ("inf", "yaml"),
         ("plain", "tiny"), ("terraced", "defaults"), ("far", "yaml"), ("plain", "huge"), ("nan", "huge")]
bad_total = 0
lines = []
for mapname, robot in cases:
    gm = variant(mapname)
    if robot in ("tiny", "huge"):
        prm = make_params("yaml")
        if robot == "tiny":
            prm.torso_length, prm.torso_width, prm.torso_height = 0.3, 0.2, 0.1
            prm.feet_off_x, prm.feet_off_y = 0.1, 0.06
            prm.reach_x = prm.reach_y = prm.reach_z = 0.06
        else:  # torso windows of ~55 samples, foot windows of ~20: the largest table levels, three-block covers
            prm.torso_length, prm.torso_width, prm.torso_height = 1.9, 1.0, 0.3
            prm.feet_off_x, prm.feet_off_y, prm.feet_off_z = 0.7, 0.45, -0.6
            prm.reach_x, prm.reach_y, prm.reach_z = 0.6, 0.35, 0.25
        ctx = Context(0, prm)
        rob = O.Robot()
        for f, _ in O.Robot._fields_:
            if hasattr(prm, f):
                setattr(rob, f, getattr(prm, f))
    else:
        ctx = Context(0, robot)
        rob = O.robot(robot)
    ctx.upload_map(gm, sampler=False)
    se3 = common.random_states(gm, n, rng, z_off=(0.0, 0.06), tilt=0.25, spread=0.53)
    t0 = time.time()
    vg = ctx.validate_states(se3)
    vo = O.OracleMap(gm).states_valid(rob, se3)
    bad = int((vg != vo).sum())
    bad_total += bad
    # the latency path (<= 16 states per call, validate_few_kernel) on a slice of the same states
    few = np.concatenate([ctx.validate_states(se3[i:i + 16]) for i in range(0, 4096, 16)])
    bad_few = int((few != vo[:4096]).sum())
    bad_total += bad_few
    edge_txt = ""
    if n_edges:
        bad_e, vf, vi = edge_campaign(ctx, gm, rob, se3, vo)
        bad_total += bad_e
        bf, mf = _EDGE_JOB.get("bad_few", (0, 0))
        bp, mp_ = _EDGE_JOB.get("bad_pool", (0, 0))
        edge_txt = (f" | {n_edges} edges x (checkMotion, lastValid pair, 0.5 m rule + n_interp): mismatches={bad_e} "
                    f"(valid {vf:.3f} / {vi:.3f}); of which the latency form on {mf} of them in calls of 1..64 edges: {bf}, "
                    f"the resident pool on {mp_} of them in calls of 1..2 edges: {bp}")
    lines.append(f"{mapname:9s} {robot:8s} states={n} valid={vg.mean():.3f} mismatches={bad} latency-path mismatches={bad_few}/4096 "
                 f"counters={ctx.pipeline_counters()}{edge_txt} ({time.time() - t0:.1f}s)")
    print(lines[-1], flush=True)
    ctx.close()
lines.append(f"TOTAL MISMATCHES {bad_total} over {n * len(cases)} states (batch pipeline) + {4096 * len(cases)} (latency path)" +
             (f" + {n_edges * len(cases)} edges through each of the three edge entry points (+ {min(n_edges, 2048) * len(cases)} "
              "of them again through the one-launch latency form and through the resident pool)" if n_edges else ""))
print(lines[-1])
out = os.path.join(ROOT, "gpurun_out", "parity_campaign.txt")
os.makedirs(os.path.dirname(out), exist_ok=True)
with open(out, "w") as f:
    f.write("scripts/parity_campaign.py: GPU labels (C ABI) vs CPU oracle, random states over map / robot families\n")
    f.write("\n".join(lines) + "\n")
sys.exit(1 if bad_total else 0)
