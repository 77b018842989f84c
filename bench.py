#!/usr/bin/env python3
"""bench.py -- validated states/s (+ edges/s) of the sampling + validity hot path on MI355X.

Workload (BASELINE.json configs[1], "C2"): lazy_prm_star_min_update front end, 400x400 @ 0.04 m
Perlin terrain + obstacles (seed 1234), YAML robot, one batch validity checker per GPU.
A "step" = one pass of the hot path over one batch of S = 2^22 candidate states per GPU:
    SE3FromSE2Sampler::sampleUniform (batched, counter-based RNG)  ->  StateValidityChecker::isValid
with all inputs (map layers) resident in HBM before the timed region.  With N > 1 GPUs every rank
owns a disjoint sample-index range (weak scaling) and the accepted states are compacted and
all-gathered over RCCL on a side stream (the planners need every accepted state on every rank).

Prints ONE JSON line on rank 0 (see the repo prompt's bench contract) incl. `roofline` and
`cpu_baseline`.
"""
import argparse
import hashlib
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def cpu_baseline(gm, states, target_s=12.0):
    """The CPU oracle ("port": bit-identical restatement of the reference OMPL+ODE validity path,
    faithful algorithmic structure) timed on this box's host cores on a bounded sample of the SAME
    states the GPU validated.  Only the checker is used here -- never the product path."""
    import oracle_py as O
    rob = O.robot("yaml")
    om = O.OracleMap(gm)
    n1 = min(16384, len(states))
    t0 = time.perf_counter()
    v1 = om.states_valid(rob, states[:n1])
    t1 = time.perf_counter()
    r1 = n1 / (t1 - t0)
    cores = os.cpu_count() or 1
    n = int(min(len(states), max(n1, r1 * target_s * min(cores, 8) / 2)))
    chunks = np.array_split(np.arange(n), cores)
    out = np.empty(n, np.uint8)
    maps = [O.OracleMap(gm) for _ in range(cores)]  # one private checker pair per thread

    def work(k):
        idx = chunks[k]
        if len(idx):
            out[idx] = maps[k].states_valid(rob, states[idx])

    th = [threading.Thread(target=work, args=(k,)) for k in range(cores)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "states/s", "cores": cores, "kind": "port",
            "sample": f"first {n} sampler states of batch 0 (seed 42), oracle/artp_oracle.c faithful mode, "
                      f"{cores} threads with private checkers; single thread: {r1:.0f} states/s on {n1}",
            "single_core_value": r1}, out, v1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=1 << 22, help="candidate states per GPU per step")
    ap.add_argument("--edges", type=int, default=1 << 18)
    ap.add_argument("--map", type=int, default=400)
    ap.add_argument("--res", type=float, default=0.04)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--skip-extras", action="store_true", help="no edge / motion-cost measurements (profiling)")
    ap.add_argument("--materialise", type=int, default=1 << 16,
                    help="N>1: accepted states of EVERY rank re-materialised on every rank per step, per rank block "
                         "(-1 = all of them, 0 = none; the gathered index lists are always complete)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed and run the all-gather path even with one rank (self-test)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    N = args.gpus
    assert world == N or (N == 1 and world == 1), f"--gpus {N} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product has no CPU path)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if N > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from art_planner_amd.context import Context
    from art_planner_amd.synthetic import make_map

    gm = make_map(args.map, args.res, seed=1234)
    ctx = Context(local_rank, "yaml")
    ctx.upload_map(gm)
    # one explicit stream carries the kernels AND the HIP events that time them
    main_stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(main_stream)
    ctx.use_torch_stream()

    S, K, W, seed = args.batch, args.steps, args.warmup, 42
    se3 = torch.empty((S, 7), dtype=torch.float64, device=dev)
    valid = torch.empty(S, dtype=torch.uint8, device=dev)
    do_gather = (N > 1 or args.force_dist) and not args.no_gather
    comm = torch.cuda.Stream(device=dev) if do_gather else None

    from art_planner_amd.distributed import shard_first_index

    def first_index(step):
        return shard_first_index(step, rank, N, S)

    # ---- warmup (also sizes the fixed-capacity all-gather blocks) -----------------------------
    cap = 0
    compact = [None, None]
    counts = [None, None]
    gatherer = None
    gather_error = None
    for i in range(max(W, 1)):
        c = ctx.sample_and_validate_dev(seed, first_index(1000000 + i), S, se3, valid, count=True)
        cap = max(cap, c)
    torch.cuda.synchronize()
    if do_gather:
        from art_planner_amd.distributed import ValidIndexGatherer, agree_capacity
        cap = agree_capacity(cap, S, dev)
        idx_buf = [torch.zeros(S, dtype=torch.int32, device=dev) for _ in range(2)]
        counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(2)]
        gatherers = [ValidIndexGatherer(N, cap, dev), ValidIndexGatherer(N, cap, dev)]  # double-buffered
        gatherer = gatherers[0]
        # every rank's accepted states, re-materialised from the gathered indices: the first mat_cap per rank and
        # step (default 2^16 = 6.5x the reference's whole roadmap, max_n_vertices = 10^4); anything beyond is
        # one artp_sample_states_at_dev call away because the index lists are complete
        mat_cap = cap if args.materialise < 0 else min(cap, args.materialise)
        all_states = torch.empty((N, max(mat_cap, 1), 7), dtype=torch.float64, device=dev)
        done_ev = [torch.cuda.Event(), torch.cuda.Event()]
        for e in done_ev:
            e.record()
        try:  # trial exchange outside the timed region; a failing collective must not lose the whole run
            ctx.compact_valid_indices_dev(valid, idx_buf[0], counts[0])
            gatherers[0].gather(idx_buf[0], counts[0])
            torch.cuda.synchronize()
        except Exception as ex:  # pragma: no cover
            gather_error = repr(ex)
            do_gather = False

    def step(i):
        ctx.sample_and_validate_dev(seed, first_index(i), S, se3, valid)
        if do_gather:
            b = i & 1
            torch.cuda.current_stream().wait_event(done_ev[b])  # buffer b free again
            ctx.compact_valid_indices_dev(valid, idx_buf[b], counts[b])
            ready = torch.cuda.Event()
            ready.record()
            comm.wait_event(ready)
            with torch.cuda.stream(comm):
                gatherers[b].gather(idx_buf[b], counts[b])     # 4 B per accepted state over xGMI
                done_ev[b].record()
            # materialise the accepted states of every rank for the PREVIOUS step (its gather has had a
            # whole step to complete): a state is a pure function of (seed, index).  On the main stream:
            # the validity kernels are persistent grids with static striding, and a kernel that shares their
            # CUs from a side stream costs them more than it hides (measured: +0.36 ms for 0.15 ms of work).
            if i > 0 and mat_cap > 0:
                materialise(i - 1)

    def materialise(j):
        gb = gatherers[j & 1]
        torch.cuda.current_stream().wait_event(done_ev[j & 1])
        for r in range(N):
            ctx.sample_states_at_dev(seed, shard_first_index(j, r, N, S), gb.gathered[r], gb.counts[r:r + 1], mat_cap,
                                     all_states[r])

    # ---- timed region: exactly K steps, barrier + synchronize on both sides -----------------------
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        step(i)
    if do_gather and mat_cap > 0:
        materialise(K - 1)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        assert (not do_gather) or max(int(g_.counts.max().item()) for g_ in gatherers) <= cap, "all-gather block capacity exceeded"
    total_states = N * S * K
    value = total_states / dt

    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- everything below: rank 0, outside the timed region ------------------------------------
    # batch 0 again (deterministic) for the roofline, label hash, edges and CPU baseline
    ctx.sample_and_validate_dev(seed, 0, S, se3, valid)
    torch.cuda.synchronize()
    labels = valid.cpu().numpy()
    label_hash = hashlib.sha1(labels.tobytes()).hexdigest()[:16]
    valid_frac = float(labels.mean())

    # dominant kernel: validate_states_kernel; HIP events on the stream it is launched on
    alg_vertices = ctx.algorithmic_vertices_dev(se3)
    alg_bytes = 4 * alg_vertices + 29 * S  # 28 B pose in (7 f32) + 1 B label out per state
    reps = max(5, min(K, 20))
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ctx.validate_states_dev(se3, valid)
    torch.cuda.synchronize()
    ev0.record()
    for _ in range(reps):
        ctx.validate_states_dev(se3, valid)
    ev1.record()
    torch.cuda.synchronize()
    k_ms = ev0.elapsed_time(ev1) / reps
    pipeline_counts = ctx.pipeline_counters()
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9
    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc_path):
        try:
            traffic = json.load(open(pmc_path)).get("validate_states_kernel_hbm_bytes_per_launch")
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "kernel": "validate_states_kernel", "kernel_ms": k_ms,
                "kernel_launches": "artp_validate_states_dev = classify_states_kernel + feet_stream_kernel<4> + "
                                   "resolve_boxes_kernel<2,64,0> + 5 near-empty fallback launches (profiles/README.md)",
                "algorithmic_bytes_per_launch": alg_bytes,
                "algorithmic_bytes_per_state": alg_bytes / S,
                "validate_only_states_per_s": S / (k_ms * 1e-3),
                "note": "achieved = ALGORITHMIC bytes (what the reference's scan reads, SURVEY.md 8d) / kernel time; "
                        "frac > 1 means exact range / partner tables avoided reading them -- `traffic` is what moved"}

    # sampler alone
    ev0.record()
    for _ in range(reps):
        ctx.sample_states_dev(seed, 0, S, se3)
    ev1.record()
    torch.cuda.synchronize()
    sample_ms = ev0.elapsed_time(ev1) / reps
    ctx.sample_and_validate_dev(seed, 0, S, se3, valid)
    torch.cuda.synchronize()

    # edges: accepted state i paired with accepted state i+1 when their lateral distance < 2 m
    states = se3.cpu().numpy()
    acc = states[labels != 0]
    a, b = acc[:-1], acc[1:]
    near = np.hypot(a[:, 0] - b[:, 0], a[:, 1] - b[:, 1]) < 2.0
    a, b = a[near][:args.edges], b[near][:args.edges]
    E = 0 if args.skip_extras else len(a)
    edges = {}
    if E > 0:
        s1 = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        s2 = torch.from_numpy(np.ascontiguousarray(b)).to(dev)
        ev = torch.empty(E, dtype=torch.uint8, device=dev)
        for name, fn in (("check_motion", lambda: ctx.check_motions_dev(s1, s2, ev)),
                         ("interp_0p5m", lambda: ctx.check_edges_interp_dev(s1, s2, ev))):
            fn()
            torch.cuda.synchronize()
            ev0.record()
            for _ in range(3):
                fn()
            ev1.record()
            torch.cuda.synchronize()
            ms = ev0.elapsed_time(ev1) / 3
            edges[name] = {"edges": E, "edges_per_s": E / (ms * 1e-3), "ms": ms,
                           "valid_frac": float(ev.float().mean().item())}

    # ---- C3 extras: learned motion cost (seeded random weights: the trained ones are git-LFS stubs) ----
    motion_cost = None
    try:
        if args.skip_extras:
            raise RuntimeError("skipped (--skip-extras)")
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import convert_weights
        ctx.cost_load_weights(convert_weights.to_blob(convert_weights.random_params(0)))
        elv = np.ascontiguousarray(gm["elevation"][::-1, ::-1]).astype(np.float32)  # cost_query_server.py:66-74
        ctx.cost_update_map(elv, gm.res, gm.len_x, gm.len_y)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            ctx.cost_update_map(elv, gm.res, gm.len_x, gm.len_y)   # H2D of the map + CNN, synchronous
        cnn_ms = (time.perf_counter() - t0) / 5 * 1e3
        shp = []
        h = gm.rows
        for (k, cin, cout, pool) in ((3, 1, 24, 0), (3, 24, 24, 2), (3, 24, 48, 0), (3, 48, 48, 3), (3, 48, 48, 0),
                                     (15, 48, 48, 0)):
            h = h - k + 1
            shp.append(2.0 * k * k * cin * cout * h * h)
            if pool == 2:
                h //= 2
            elif pool == 3:
                h -= 2
        cnn_gflop = sum(shp) / 1e9
        Bq = 1 << 20
        if E > 0:
            reps_e = (Bq + E - 1) // E
            em = np.concatenate([np.concatenate([b[:, [0, 1]], np.arctan2(2 * (b[:, 6] * b[:, 5] + b[:, 3] * b[:, 4]),
                                 1 - 2 * (b[:, 4] ** 2 + b[:, 5] ** 2))[:, None],
                                 a[:, [0, 1]], np.arctan2(2 * (a[:, 6] * a[:, 5] + a[:, 3] * a[:, 4]),
                                 1 - 2 * (a[:, 4] ** 2 + a[:, 5] ** 2))[:, None]], 1)] * reps_e)[:Bq]
            edges_t = torch.from_numpy(np.ascontiguousarray(em, dtype=np.float32)).to(dev)
            cost_t = torch.empty((Bq, 3), dtype=torch.float32, device=dev)
            ctx.cost_query_dev(edges_t, cost_t)
            torch.cuda.synchronize()
            ev0.record()
            for _ in range(5):
                ctx.cost_query_dev(edges_t, cost_t)
            ev1.record()
            torch.cuda.synchronize()
            q_ms = ev0.elapsed_time(ev1) / 5
            motion_cost = {"cnn_ms_incl_h2d": cnn_ms, "cnn_gflop": cnn_gflop,
                           "cnn_tflops": cnn_gflop / cnn_ms, "cnn_frac_of_mfma_f16_peak": cnn_gflop / cnn_ms / 2500.0,
                           "cost_queries": Bq, "cost_queries_per_s": Bq / (q_ms * 1e-3), "cost_query_ms": q_ms,
                           "weights": "seeded random (tools/convert_weights.random_params(0))"}
    except Exception as ex:  # pragma: no cover
        motion_cost = {"error": repr(ex)}

    # ---- C5 extras: persistent HBM map with incremental updates, one replanning cycle -----------------
    c5 = None
    try:
        if args.skip_extras:
            raise RuntimeError("skipped (--skip-extras)")
        rng5 = np.random.default_rng(55)
        elev0 = gm["elevation"].copy()
        n5, cyc = 1 << 18, []
        se5 = torch.empty((n5, 7), dtype=torch.float64, device=dev)
        va5 = torch.empty(n5, dtype=torch.uint8, device=dev)
        for c_i in range(10):
            t0 = time.perf_counter()
            for _ in range(3):  # 3 rectangles ~ 5 % of the cells
                r0, c0 = int(rng5.integers(0, gm.rows - 52)), int(rng5.integers(0, gm.cols - 52))
                patch = (elev0[r0:r0 + 52, c0:c0 + 52] + np.float32(rng5.normal(0, 0.02))).astype(np.float32)
                ctx.update_layer_rect(0, patch, r0, c0)   # dirty tiles -> HBM + range-table refresh
            ctx.sample_and_validate_dev(seed, 7_000_000 + c_i * n5, n5, se5, va5)
            torch.cuda.synchronize()
            cyc.append((time.perf_counter() - t0) * 1e3)
        ctx.upload_map(gm)  # restore
        c5 = {"cycle_ms_median": float(np.median(cyc)), "cycle_ms_max": float(np.max(cyc)),
              "states_per_cycle": n5, "dirty_rects_per_cycle": 3, "budget_ms_at_10hz": 100.0}
    except Exception as ex:  # pragma: no cover
        c5 = {"error": repr(ex)}

    # ---- N1 extras: batched roadmap front end (sample -> k-NN -> edge rule -> costs -> A* + lazy check) -----
    roadmap = None
    try:
        if args.skip_extras:
            raise RuntimeError("skipped (--skip-extras)")
        from art_planner_amd.roadmap import Roadmap
        probe = ctx.sample_states(seed, 9_000_000, 1 << 15)
        okp = probe[ctx.validate_states(probe) != 0]
        s_state = okp[np.argmin(np.hypot(okp[:, 0] - (gm.pos_x - 0.4 * gm.len_x), okp[:, 1] - (gm.pos_y - 0.4 * gm.len_y)))]
        g_state = okp[np.argmin(np.hypot(okp[:, 0] - (gm.pos_x + 0.4 * gm.len_x), okp[:, 1] - (gm.pos_y + 0.4 * gm.len_y)))]
        roadmap = {}
        for n_m in (10_000, 100_000):  # 10 000 = Params::planner.prm_motion_cost.max_n_vertices
            Roadmap(ctx, s_state, g_state, n_milestones=1000, seed=seed).close()  # warm the allocators
            t0 = time.perf_counter()
            rm = Roadmap(ctx, s_state, g_state, n_milestones=n_m, seed=seed)
            t1 = time.perf_counter()
            path, cost, rep = rm.solve()
            t2 = time.perf_counter()
            st = rm.stats()
            t3 = time.perf_counter()
            rm.revalidate()          # the kept roadmap after a map update: every vertex and edge re-checked
            t4 = time.perf_counter()
            rm.solve()
            t5 = time.perf_counter()
            rm.close()
            roadmap[f"milestones_{n_m}"] = {
                "build_ms": (t1 - t0) * 1e3, "solve_ms": (t2 - t1) * 1e3, "revalidate_ms": (t4 - t3) * 1e3,
                "resolve_ms": (t5 - t4) * 1e3, "k": int(st["k"]),
                "candidate_edges": int(st["candidate_edges"]), "valid_edges": int(st["valid_edges"]),
                "path_states": None if path is None else int(len(path)), "path_cost_s": cost, "lazy_removals": rep,
                "straight_line_cost_s": float(np.linalg.norm(g_state[:3] - s_state[:3]) / 0.5)}
    except Exception as ex:  # pragma: no cover
        roadmap = {"error": repr(ex)}

    # ---- N2 extras: the per-map preprocessing chain on the device ------------------------------------------
    preprocess = None
    try:
        if args.skip_extras:
            raise RuntimeError("skipped (--skip-extras)")
        pre_ms, inst_ms = [], []
        for _ in range(5):
            t0 = time.perf_counter()
            pp = ctx.preprocess_map(gm["elevation"], gm.len_x, gm.len_y, gm.pos_x, gm.pos_y,
                                    traversability=gm["traversability"])
            t1 = time.perf_counter()
            pp.install()
            ctx.synchronize()
            t2 = time.perf_counter()
            pp.close()
            pre_ms.append((t1 - t0) * 1e3)
            inst_ms.append((t2 - t1) * 1e3)
        preprocess = {"preprocess_ms_incl_h2d": float(np.median(pre_ms)),
                      "install_ms_incl_tables": float(np.median(inst_ms)), "map": f"{gm.rows}x{gm.cols}"}
    except Exception as ex:  # pragma: no cover
        preprocess = {"error": repr(ex)}

    # ---- C4 extras: 800 x 800 @ 0.04 m map, Params-default robot ---------------------------------------------
    c4 = None
    try:
        if args.skip_extras:
            raise RuntimeError("skipped (--skip-extras)")
        from art_planner_amd.synthetic import RobotDims
        gm4 = make_map(800, 0.04, seed=77, robot=RobotDims(1.05, 0.55, 0.25, 0.1))
        ctx4 = Context(local_rank, "defaults")
        ctx4.upload_map(gm4)
        ctx4.use_torch_stream()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ctx4.sample_and_validate_dev(seed, 0, S, se3, valid)
        ev0.record()
        for i4 in range(3):
            ctx4.sample_and_validate_dev(seed, (i4 + 1) * S, S, se3, valid)
        ev1.record()
        torch.cuda.synchronize()
        ms4 = ev0.elapsed_time(ev1) / 3
        c4 = {"states_per_s": S / (ms4 * 1e-3), "ms_per_batch": ms4, "valid_frac": float(valid.float().mean().item()),
              "map": "800x800@0.04", "robot": "Params defaults"}
        ctx4.close()
        # secondary robot of SURVEY.md 8d on the C2 map
        gm2d = make_map(args.map, args.res, seed=1234, robot=RobotDims(1.05, 0.55, 0.25, 0.1))
        ctx2d = Context(local_rank, "defaults")
        ctx2d.upload_map(gm2d)
        ctx2d.use_torch_stream()
        ctx2d.sample_and_validate_dev(seed, 0, S, se3, valid)
        ev0.record()
        for i4 in range(3):
            ctx2d.sample_and_validate_dev(seed, (i4 + 1) * S, S, se3, valid)
        ev1.record()
        torch.cuda.synchronize()
        c4["c2_map_defaults_robot"] = {"states_per_s": S / (ev0.elapsed_time(ev1) / 3 * 1e-3),
                                       "valid_frac": float(valid.float().mean().item())}
        ctx2d.close()
        ctx.use_torch_stream()
    except Exception as ex:  # pragma: no cover
        c4 = {"error": repr(ex)}

    cpu = None
    if N == 1 and not args.no_cpu_baseline:
        cpu, cpu_labels, _ = cpu_baseline(gm, states)
        n_cpu = len(cpu_labels)
        cpu["labels_match_gpu"] = bool(np.array_equal(cpu_labels, labels[:n_cpu]))
        # the real patched ODE (kind "reference"), single thread, when oracle/_ref travelled here
        try:
            import oracle_py as O
            if O.have_ref():
                rob = O.robot("yaml")
                om = O.OracleMap(gm)
                m = min(20000, len(states))
                poses, inside = om.state_poses(rob, states[:m])
                rb = O.RefChecker(rob.torso, gm["elevation"], gm.len_x, gm.len_y)
                rf = O.RefChecker(rob.foot, gm["elevation_masked"], gm.len_x, gm.len_y)
                t0 = time.perf_counter()
                hb = rb.check(poses[:, 0])
                ok = (hb == 0) | (inside[:, 0] == 0)
                nbox = m
                for k in range(4):  # same short-circuit as the reference
                    idx = np.flatnonzero(ok)
                    hk = rf.check(poses[idx, 1 + k])
                    nbox += len(idx)
                    ok[idx] = np.where(inside[idx, 1 + k] != 0, hk != 0, False)
                dtr = time.perf_counter() - t0
                cpu["reference_ode_single_core_states_per_s"] = m / dtr
                cpu["reference_ode_labels_match_gpu"] = bool(np.array_equal(ok.astype(np.uint8), labels[:m]))
        except Exception as e:  # pragma: no cover
            cpu["reference_ode_error"] = repr(e)

    out = {
        "metric": "validated states/sec on 400x400@0.04m map (sample + validity check)",
        "value": value, "unit": "states/s", "n_gpus": N, "steps": K, "warmup": W,
        "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "C2: lazy_prm_star_min_update front end, 400x400@0.04m Perlin terrain "
                               "(seed 1234) + 12 obstacles, YAML robot, batch sampler + validity checker",
                   "states_per_gpu_per_step": S, "map": f"{args.map}x{args.map}@{args.res}",
                   "sharding": f"sample-index ranges over {N} GPU(s)" +
                               (", accepted-state indices all-gathered over RCCL (complete lists) + the first "
                                f"{'all' if args.materialise < 0 else args.materialise} accepted states of every rank per step "
                                "re-materialised on every rank" if do_gather else "")},
        "roofline": roofline, "cpu_baseline": cpu,
        "valid_fraction": valid_frac, "label_hash_batch0": label_hash,
        "sampler_ms_per_batch": sample_ms, "edges": edges, "pipeline_counts_batch0": pipeline_counts, "motion_cost_c3": motion_cost, "replan_cycle_c5": c5, "roadmap_n1": roadmap, "preprocess_n2": preprocess, "c4_800_defaults": c4,
        "device": ctx.arch, "gather_error": gather_error,
    }
    print(json.dumps(out))
    sys.stdout.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
