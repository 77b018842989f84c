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
`cpu_baseline`.  `roofline.pmc` is measured IN THIS RUN: bench.py re-launches itself (`--pmc-child`, a few
sample+validate batches and nothing else) under `rocprofv3 --kernel-trace --pmc ...`, one pass per counter
group (PMC is never combined with other trace domains), and derives HBM traffic, VALU / LDS busy and L2 hit rate
per pipeline kernel.  If rocprofv3 is not usable it falls back to the committed profiles/pmc_r03.json -- only
when that file was measured on the very kernel sources of this checkout (hash of art_planner_amd/csrc).
"""
import argparse
import glob
import hashlib
import json
import math
import os
import shutil
import sqlite3
import subprocess
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense f16 peak, same guide
N_SIMD = 1024           # 256 CUs x 4 SIMDs
PIPELINE = ["classify_states_kernel", "feet_stream_kernel", "feet_lane_kernel", "resolve_boxes_kernel<2, 64, 0>",
            "resolve_boxes_kernel<2, 64, 3>", "resolve_boxes_kernel<2, 16, 1>", "resolve_boxes_kernel<2, 64, 2>",
            "plane_stage_kernel", "sample_states_kernel"]
PMC_PASSES = [
    ["FETCH_SIZE"],
    ["WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum"],
    ["SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_INSTS_LDS", "SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY",
     "SQ_LDS_BANK_CONFLICT", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"],
    # lane utilisation: thread-cycles of VALU work / (64 lanes x VALU instructions), both from THIS pass.  Calibration:
    # kernels that run with a full EXEC mask (copyBuffer, pre_fill_kernel, morph_kernel) read 1.000 (profiles/r05_lane_util.txt)
    ["SQ_THREAD_CYCLES_VALU", "SQ_INSTS_VALU"],
]


CONTRACT_LINE_MAX = 6144   # bytes; the driver's record could not hold round 5's 25.5 kB line


def _pick(d, keys):
    return None if not isinstance(d, dict) else {k: d.get(k) for k in keys if k in d}


def _clean(x, max_str=160):
    """JSON-safe and bounded: non-finite floats -> None (no NaN / Infinity tokens), floats to 6 significant digits,
    strings clipped, containers recursed."""
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, (np.floating, float)):
        x = float(x)
        return None if not math.isfinite(x) else float(f"{x:.6g}")
    if isinstance(x, (np.integer, int)):
        return int(x)
    if isinstance(x, str):
        return x if len(x) <= max_str else x[:max_str - 3] + "..."
    if isinstance(x, dict):
        return {str(k): _clean(v, max_str) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_clean(v, max_str) for v in x]
    return _clean(repr(x), max_str)


def contract_line(full):
    """The ONE line of the bench contract, built from the full result: the contract keys, `config`, a flat `headline`,
    `roofline` and `cpu_baseline` -- nothing bulky (per-kernel counter tables, thread sweeps, the extras' blocks go to
    the earlier `BENCH_DETAIL ` line and gpurun_out/bench_detail.json).  Guaranteed single line, < CONTRACT_LINE_MAX
    bytes, no NaN / Infinity tokens (tests/test_bench_helpers.py bounds it)."""
    r = full.get("roofline") or {}
    alg = r.get("algorithmic_hbm") or {}
    cpu = full.get("cpu_baseline")
    cfg = dict(full.get("config") or {})
    roof = _pick(r, ("bound", "achieved", "peak", "unit", "frac", "frac_is", "traffic", "hbm_traffic_frac", "kernel",
                     "kernel_ms", "fused_sample_validate_ms", "valu_lane_util_time_weighted", "occupancy_fractions",
                     "validate_only_states_per_s", "dominant_kernel", "pmc_source", "csrc_hash")) or {}
    roof["algorithmic_hbm"] = _pick(alg, ("achieved", "peak", "unit", "ratio_to_peak", "bytes_per_launch", "bytes_per_state"))
    cpu_s = None
    if cpu:
        cpu_s = _pick(cpu, ("value", "unit", "cores", "kind", "sample", "threads_at_best", "single_core_value",
                            "labels_match_gpu", "reference_ode_single_core_states_per_s", "reference_ode_best_states_per_s",
                            "reference_ode_threads_at_best", "reference_ode_states", "reference_ode_labels_match_gpu",
                            "reference_ode_error"))
        e = cpu.get("edges") or {}
        cpu_s["check_motion_edges_per_s_single_thread"] = e.get("check_motion_edges_per_s")
        cpu_s["edge_verdicts_match_gpu"] = e.get("verdicts_match_gpu")
    out = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                    "scaling", "vs_baseline", "dtype", "data")}
    out["config"] = cfg
    out["headline"] = full.get("headline")
    out["value_edges"] = full.get("value_edges")
    out["value_edges_interp"] = full.get("value_edges_interp")
    out["unit_edges"] = full.get("unit_edges")
    out["roofline"] = roof
    out["cpu_baseline"] = cpu_s
    for k in ("valid_fraction", "label_hash_batch0", "device", "gather_error", "headline_includes_exchange", "detail"):
        if k in full:
            out[k] = full.get(k)
    d = full.get("distributed")
    if d:
        out["distributed"] = _pick(d, ("world_size", "rccl_ranks_seen", "exchange", "headline_includes_exchange", "rehearsal",
                                       "no_exchange", "states_per_s_default", "states_per_s_materialise_all",
                                       "states_per_s_materialise_none", "edge_exchange_edges_per_s", "error"))
    for max_str in (200, 120, 80, 48):          # clip the prose harder until the line fits
        line = json.dumps(_clean(out, max_str), allow_nan=False, separators=(",", ":"))
        if len(line) < CONTRACT_LINE_MAX:
            return line
    for k in ("distributed", "headline"):      # last resort: drop the optional blocks, never the contract keys
        out.pop(k, None)
        line = json.dumps(_clean(out, 48), allow_nan=False, separators=(",", ":"))
        if len(line) < CONTRACT_LINE_MAX:
            return line
    raise RuntimeError(f"bench contract line is {len(line)} bytes")


def emit(full, stream=None, partial=False):
    """BENCH_DETAIL line (everything) first, then the contract line LAST.  partial: a run with parts switched off (--no-pmc,
    --skip-extras, --no-cpu-baseline: the profiler's and the scripts' runs) keeps its record apart from the default run's."""
    stream = stream or sys.stdout
    detail_path = None
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        detail_path = os.path.join("gpurun_out", "bench_detail_partial.json" if partial else "bench_detail.json")
        with open(os.path.join(ROOT, detail_path), "w") as f:
            json.dump(_clean(full, 100000), f, indent=1)
    except Exception:
        detail_path = None
    full = dict(full, detail=f"the full record is the earlier stdout line prefixed 'BENCH_DETAIL ' (also {detail_path})")
    stream.write("BENCH_DETAIL " + json.dumps(_clean(full, 100000), allow_nan=False) + "\n")
    stream.write(contract_line(full) + "\n")
    stream.flush()


def csrc_hash():
    """Identity of the kernel sources a PMC profile belongs to."""
    h = hashlib.sha1()
    d = os.path.join(ROOT, "art_planner_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".h", ".hip")) or f == "Makefile":
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


# ---------------------------------------------------------------------------------------------------------
# PMC: child workload + parent-side collection
# ---------------------------------------------------------------------------------------------------------
EDGE_KERNELS = ["motion_plan_kernel", "expand_edges_recs_kernel", "reduce_edges_kernel"]


def pmc_child(args):
    """Only the launches to be measured (bench.py --pmc-child <mode>), nothing else in the process:
       states       warm-up + 3 fused sample+validate batches of S states (the headline step);
       check_motion E pairs of accepted states through artp_check_motions_dev, 1 + 3 times -- the inputs come
                    from one S-state batch, smaller than the ~13 M interior states of an edge batch of 2^18, so that
                    the LARGEST dispatch of every pipeline kernel is the edge batch's;
       sampler      4 batches of artp_sample_states_dev alone."""
    from art_planner_amd.context import Context
    from synthetic import map_from_device, raw_map
    dev = torch.device("cuda", 0)
    ctx = Context(0, "yaml")
    map_from_device(ctx, raw_map(args.map, args.res, seed=1234))
    ctx.use_torch_stream()
    mode = args.pmc_child
    if mode == "check_motion":
        n0 = args.batch   # the ~13 M interior states of E = 2^18 edges still make the edge batch every kernel's largest dispatch
        se3 = torch.empty((n0, 7), dtype=torch.float64, device=dev)
        valid = torch.empty(n0, dtype=torch.uint8, device=dev)
        ctx.sample_and_validate_dev(42, 0, n0, se3, valid)
        torch.cuda.synchronize()
        st = se3.cpu().numpy()
        acc = st[valid.cpu().numpy() != 0]
        ii, jj = pair_edges(acc, args.edges)
        s1 = torch.from_numpy(np.ascontiguousarray(acc[ii])).to(dev)
        s2 = torch.from_numpy(np.ascontiguousarray(acc[jj])).to(dev)
        ev = torch.empty(len(ii), dtype=torch.uint8, device=dev)
        # --pmc-reps R batches (default 4; 0 = the setup alone: collect_pmc_live subtracts it, so that EVERY dispatch of a
        # batch is accounted for -- two validity passes per batch, not just each kernel's largest launch)
        for _ in range(args.pmc_reps):
            ctx.check_motions_dev(s1, s2, ev)
    elif mode == "sampler":
        se3 = torch.empty((args.batch, 7), dtype=torch.float64, device=dev)
        for i in range(4):
            ctx.sample_states_dev(42, i * args.batch, args.batch, se3)
    else:
        se3 = torch.empty((args.batch, 7), dtype=torch.float64, device=dev)
        valid = torch.empty(args.batch, dtype=torch.uint8, device=dev)
        # The step as HIP events see it IN THIS PROCESS, cold and warm: the first batches of a process run ~7 % slower than
        # the steady state the bench line reports (5 warm-up steps + K timed ones) -- the per-kernel durations of a trace of
        # a FEW batches add up to the cold figure, not to ms_per_step (round 3's profiles did exactly that).  The child
        # therefore runs 4 cold batches, 36 more to warm up, and 8 warm ones; the summary's `min` column is the warm launch.
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ctx.sample_and_validate_dev(42, 0, args.batch, se3, valid)
        torch.cuda.synchronize()
        e0.record()
        for i in range(1, 4):
            ctx.sample_and_validate_dev(42, i * args.batch, args.batch, se3, valid)
        e1.record()
        torch.cuda.synchronize()
        cold = e0.elapsed_time(e1) / 3
        for i in range(4, 40):
            ctx.sample_and_validate_dev(42, i * args.batch, args.batch, se3, valid)
        torch.cuda.synchronize()
        e0.record()
        for i in range(40, 48):
            ctx.sample_and_validate_dev(42, i * args.batch, args.batch, se3, valid)
        e1.record()
        torch.cuda.synchronize()
        print("PMC_CHILD_STEP_MS cold %.4f warm %.4f (HIP events around fused sample + validate batches of %d states in this "
              "process: batches 2-4 / batches 41-48)" % (cold, e0.elapsed_time(e1) / 8, args.batch), flush=True)
    torch.cuda.synchronize()
    ctx.close()


def _read_pass(db_path):
    """{kernel: {"n": dispatches, "avg_us": .., counters..}} of one rocprofv3 rocpd database, the LARGEST
    dispatches of each kernel only (the S-state launches; map upload launches smaller grids of other kernels)."""
    db = sqlite3.connect(db_path)
    out = {}
    durs = {}
    for name, d in db.execute("select name, end-start from kernels"):
        durs.setdefault(name, []).append(d)
    for name, ds in durs.items():
        mx = max(ds)
        # "max_us" = the duration of the kernel's LARGE launches in the warm state: the median over the dispatches within
        # a factor 2 of the longest one (the S-state launches; the first few of a process run ~7 % slower than the rest)
        big = sorted(d for d in ds if 2 * d >= mx)
        out[name] = {"n": len(ds), "avg_us": sum(ds) / len(ds) / 1e3, "max_us": big[len(big) // 2] / 1e3,
                     "sum_us": sum(ds) / 1e3}
    try:
        rows = db.execute("select kernel_name, counter_name, avg(value), max(value), sum(value) from counters_collection "
                          "group by kernel_name, counter_name").fetchall()
    except sqlite3.OperationalError:
        rows = []
    for kn, cn, avg, mx, sm in rows:
        out.setdefault(kn, {})[cn] = mx  # the S-state launch is the largest dispatch of its kernel
        out[kn].setdefault("sums", {})[cn] = sm
    return out


def per_batch_from_two_runs(with_reps, setup_only, reps):
    """Per-kernel figures of ONE batch from two profiles of the same child -- `with_reps` ran the setup and `reps` batches,
    `setup_only` the setup alone: (sum over all dispatches of A - the same of B) / reps, durations and counters alike.
    Every dispatch of a batch is in the result (a two-pass checkMotion launches each pipeline kernel twice)."""
    out = {}
    for kn, a in with_reps.items():
        b = setup_only.get(kn, {})
        n = (a.get("n", 0) - b.get("n", 0)) / float(reps)
        if n <= 0:
            continue
        d = {"dispatches_per_batch": n, "max_us": (a.get("sum_us", 0.0) - b.get("sum_us", 0.0)) / reps}
        for cn, sm in a.get("sums", {}).items():
            d[cn] = (sm - b.get("sums", {}).get(cn, 0.0)) / reps
        out[kn] = d
    return out


def collect_pmc_live(args, mode="states", timeout_s=150):
    """Run the PMC passes of one child workload; returns (summary dict | None, note)."""
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "rocprofv3 not found"
    work = tempfile.mkdtemp(prefix="artp_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    merged = {}

    def run_child(i, counters, reps, tag):
        out_dir = os.path.join(work, f"p{i}{tag}")
        cmd = [exe, "--kernel-trace", "--pmc", *counters, "-d", out_dir, "-o", f"p{i}", "--",
               sys.executable, os.path.abspath(__file__), "--pmc-child", mode, "--batch", str(args.batch),
               "--map", str(args.map), "--res", str(args.res), "--edges", str(args.edges), "--pmc-reps", str(reps)]
        r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
        dbs = glob.glob(os.path.join(out_dir, "**", "*_results.db"), recursive=True)
        if r.returncode != 0 or not dbs:
            raise RuntimeError(f"rocprofv3 pass {i}{tag} failed (rc {r.returncode}): {(r.stderr or r.stdout)[-300:]}")
        return _read_pass(dbs[0])

    try:
        for i, counters in enumerate(PMC_PASSES):
            if mode == "check_motion":
                # all dispatches of a batch: (setup + 4 batches) - (setup alone), per kernel
                res = per_batch_from_two_runs(run_child(i, counters, 4, "a"), run_child(i, counters, 0, "b"), 4)
            else:
                res = run_child(i, counters, 4, "")
            for kn, vals in res.items():
                m = merged.setdefault(kn, {})
                if vals.get("SQ_THREAD_CYCLES_VALU") is not None and vals.get("SQ_INSTS_VALU"):
                    m["valu_lane_util"] = lane_utilisation(vals["SQ_THREAD_CYCLES_VALU"], vals["SQ_INSTS_VALU"])
                for key, v in vals.items():
                    if key == "max_us":
                        m.setdefault("max_us_by_pass", []).append(v)
                    elif key not in ("n", "avg_us", "sum_us", "sums"):
                        m[key] = v
    except RuntimeError as ex:
        return None, str(ex)
    except Exception as ex:  # pragma: no cover
        return None, f"pmc collection failed: {ex!r}"
    finally:
        shutil.rmtree(work, ignore_errors=True)
    return summarise_pmc(merged, EDGE_KERNELS if mode == "check_motion" else ()), "live"


def lane_utilisation(thread_cycles_valu, insts_valu):
    """Useful lanes per VALU instruction: SQ_THREAD_CYCLES_VALU / (64 x SQ_INSTS_VALU).  1.0 = every VALU instruction ran
    with a full EXEC mask (calibrated on full-mask kernels: 1.000); an instruction issued for 16 of 64 lanes counts 0.25."""
    return float(thread_cycles_valu) / (64.0 * float(insts_valu)) if insts_valu else None


def summarise_pmc(per_kernel, extra_kernels=()):
    """Derived figures per pipeline kernel of one batch (the largest dispatch of each kernel).  Corrections per MI355X_MICROARCH.md (HBM):
    FETCH_SIZE (KiB) x 2 -- gfx950 tallies 128-B requests as 64 B; WRITE_SIZE (KiB) as reported (calibrated 1.000x on
    sample_states_kernel's 7*8*S bytes in round 1).  busy = SQ_ACTIVE_INST_x * 4 / (1024 SIMDs * kernel cycles),
    kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs."""
    kernels = {}
    tot_fetch = tot_write = tot_us = 0.0
    w_valu = w_lds = w_lane = w_useful = w_lane_us = 0.0
    for kn, v in per_kernel.items():
        pk = next((p for p in list(PIPELINE) + list(extra_kernels) if p in kn), None)
        if pk is None or "GRBM_GUI_ACTIVE" not in v:
            continue
        cyc = v["GRBM_GUI_ACTIVE"] / 8.0
        us = float(np.median(v.get("max_us_by_pass", [0.0])))
        fetch = 2.0 * 1024.0 * v.get("FETCH_SIZE", 0.0)
        write = 1024.0 * v.get("WRITE_SIZE", 0.0)
        hit, miss = v.get("TCC_HIT_sum", 0.0), v.get("TCC_MISS_sum", 0.0)
        k = {"us": us, "hbm_fetch_bytes": fetch, "hbm_write_bytes": write,
             "valu_busy": v.get("SQ_ACTIVE_INST_VALU", 0.0) * 4.0 / (N_SIMD * cyc) if cyc else None,
             "lds_busy": v.get("SQ_ACTIVE_INST_LDS", 0.0) * 4.0 / (N_SIMD * cyc) if cyc else None,
             "lds_insts": v.get("SQ_INSTS_LDS"), "lds_bank_conflict_cycles": v.get("SQ_LDS_BANK_CONFLICT"),
             # every LDS wave-instruction moves at least 64 lanes x 4 B: a lower bound of the LDS traffic
             "lds_gbs_min": (v.get("SQ_INSTS_LDS", 0.0) * 256.0 / (us * 1e-6) / 1e9) if us else None,
             "wait_frac": (v.get("SQ_WAIT_ANY", 0.0) / v["SQ_WAVE_CYCLES"]) if v.get("SQ_WAVE_CYCLES") else None,
             "l2_hit": hit / (hit + miss) if hit + miss else None,
             # useful lanes per VALU instruction (its own PMC pass) and issue occupancy x lane utilisation
             "valu_lane_util": v.get("valu_lane_util"), "dispatches_per_batch": v.get("dispatches_per_batch")}
        k["valu_useful"] = (k["valu_busy"] * k["valu_lane_util"]) if (k["valu_busy"] is not None and k["valu_lane_util"] is not None) else None
        kernels[pk] = k
        if pk not in ("sample_states_kernel",):
            tot_fetch += fetch
            tot_write += write
            if us > 5.0:  # the near-empty fallback launches do not carry the average
                tot_us += us
                w_valu += (k["valu_busy"] or 0.0) * us
                w_lds += (k["lds_busy"] or 0.0) * us
                if k["valu_lane_util"] is not None:
                    w_lane_us += us
                    w_lane += k["valu_lane_util"] * us
                    w_useful += (k["valu_useful"] or 0.0) * us
    if not kernels:
        return None
    return {"kernels": kernels, "validity_hbm_bytes_per_launch": tot_fetch + tot_write,
            "validity_kernel_us_sum": tot_us, "valu_busy_time_weighted": w_valu / tot_us if tot_us else None,
            "lds_busy_time_weighted": w_lds / tot_us if tot_us else None,
            "valu_lane_util_time_weighted": w_lane / w_lane_us if w_lane_us else None,
            "valu_useful_time_weighted": w_useful / w_lane_us if w_lane_us else None, "csrc_hash": csrc_hash()}


def binding_fractions(pmc, hbm_traffic_frac):
    """The occupancy fractions a bound is chosen from (the largest binds): VALU issue and LDS issue, time-weighted over
    the pipeline's kernels, and HBM traffic / time / peak -- all from PMC counters.  {} without counters."""
    if not pmc:
        return {}
    out = {"valu_issue": float(pmc["valu_busy_time_weighted"]), "lds": float(pmc["lds_busy_time_weighted"])}
    if hbm_traffic_frac is not None:
        out["hbm"] = float(hbm_traffic_frac)
    return out


def roofline_fraction(pmc, bound, fracs):
    """`roofline.frac`: for the VALU bound, issue occupancy x lane utilisation (an instruction issued for a quarter of the
    lanes fills the issue port like a full one; only the product says how much of the vector unit does needed work); for
    the other bounds the occupancy fraction itself."""
    if bound is None:
        return None
    if bound == "valu_issue" and pmc and pmc.get("valu_useful_time_weighted") is not None:
        return min(1.0, float(pmc["valu_useful_time_weighted"]))
    return min(1.0, fracs[bound])


def load_committed_pmc():
    """The newest profiles/pmc_rNN.json, only if it was measured on THIS checkout's kernel sources."""
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "pmc_r[0-9][0-9].json")), reverse=True)
    if not paths:
        return None, "no committed PMC profile"
    note = None
    for path in paths:
        name = os.path.basename(path)
        try:
            d = json.load(open(path))
        except Exception as ex:  # pragma: no cover
            note = note or f"unreadable committed PMC profile {name}: {ex!r}"
            continue
        if d.get("csrc_hash") != csrc_hash():
            note = note or f"committed PMC profile {name} is STALE (measured on csrc {d.get('csrc_hash')}, checkout is {csrc_hash()})"
            continue
        return d, f"committed profiles/{name} (same kernel sources)"
    return None, note


# ---------------------------------------------------------------------------------------------------------
# CPU baseline legs (the ONLY users of the oracle in this file)
# ---------------------------------------------------------------------------------------------------------
def parse_cpu_max(text):
    """cgroup v2 cpu.max: "max 100000" -> None (no quota), "800000 100000" -> 8.0 CPUs."""
    parts = text.split()
    if len(parts) < 2 or parts[0] == "max":
        return None
    try:
        q, per = float(parts[0]), float(parts[1])
    except ValueError:
        return None
    return q / per if q > 0 and per > 0 else None


def cgroup_cpu_quota(root="/sys/fs/cgroup"):
    """CPUs the cgroup's bandwidth quota allows (None = unlimited / unknown): v2 cpu.max, v1 cpu.cfs_quota_us / period."""
    try:
        return parse_cpu_max(open(os.path.join(root, "cpu.max")).read())
    except OSError:
        pass
    try:
        q = float(open(os.path.join(root, "cpu", "cpu.cfs_quota_us")).read())
        per = float(open(os.path.join(root, "cpu", "cpu.cfs_period_us")).read())
        return q / per if q > 0 and per > 0 else None
    except (OSError, ValueError):
        return None


def host_cores(affinity=None, quota="probe"):
    """(cores this process can really use, how that was derived): the scheduler affinity mask capped by the cgroup quota --
    os.cpu_count() reports the MACHINE's logical CPUs, which a container may not get (`cores: 256` beside
    an 8.3x speed-up)."""
    if affinity is None:
        affinity = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    if quota == "probe":
        quota = cgroup_cpu_quota()
    usable = affinity if quota is None else max(1, min(affinity, int(quota + 1e-9)))
    return usable, {"os_cpu_count": os.cpu_count(), "sched_affinity": affinity, "cgroup_quota_cpus": quota}


def sweep_thread_counts(affinity):
    """Thread counts of the CPU legs: 1, 8, 32, 64, 128 and every CPU of the affinity mask."""
    return sorted({t for t in (1, 8, 32, 64, 128, affinity) if 1 <= t <= affinity})


def run_threads(n_threads, work):
    """work(k) on n_threads Python threads (the oracle calls release the GIL); returns the wall time."""
    th = [threading.Thread(target=work, args=(k,)) for k in range(n_threads)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    return time.perf_counter() - t0


def cpu_baseline(gm, states, target_s=14.0):
    """The CPU oracle ("port": bit-identical restatement of the reference OMPL+ODE validity path, faithful algorithmic
    structure) timed on this box's host cores on bounded samples of the SAME states the GPU validated: a sweep over
    thread counts (one private checker pair per thread, static partition), the best point is `value`.  `cores` = what this
    process may use (host_cores()).  Only the checker is used here -- never the product path."""
    import oracle_py as O
    rob = O.robot("yaml")
    cores, how = host_cores()
    aff = how["sched_affinity"]
    counts = sweep_thread_counts(aff)
    maps = [O.OracleMap(gm) for _ in range(max(counts))]  # one private checker pair per thread
    n1 = min(16384, len(states))
    t0 = time.perf_counter()
    v1 = maps[0].states_valid(rob, states[:n1])
    r1 = n1 / (time.perf_counter() - t0)
    t_point = target_s / max(len(counts), 1)
    sweep, best, best_labels = [], None, v1
    for nt in counts:
        # sized for ~t_point seconds if the threads scaled like min(nt, 16) cores; a point that scales worse just takes longer
        n = int(min(len(states), max(n1, r1 * t_point * min(nt, 16))))
        chunks = np.array_split(np.arange(n), nt)
        out = np.empty(n, np.uint8)

        def work(k, chunks=chunks, out=out):
            idx = chunks[k]
            if len(idx):
                out[idx] = maps[k].states_valid(rob, states[idx])

        dt = run_threads(nt, work)
        pt = {"threads": nt, "states_per_s": n / dt, "states": n, "speedup_vs_1_thread": (n / dt) / r1}
        sweep.append(pt)
        if best is None or pt["states_per_s"] > best["states_per_s"]:
            best, best_labels = pt, out
    eff = best["states_per_s"] / r1
    return {"value": best["states_per_s"], "unit": "states/s", "cores": cores, "cores_how": how, "kind": "port",
            "threads_at_best": best["threads"], "thread_sweep": sweep,
            "effective_cores_at_best": eff,
            "sample": f"first {best['states']} sampler states of batch 0 (seed 42), oracle/artp_oracle.c faithful mode, "
                      f"best of a sweep over {counts} threads (private checkers, static partition); single thread: "
                      f"{r1:.0f} states/s on {n1}; the best point runs {eff:.1f}x one thread",
            "single_core_value": r1}, best_labels, v1


def reference_ode_rates(gm, states, labels, thread_counts, m=200000):
    """The REAL patched ODE (oracle/_ref, kind "reference") driven like HeightMapBoxChecker, same states: one thread and the
    given thread counts (every thread: dAllocateODEDataForThread + its own world / space / geoms)."""
    import oracle_py as O
    rob = O.robot("yaml")
    om = O.OracleMap(gm)
    m = min(m, len(states))
    poses, inside = om.state_poses(rob, states[:m])
    ok_all = np.zeros(m, np.uint8)

    def leg(nt):
        chunks = np.array_split(np.arange(m), nt)
        gate = threading.Barrier(nt + 1)   # the clock starts when every thread has its ODE world and height fields

        def work(k):
            O.ref_lib().artp_ref_thread_init()
            rb = O.RefChecker(rob.torso, gm["elevation"], gm.len_x, gm.len_y)
            rf = O.RefChecker(rob.foot, gm["elevation_masked"], gm.len_x, gm.len_y)
            gate.wait()
            idx0 = chunks[k]
            if len(idx0):
                hb = rb.check(poses[idx0, 0])
                ok = (hb == 0) | (inside[idx0, 0] == 0)
                for f in range(4):  # same short-circuit as the reference
                    sel = np.flatnonzero(ok)
                    hk = rf.check(poses[idx0[sel], 1 + f])
                    ok[sel] = np.where(inside[idx0[sel], 1 + f] != 0, hk != 0, False)
                ok_all[idx0] = ok.astype(np.uint8)
            rb.close()
            rf.close()

        th = [threading.Thread(target=work, args=(k,)) for k in range(nt)]
        for t in th:
            t.start()
        gate.wait()
        t0 = time.perf_counter()
        for t in th:
            t.join()
        return m / (time.perf_counter() - t0)

    out = {"states": m, "threads": {}}
    for nt in thread_counts:
        out["threads"][str(nt)] = leg(nt)
        out.setdefault("labels_match_gpu", True)
        out["labels_match_gpu"] = bool(out["labels_match_gpu"] and np.array_equal(ok_all, labels[:m]))
    return out


def c1_leg(local_rank):
    """BASELINE config C1: lazy_prm_star_min_update on a flat 100x100 @ 0.1 m map, CPU only --
    the restated LazyPRM* loop over the C oracle (sampler, validity, discrete motion validator; OMPL itself cannot
    be built here) -- next to the batched GPU front end on the same query (-4, -4, yaw 0) -> (4, 4)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import lazy_prm_cpu as LP
    import oracle_py as O
    from art_planner_amd.context import Context
    from art_planner_amd.roadmap import Roadmap
    from synthetic import make_map
    gm = make_map(100, 0.1, flat=True)
    rob, om, smp = O.robot("yaml"), O.OracleMap(gm), O.OracleSampler(gm)
    probe, _ = smp.sample(rob, 1, 0, 64)
    z0 = float(probe[om.states_valid(rob, probe) != 0][0, 2])
    s = np.array([-4.0, -4.0, z0, 0, 0, 0, 1.0])
    g = np.array([4.0, 4.0, z0, 0, 0, 0, 1.0])
    cpu = LP.lazy_prm_star(om, smp, rob, s, g, 2000, seed=42)
    # the reference planner's OWN graph construction on the same accepted states (oracle/prm_incremental.py:
    # LazyPRMStarMinUpdate literally -- start and goal first, predecessors-only k-NN, lazy edge checks)
    import prm_incremental as PI
    se3_all, _ = smp.sample(rob, 42, 0, cpu["states_drawn"])
    acc_all = se3_all[om.states_valid(rob, se3_all) != 0]
    lit = PI.lazy_prm_star_min_update(om, rob, acc_all, s, g, 2000)
    cpu["reference_construction"] = {k: v for k, v in lit.items() if k not in ("path", "graph")}
    path = cpu.pop("path")
    simp, c_simp = LP.shortcut(om, rob, path) if path is not None else (None, None)
    cpu["simplified_path_cost"] = c_simp
    optimum = 8.0 * np.sqrt(2.0) / 0.5
    # the same query on the GPU front end; labels of the CPU leg's sample stream must hash the same
    ctx = Context(local_rank, "yaml")
    ctx.upload_map(gm)
    n_drawn = cpu["states_drawn"]
    gl = ctx.validate_states(ctx.sample_states(42, 0, n_drawn))
    gpu = {"label_hash": hashlib.sha1(gl.tobytes()).hexdigest()[:16]}
    rm = Roadmap(ctx, s, g, n_milestones=2000, seed=42)
    p, c, _ = rm.solve()
    q, d = rm.simplify(p)
    gpu.update({"path_cost": c, "simplified_path_cost": d})
    rm.close()
    # C1's own planner on the device: construction 2 = LazyPRMStarMinUpdate's graph (start, goal, milestones; every
    # vertex to the k nearest of its predecessors, k at its own insertion) as one predecessor-only k-NN batch
    t0 = time.perf_counter()
    rm2 = Roadmap(ctx, s, g, n_milestones=2000, seed=42, construction=2)
    t1 = time.perf_counter()
    p2, c2, rep2 = rm2.solve()
    t2 = time.perf_counter()
    st2 = rm2.stats()
    gpu_ref = {"construction": 2, "vertices": int(st2["vertices"]), "edges": int(st2["candidate_edges"]), "path_cost": c2,
               "path_states": None if p2 is None else int(len(p2)), "lazy_removals": int(rep2),
               "build_ms": (t1 - t0) * 1e3, "solve_ms": (t2 - t1) * 1e3,
               "same_edge_count_as_reference_construction": bool(int(st2["candidate_edges"]) == lit["edges"]),
               "same_cost_as_reference_construction": bool(abs(c2 - lit["path_cost"]) < 1e-9)}
    rm2.close()
    ctx.close()
    return {"config": "C1: flat 100x100@0.1m, start (-4,-4,yaw 0) -> goal (4,4), PathLengthObjective, 2000 milestones",
            "cpu_lazy_prm_star": cpu, "gpu_batch_prm": gpu, "gpu_lazy_prm_star_graph": gpu_ref, "analytic_optimum_s": optimum,
            "labels_match": gpu["label_hash"] == cpu["label_hash"],
            # north star "path cost within 1e-4": the batched GPU plan (simplified, as Planner::plan returns it) against
            # the cost of the reference planner's own incremental construction on the same states
            "path_cost_within_1e-4": bool(abs(d - lit["path_cost"]) < 1e-4 and abs(lit["path_cost"] - optimum) < 1e-4),
            # the two comparisons on their own (rounds 1-2 reported the first under the combined key)
            "path_cost_vs_analytic_optimum_within_1e-4": bool(abs(d - optimum) < 1e-4),
            "path_cost_vs_reference_construction_within_1e-4": bool(abs(d - lit["path_cost"]) < 1e-4),
            "reference_order_graph_cost_vs_reference_construction_within_1e-4": bool(abs(c2 - lit["path_cost"]) < 1e-4)}


def cnn_flops(n):
    h, tot = n, 0.0
    for (k, cin, cout, pool) in ((3, 1, 24, 0), (3, 24, 24, 2), (3, 24, 48, 0), (3, 48, 48, 3), (3, 48, 48, 0),
                                 (15, 48, 48, 0)):
        h = h - k + 1
        tot += 2.0 * k * k * cin * cout * h * h
        if pool == 2:
            h //= 2
        elif pool == 3:
            h -= 2
    return tot


def mfma_clock_probe(timeout_s=90):
    """tests/cpp/mfma_clock_probe.hip compiled and run on this box: what the matrix cores sustain with no memory traffic at all
    (two wavefronts per SIMD: ~17 cycles per v_mfma_f32_16x16x32_f16 at the clock the part keeps under that load; one: ~26).
    Returns the probe's JSON record, or {"error": ...} (no hipcc, no GPU)."""
    import shutil
    import tempfile
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    src = os.path.join(ROOT, "tests", "cpp", "mfma_clock_probe.hip")
    if not (os.path.exists(hipcc) and os.path.exists(src)):
        return {"error": "hipcc or the probe source is missing"}
    try:
        with tempfile.TemporaryDirectory() as td:
            exe = os.path.join(td, "mfma_clock_probe")
            subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O2", "-o", exe, src], stdout=subprocess.DEVNULL,
                                  stderr=subprocess.DEVNULL, timeout=timeout_s)
            out = subprocess.run([exe], capture_output=True, text=True, timeout=timeout_s)
        if out.returncode != 0:
            return {"error": (out.stderr.strip() or f"the probe exited with {out.returncode}")[-200:]}
        rec = json.loads(out.stdout.strip().splitlines()[-1])
        rec["source"] = "tests/cpp/mfma_clock_probe.hip, compiled and run by bench.py on this box"
        return rec
    except Exception as e:  # noqa: BLE001 -- a reported figure, never a reason to fail the bench
        return {"error": f"{type(e).__name__}: {e}"}


def cnn_flops_executed(n):
    """What the kernels execute: conv1 and conv2 (no activation in between) are composed on the host into one 5 x 5
    layer 1 -> 24 (cost_kernels.h conv12_pooled16, inside conv345_kernel's patch phase), everything else as counted by
    cnn_flops (unique outputs: the halo recomputation of the fused tiles and the hi / lo weight pairs are not counted)."""
    h1, h2 = n - 2, n - 4
    return cnn_flops(n) - 2.0 * 9 * 1 * 24 * h1 * h1 - 2.0 * 9 * 24 * 24 * h2 * h2 + 2.0 * 25 * 1 * 24 * h2 * h2


def yaw_of(q):
    return np.arctan2(2 * (q[:, 6] * q[:, 5] + q[:, 3] * q[:, 4]), 1 - 2 * (q[:, 4] ** 2 + q[:, 5] ** 2))


def edge_rows(a, b):
    """MotionCostFunc rows: target x y yaw, start x y yaw (prm_motion_cost.cpp:41-52)."""
    return np.concatenate([b[:, [0, 1]], yaw_of(b)[:, None], a[:, [0, 1]], yaw_of(a)[:, None]], 1).astype(np.float32)


def pair_edges(acc, want, max_gap=3):
    """SURVEY.md 8d edges: accepted state i paired with accepted states i+1 .. i+max_gap when their lateral
    distance is below 2 m.  Returns (index_i, index_j) into acc."""
    ii, jj = [], []
    for d in range(1, max_gap + 1):
        a, b = acc[:-d], acc[d:]
        near = np.flatnonzero(np.hypot(a[:, 0] - b[:, 0], a[:, 1] - b[:, 1]) < 2.0)
        ii.append(near)
        jj.append(near + d)
        if sum(len(x) for x in ii) >= want:
            break
    return np.concatenate(ii)[:want], np.concatenate(jj)[:want]


class Watchdog:
    """A multi-GPU run must not end without its JSON line.  Every rank runs one: the main thread names the stage it
    enters and how long it may take; when a stage overruns (a collective whose peer never arrived, a rank that died)
    rank 0 prints the line from what has been measured so far -- `value` = the no-exchange aggregate when the exchange
    is what hung, `gather_error` = the stage that hung -- and every rank leaves with os._exit (a stuck RCCL kernel cannot
    be unwound).  ctypes calls and torch.cuda.synchronize release the GIL, so the timer thread runs while the main thread
    is stuck inside them."""

    def __init__(self, rank, enabled=True):
        self.rank, self.enabled = rank, enabled
        self.partial = {}          # rank 0: the fields of the line known so far
        self.stage_name, self.deadline = "start", None
        self.lock = threading.Lock()
        self.fired = False
        self.line_printed = False  # rank 0 has printed the real line: a late stage (final barrier) only ends the process
        if enabled:
            threading.Thread(target=self._run, daemon=True).start()

    def stage(self, name, timeout_s):
        with self.lock:
            self.stage_name, self.deadline = name, time.monotonic() + timeout_s

    def done(self):
        with self.lock:
            self.deadline = None

    def _run(self):
        while True:
            time.sleep(0.5)
            with self.lock:
                late = self.deadline is not None and time.monotonic() > self.deadline
                name = self.stage_name
            if late:
                self.fired = True
                if self.rank == 0 and self.line_printed:
                    os._exit(0)
                if self.rank == 0:
                    out = dict(self.partial)
                    out["gather_error"] = f"watchdog: stage '{name}' did not finish in time"
                    out.setdefault("value", None)
                    try:
                        import ctypes
                        ctypes.CDLL(None).fflush(None)
                    except Exception:
                        pass
                    try:
                        print(contract_line(out))
                    except Exception:
                        print(json.dumps({k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "gather_error")}))
                    sys.stdout.flush()
                else:
                    sys.stderr.write(f"[bench rank {self.rank}] watchdog: stage '{name}' did not finish in time\n")
                    sys.stderr.flush()
                os._exit(0 if self.rank == 0 else 3)


def self_launch(n_gpus, argv):
    """`python bench.py --gpus N` with no launcher around it: re-exec under torch.distributed.run, one rank per GPU
    of this node over RCCL (what the driver's `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...` does from outside).  Returns the exit code."""
    import socket
    found = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if found < n_gpus and "--rehearse-on-one-gpu" not in argv:
        raise SystemExit(f"bench.py --gpus {n_gpus}: needs {n_gpus} GPUs on this node, found {found}")
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--spin-up", type=int, default=32,
                    help="untimed batches in front of every timed region, on top of --warmup (see timed_region)")
    ap.add_argument("--batch", type=int, default=1 << 22, help="candidate states per GPU per step")
    ap.add_argument("--edges", type=int, default=1 << 18)
    ap.add_argument("--map", type=int, default=400)
    ap.add_argument("--res", type=float, default=0.04)
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="REHEARSAL of the N > 1 code path on a one-GPU box, not a measurement: every rank uses GPU 0, "
                         "torch.distributed runs over gloo and the device group over the library $ARTP_RCCL_LIB names "
                         "(tests/cpp/loopback_rccl.cpp: RCCL refuses two ranks on one GPU); the line says so in `rehearsal`")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="do not run the rocprofv3 PMC passes")
    ap.add_argument("--pmc-child", nargs="?", const="states", default=None,
                    choices=["states", "check_motion", "sampler"], help=argparse.SUPPRESS)
    ap.add_argument("--pmc-reps", type=int, default=4, help=argparse.SUPPRESS)
    ap.add_argument("--skip-extras", action="store_true", help="no edge / motion-cost measurements (profiling)")
    ap.add_argument("--lanes", type=int, default=1,
                    help="parts of a step's batch validated side by side on as many streams of the context (1..4); "
                         "2 was worth 6 %% before the stream kernels balanced their own load, -0.7 %% on the final build")
    ap.add_argument("--materialise", type=int, default=1 << 16,
                    help="N>1: accepted states of EVERY rank re-materialised on every rank per step, per rank block "
                         "(-1 = all of them, 0 = none; the gathered index lists are always complete)")
    ap.add_argument("--materialise-on", choices=["main", "comm"], default="comm",
                    help="stream the re-materialisation runs on: lane 0's (between two batches) or the exchange's side "
                         "stream, right behind the all-gather (beside the next batch)")
    ap.add_argument("--exchange", choices=["group", "torch"], default="group",
                    help="N>1: who runs the exchange steps -- `group` = artp_group_* of the C ABI (librccl bound by "
                         "libartp.so, one C call per step: what a C++ host uses), `torch` = torch.distributed "
                         "all_gather_into_tensor driven from Python (round 1-3's path)")
    ap.add_argument("--watchdog", type=float, default=120.0,
                    help="N>1: seconds a multi-GPU stage may take before rank 0 prints the line with what it has")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed and run the all-gather path even with one rank (self-test)")
    args = ap.parse_args()
    if args.pmc_child:
        return pmc_child(args)

    N = args.gpus
    if N < 1:
        raise SystemExit("--gpus must be >= 1")
    if N > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: spawn the N ranks ourselves (rank 0 of the child job prints the JSON line on our stdout)
        raise SystemExit(self_launch(N, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.rehearse_on_one_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.rehearse_on_one_gpu and not os.environ.get("ARTP_RCCL_LIB"):
        raise SystemExit("--rehearse-on-one-gpu needs $ARTP_RCCL_LIB (the loopback test double): RCCL itself refuses two ranks on one GPU")
    if world != N:
        raise SystemExit(f"bench.py --gpus {N} was launched with WORLD_SIZE={world}: start it as `python bench.py --gpus {N}` "
                         f"(it spawns its own ranks) or under `python -m torch.distributed.run --nproc-per-node {N} "
                         f"bench.py --gpus {N}`")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU path)")
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"bench.py --gpus {N}: needs {N} GPUs on this node, found {torch.cuda.device_count()}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if N > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from art_planner_amd.context import Context
    from art_planner_amd.distributed import DeviceGroup, shard_first_index
    from synthetic import map_from_device, raw_map

    multi = N > 1 or args.force_dist
    wd = Watchdog(rank, enabled=multi)
    if N > 1:  # rank 0's extras would keep the other ranks waiting at the last barrier for minutes
        args.skip_extras, args.no_cpu_baseline, args.no_pmc = True, True, True
    gather_error = None
    # synthetic inputs: raw terrain + traversability; every derived layer (masked elevation, normals, CDF) comes
    # from the product's device preprocessing, installed as the context's map
    ctx = Context(local_rank, "yaml")
    gm = map_from_device(ctx, raw_map(args.map, args.res, seed=1234))
    # one explicit stream carries the kernels AND the HIP events that time them
    main_stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(main_stream)
    ctx.use_torch_stream()

    S, K, W, seed = args.batch, args.steps, args.warmup, 42
    # Lanes: the batch of a step can be validated as `lanes` contiguous parts on as many streams of the SAME context
    # (include/artp_c.h artp_set_lane).  Nothing joins the lanes between steps, so the short serial kernels at the
    # end of one part's pipeline overlap another part's long kernels.  Default 1 = everything on one stream (what the
    # per-kernel profiles and `roofline.kernel_ms` use): on the final kernels a second lane no longer pays.
    lanes = max(1, min(4, args.lanes))
    if lanes == 4 and args.materialise_on == "comm":
        args.materialise_on = "main"  # lane 3 is taken by the fourth batch part
    if S % (128 * lanes):
        raise SystemExit("--batch must be a multiple of 128 * lanes")
    part = S // lanes
    lane_streams = [main_stream] + [torch.cuda.Stream(device=dev) for _ in range(lanes - 1)]
    for l in range(1, lanes):
        ctx.set_lane(l)
        with torch.cuda.stream(lane_streams[l]):
            ctx.use_torch_stream()
    ctx.set_lane(0)
    se3 = torch.empty((S, 7), dtype=torch.float64, device=dev)
    valid = torch.empty(S, dtype=torch.uint8, device=dev)
    do_gather = (N > 1 or args.force_dist) and not args.no_gather
    comm = torch.cuda.Stream(device=dev) if do_gather else None

    def first_index(step):
        return shard_first_index(step, rank, N, S)

    # ---- warmup (also sizes the fixed-capacity all-gather blocks) -----------------------------
    cap = 0
    counts = [None, None]
    for i in range(max(W, 1)):
        c = ctx.sample_and_validate_dev(seed, first_index(1000000 + i), S, se3, valid, count=True)
        cap = max(cap, c)
    torch.cuda.synchronize()
    mat_cap = 0
    bits_buf = gatherers = all_states = mat_states = mat_counts = idx_tmp = cnt_tmp = done_ev = None
    rccl_ranks_seen = None

    use_grp = None         # the device group that runs the exchange (None: torch.distributed does)
    grp_mat_cap = [None]

    def configure_group(mc):
        """Sizes of the group's state exchange: mc accepted states per rank re-materialised on every rank, looked for
        among the first 8 * mc candidates of a rank's block (everything when mc covers the block's accepted states)."""
        if grp_mat_cap[0] == mc:
            return
        prefix = S if mc >= cap else min(S, 8 * max(mc, 1))
        use_grp.configure(seed, S, mc, prefix)
        grp_mat_cap[0] = mc

    def setup_gather():
        """Buffers of the exchange step + one trial all-gather (outside any timed region)."""
        nonlocal cap, bits_buf, gatherers, all_states, mat_states, mat_counts, idx_tmp, cnt_tmp, done_ev, do_gather, gather_error, rccl_ranks_seen
        from art_planner_amd.distributed import ValidBitmapGatherer, agree_capacity
        if use_grp is not None:
            # the C-ABI group owns the exchange buffers, the side stream and the communicator
            wd.stage("artp_group_ranks_seen + trial exchange", args.watchdog)
            try:
                rccl_ranks_seen = use_grp.ranks_seen()                # an all-reduce of ones on the group's communicator
                if dist is not None:
                    cap = agree_capacity(cap, S, dev)
                else:
                    cap = min(int(cap * 1.1) + 1024, S)
                configure_group(min(cap, max(args.materialise, 1)) if args.materialise >= 0 else cap)
                use_grp.step(0)
                use_grp.synchronize(int(args.watchdog * 1000))
            except Exception as ex:  # pragma: no cover
                gather_error = repr(ex)
                do_gather = False
            wd.done()
            return
        ones = torch.ones(1, device=dev, dtype=torch.int64)
        dist.all_reduce(ones)                                     # the rank count RCCL itself sees
        rccl_ranks_seen = int(ones.item())
        cap = agree_capacity(cap, S, dev)
        words = (S + 63) // 64
        # what crosses xGMI per batch: the validity BITMAP of every rank's candidates (S / 8 bytes = 512 KiB per
        # rank) -- a state is a pure function of (seed, index), so a bit per candidate is all another rank needs
        bits_buf = [torch.zeros(words, dtype=torch.int64, device=dev) for _ in range(2)]
        gatherers = [ValidBitmapGatherer(N, S, dev), ValidBitmapGatherer(N, S, dev)]  # double-buffered
        # every rank's accepted states, re-materialised from the gathered bitmaps: the first mat_cap per rank and
        # step (default 2^16 = 6.5x the reference's whole roadmap, max_n_vertices = 10^4), looked for among the first
        # 8 * mat_cap candidates of the rank's block; anything beyond is one artp_indices_from_bits_dev +
        # artp_sample_states_at_dev away because the bitmaps are complete
        all_states = torch.empty((N, cap, 7), dtype=torch.float64, device=dev)
        mat_states = torch.empty((N, max(1, min(cap, max(args.materialise, 1))), 7), dtype=torch.float64, device=dev)
        mat_counts = torch.zeros(N, dtype=torch.int64, device=dev)
        idx_tmp = torch.zeros(S, dtype=torch.int32, device=dev)
        cnt_tmp = torch.zeros(1, dtype=torch.int64, device=dev)
        done_ev = [torch.cuda.Event(), torch.cuda.Event()]
        for e in done_ev:
            e.record()
        if args.materialise_on == "comm":
            ctx.set_lane(3)                      # a lane of its own (stream, scratch) for the side stream's launches
            with torch.cuda.stream(comm):
                ctx.use_torch_stream()
            ctx.set_lane(0)
        try:  # trial exchange outside the timed region; a failing collective must not lose the whole run
            ctx.pack_valid_bits_dev(valid, bits_buf[0])
            gatherers[0].gather(bits_buf[0])
            torch.cuda.synchronize()
        except Exception as ex:  # pragma: no cover
            gather_error = repr(ex)
            do_gather = False

    def materialise(j, wait=True):
        gb = gatherers[j & 1]
        if wait:
            torch.cuda.current_stream().wait_event(done_ev[j & 1])
        prefix = S if mat_cap >= cap else min(S, 8 * mat_cap)
        # every rank's accepted states in ONE call (two launches whatever N is): rank r's first mat_cap accepted states
        # among its first `prefix` candidates, re-sampled from (seed, global index)
        ctx.materialise_from_bits_dev(seed, gb.gathered, prefix, [shard_first_index(j, r, N, S) for r in range(N)],
                                      mat_cap, all_states[:, :mat_cap] if mat_cap == cap else mat_states, mat_counts)

    def step(i):
        if do_gather and use_grp is not None:
            # ONE call into the C ABI per step: sample + validate the rank's shard, pack the bitmap, all-gather on the
            # group's side stream, re-materialise behind it (group.h); the host only enqueues
            use_grp.step(i)
            return
        b = i & 1
        ready = []
        for l in range(lanes):
            ctx.set_lane(l)
            with torch.cuda.stream(lane_streams[l]):
                lo, hi = l * part, (l + 1) * part
                ctx.sample_and_validate_dev(seed, first_index(i) + lo, part, se3[lo:hi], valid[lo:hi])
                if do_gather:
                    torch.cuda.current_stream().wait_event(done_ev[b])  # buffer b free again
                    ctx.pack_valid_bits_dev(valid[lo:hi], bits_buf[b][lo // 64:hi // 64])
                    ev = torch.cuda.Event()
                    ev.record()
                    ready.append(ev)
        ctx.set_lane(0)
        if do_gather:
            for ev in ready:
                comm.wait_event(ev)
            with torch.cuda.stream(comm):
                gatherers[b].gather(bits_buf[b])               # one bit per candidate state over xGMI
                if args.materialise_on == "comm" and mat_cap > 0:
                    ctx.set_lane(3)
                    materialise(i, wait=False)                 # same stream, right behind the all-gather
                    ctx.set_lane(0)
                done_ev[b].record()
            # --materialise-on main: the accepted states of every rank for the PREVIOUS step (its gather has had a whole
            # step to complete) on lane 0's stream, between two batches.  That was the better place while the
            # re-materialisation was a 0.15 ms kernel (beside the persistent validity grids it cost them +0.36 ms); the
            # lane-per-output-state kernel is short enough to ride behind the all-gather on the side stream (default:
            # +1.2 % per step against `main`).
            if i > 0 and mat_cap > 0 and args.materialise_on == "main":
                materialise(i - 1)

    per_rank_ms = {}

    def timed_region(n_steps, tag="headline"):
        """Exactly n_steps steps, barrier + synchronize on both sides, MAX over ranks (per-rank times kept)."""
        if do_gather and use_grp is not None:
            configure_group(mat_cap)     # (re)allocation outside the clock
        wd.stage(f"timed region '{tag}' ({n_steps} steps)", args.watchdog)
        # Device spin-up, untimed, right in front of every timed region: after any pause (process start, the seconds of
        # setup between regions, even a GEMM burst) the first ~15 batches = 18 ms of this workload run up to 7 % slower
        # and decay to the steady state (scripts/cold_probe.py, profiles/r04_cold_probe.txt: the power management's
        # ramp, not this code's caches -- the same curve after 2 s of idling).  The driver's W = 5 warm-up steps are
        # 6 ms of a 1.1 ms step: without this, a K = 20 region is measured half inside the ramp.
        if dist is not None and args.spin_up:
            dist.barrier()   # a process's first barrier creates the communicator (seconds of idling): before the spin-up
        for i in range(args.spin_up):
            ctx.sample_and_validate_dev(seed, first_index(3000000 + i), S, se3, valid)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n_steps):
            step(i)
        if do_gather and use_grp is None and mat_cap > 0 and args.materialise_on == "main":
            materialise(n_steps - 1)
        torch.cuda.synchronize()
        t_local = time.perf_counter() - t0
        if dist is not None:
            dist.barrier()
        dt_ = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt_], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_ = float(t.item())
            tl = torch.tensor([t_local], device=dev, dtype=torch.float64)
            allt = torch.empty(world, device=dev, dtype=torch.float64)
            dist.all_gather_into_tensor(allt, tl)
            per_rank_ms[tag] = [float(x) / n_steps * 1e3 for x in allt.tolist()]
        else:
            per_rank_ms[tag] = [t_local / n_steps * 1e3]
        wd.done()
        return dt_

    # ---- the cold step: the first batches after the device idled (a 10 Hz replanner never reaches the steady state the
    # headline region is measured in; profiles/r04_cold_probe.txt) -- HIP events around 4 batches after 2 s without work
    cold_ms = None
    if not multi and not args.skip_extras:
        time.sleep(2.0)
        e0c, e1c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0c.record()
        for i in range(4):
            ctx.sample_and_validate_dev(seed, first_index(5000000 + i), S, se3, valid)
        e1c.record()
        torch.cuda.synchronize()
        cold_ms = e0c.elapsed_time(e1c) / 4

    # ---- the headline region -----------------------------------------------------------------------
    base_line = {"metric": "validated states/sec on 400x400@0.04m map (sample + validity check)", "unit": "states/s",
                 "n_gpus": N, "steps": K, "warmup": W, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                 "dtype": "f32", "data": "synthetic",
                 "config": {"workload": "C2: lazy_prm_star_min_update front end, 400x400@0.04m Perlin terrain (seed "
                                        "1234) + 12 obstacles, YAML robot, batch sampler + validity checker",
                            "states_per_gpu_per_step": S, "map": f"{args.map}x{args.map}@{args.res}",
                            "spin_up_batches_before_each_timed_region": args.spin_up}}
    no_exchange = None
    if multi:
        wd.partial = dict(base_line, value=None)   # whatever happens from here on, the line has its fixed fields
    if multi and do_gather:
        # first the same K steps WITHOUT any exchange (only the barrier and the clock cross ranks): if the exchange hangs
        # on this node, the watchdog's line still carries a measured aggregate -- flagged as such
        do_gather, keep = False, True
        k0 = max(3, min(K, 20))
        dt0 = timed_region(k0, "no_exchange")
        do_gather = keep
        no_exchange = {"states_per_s": N * S * k0 / dt0, "ms_per_step": dt0 / k0 * 1e3, "steps": k0,
                       "per_rank_ms_per_step": per_rank_ms.get("no_exchange")}
        wd.partial = dict(base_line, value=no_exchange["states_per_s"], ms_per_step=no_exchange["ms_per_step"],
                          steps=k0, headline_includes_exchange=False,
                          distributed={"world_size": world, "rccl_ranks_seen": rccl_ranks_seen,
                                       "no_exchange": no_exchange, "exchange": "not reached"})
    # Only now -- with a measured aggregate already in the watchdog's hands -- anything that can hang on a node nobody
    # has run on before: the RCCL communicator of the C ABI's device group and the trial exchange.
    if multi and do_gather and args.exchange == "group" and args.lanes == 1:
        # the device group of the C ABI, one process per GPU: rank 0 makes the RCCL id, torch.distributed (already up
        # for the barrier and the max-over-ranks clock) carries its 128 bytes to the others.  The group owns its own
        # context (the map is replicated per member, as it is across ranks)
        wd.stage("artp_group_create_rank (RCCL communicator)", args.watchdog)
        grp = None
        try:
            uid = torch.zeros(128, dtype=torch.uint8, device=dev)
            if rank == 0:
                uid = torch.frombuffer(bytearray(DeviceGroup.unique_id()), dtype=torch.uint8).to(dev)
            dist.broadcast(uid, src=0)
            grp = DeviceGroup.from_rank(local_rank, rank, world, bytes(uid.cpu().numpy().tobytes()), "yaml")
            map_from_device(grp.contexts[0], raw_map(args.map, args.res, seed=1234))
            with torch.cuda.stream(main_stream):
                grp.contexts[0].use_torch_stream()
        except Exception as ex:  # pragma: no cover
            gather_error = "group: " + repr(ex)
            if grp is not None:
                try:
                    grp.close()
                except Exception:
                    pass
            grp = None
        # every rank must take the same path: fall back to the torch exchange everywhere if any rank failed
        ok_t = torch.tensor([1 if grp is not None else 0], device=dev, dtype=torch.int64)
        dist.all_reduce(ok_t, op=dist.ReduceOp.MIN)
        if int(ok_t.item()) == 0 and grp is not None:
            grp.close()
            grp = None
        use_grp = grp
        wd.done()
    if do_gather:
        setup_gather()
    mat_cap = (cap if args.materialise < 0 else min(cap, args.materialise)) if do_gather else 0
    dt = timed_region(K)
    value = N * S * K / dt
    if multi:
        wd.partial = dict(base_line, value=value, ms_per_step=dt / K * 1e3, headline_includes_exchange=bool(do_gather),
                          distributed={"world_size": world, "rccl_ranks_seen": rccl_ranks_seen, "no_exchange": no_exchange,
                                       "per_rank_ms_per_step": per_rank_ms.get("headline"),
                                       "exchange": ("artp_group (C ABI, librccl bound by libartp.so)" if use_grp
                                                    else "torch.distributed") if do_gather else None})

    # ---- the exchange step under the other materialisation settings and the edge exchange, all ranks; at N = 1
    # without --force-dist a one-rank RCCL group is brought up AFTER the headline so that every N reports the block ----
    headline_gathers = do_gather
    dist_extras = None
    if N == 1 and dist is None and not args.no_gather and not args.skip_extras:
        try:
            if args.exchange == "group":
                # a one-member RCCL group of the C ABI (its own context: the map is replicated per member)
                use_grp = DeviceGroup.single_process([local_rank], "yaml")
                map_from_device(use_grp.contexts[0], raw_map(args.map, args.res, seed=1234))
                with torch.cuda.stream(main_stream):
                    use_grp.contexts[0].use_torch_stream()
            else:
                import torch.distributed as dist
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
                dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
                comm = torch.cuda.Stream(device=dev)
            do_gather = True
            setup_gather()
        except Exception as ex:  # pragma: no cover
            gather_error = repr(ex)
            do_gather = False
            dist = None
    if do_gather:
        k2 = max(3, min(K, 20))
        mat_cap = (cap if args.materialise < 0 else min(cap, args.materialise))
        dist_extras = {"world_size": world, "rccl_ranks_seen": rccl_ranks_seen,
                       "exchange": ("artp_group_* of the C ABI (librccl bound by libartp.so; one call per step)"
                                    if use_grp is not None else "torch.distributed " + dist.get_backend()),
                       "headline_includes_exchange": bool(headline_gathers),
                       "rehearsal": ("NOT a measurement: %d ranks share GPU 0, torch.distributed over gloo, the device group over "
                                     "$ARTP_RCCL_LIB = %s" % (world, os.environ.get("ARTP_RCCL_LIB"))) if args.rehearse_on_one_gpu else None,
                       "materialise_default": args.materialise, "no_exchange": no_exchange,
                       "per_rank_ms_per_step": per_rank_ms,
                       "states_per_s_default": value if headline_gathers else N * S * k2 / timed_region(k2, "default")}
        for name, mc in (("states_per_s_materialise_all", cap), ("states_per_s_materialise_none", 0)):
            mat_cap = mc
            dist_extras[name] = N * S * k2 / timed_region(k2, name[13:])
        mat_cap = (cap if args.materialise < 0 else min(cap, args.materialise))
        dist_extras["per_gpu_states_per_s"] = {k_: v_ / N for k_, v_ in dist_extras.items()
                                                 if k_.startswith("states_per_s_")}
        # edges follow the GPU that owns the source state: every rank validates E edges between its
        # own accepted states (0.5 m interpolation rule), scores them with the learned cost when weights are
        # there (else the length triple), packs the valid ones as {u32 i, u32 j, f32 cost[3]} and all-gathers them
        try:
            from art_planner_amd.distributed import EdgeResultGatherer
            ctx.sample_and_validate_dev(seed, first_index(2000000), S, se3, valid)
            torch.cuda.synchronize()
            st_h = se3.cpu().numpy()
            pos = np.flatnonzero(valid.cpu().numpy())
            ii, jj = pair_edges(st_h[pos], args.edges)
            # the blocks of the all-gather have ONE size: the smallest edge count any rank found (normally --edges)
            wd.stage("edge exchange", args.watchdog)
            e_t = torch.tensor([len(ii)], device=dev, dtype=torch.int64)
            if dist is not None:
                dist.all_reduce(e_t, op=dist.ReduceOp.MIN)
            E = int(e_t.item())
            if E == 0:
                raise RuntimeError("no edge pairs on some rank")
            ii, jj = ii[:E], jj[:E]
            s1 = torch.from_numpy(np.ascontiguousarray(st_h[pos[ii]])).to(dev)
            s2 = torch.from_numpy(np.ascontiguousarray(st_h[pos[jj]])).to(dev)
            ei = torch.from_numpy(pos[ii].astype(np.int32)).to(dev)
            ej = torch.from_numpy(pos[jj].astype(np.int32)).to(dev)
            rows = torch.from_numpy(edge_rows(st_h[pos[ii]], st_h[pos[jj]])).to(dev)
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import convert_weights
            ctx.cost_load_weights(convert_weights.to_blob(convert_weights.random_params(0)))
            elv_t = torch.from_numpy(np.ascontiguousarray(gm["elevation"][::-1, ::-1]).astype(np.float32)).to(dev)
            ctx.cost_update_map_dev(elv_t, gm.res, gm.len_x, gm.len_y)
            ev = torch.empty(E, dtype=torch.uint8, device=dev)
            cost3 = torch.empty((E, 3), dtype=torch.float32, device=dev)
            rec = torch.zeros((E, 5), dtype=torch.int32, device=dev)
            ecount = torch.zeros(1, dtype=torch.int64, device=dev)
            eg = EdgeResultGatherer(N, E, dev) if use_grp is None else None

            def edge_step():
                ctx.check_edges_interp_dev(s1, s2, ev)
                ctx.cost_query_dev(rows, cost3)
                if use_grp is not None:
                    # pack + all-gather of the 20-byte records inside the C ABI; the next exchange waits for this one
                    # on the device, the host does not
                    use_grp.exchange_edges([(ev.data_ptr(), ei.data_ptr(), ej.data_ptr(), cost3.data_ptr(), E)], E)
                    return
                ctx.pack_edge_results_dev(ev, ei, ej, cost3, rec, ecount)
                ready = torch.cuda.Event()
                ready.record()
                comm.wait_event(ready)
                with torch.cuda.stream(comm):
                    eg.gather(rec, ecount)
                torch.cuda.current_stream().wait_stream(comm)

            edge_step()
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            t0 = time.perf_counter()
            for _ in range(k2):
                edge_step()
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            t = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
            if dist is not None:
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
            if use_grp is not None:
                from art_planner_amd.distributed import device_view
                rec_p, cnt_p = use_grp.edge_pointers(0)
                torch.cuda.synchronize()
                cnts = device_view(cnt_p, (N,), "<i8", dev).cpu().numpy()   # the W record counts, as gathered on this rank
                n_gathered, ok = int(cnts.sum()), bool((cnts <= E).all())
            else:
                ij, cst, ok = eg.global_records(0, S)
                n_gathered = int(ij.shape[0])
            wd.done()
            dist_extras["edges"] = {"edges_per_gpu_per_step": E, "steps": k2,
                                    "edges_per_s": N * E * k2 / float(t.item()),
                                    "valid_edges_gathered": n_gathered, "blocks_ok": bool(ok),
                                    "what": "0.5 m interpolation rule + learned cost (seeded weights) + "
                                            "{u32 i, u32 j, f32 cost[3]} all-gather (RCCL, side stream)"}
        except Exception as ex:  # pragma: no cover
            dist_extras["edges"] = {"error": repr(ex)}

    if rank != 0:
        if dist is not None:
            wd.stage("waiting for rank 0's report", max(args.watchdog, 600.0))
            dist.barrier()
            wd.done()
            if use_grp is not None:
                use_grp.close()
            dist.destroy_process_group()
        return
    if multi:
        wd.stage("rank 0: roofline timing and report", max(args.watchdog, 600.0))
        wd.partial["distributed"] = dist_extras

    # ---- everything below: rank 0, outside the timed region ------------------------------------
    # batch 0 again (deterministic) for the roofline, label hash, edges and CPU baseline
    ctx.sample_and_validate_dev(seed, 0, S, se3, valid)
    torch.cuda.synchronize()
    labels = valid.cpu().numpy()
    label_hash = hashlib.sha1(labels.tobytes()).hexdigest()[:16]
    valid_frac = float(labels.mean())

    # dominant kernel: the validity pipeline; HIP events on the stream it is launched on
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
    # the fused step the timed region runs (sampling inside the first validity kernel where the build has it)
    ev0.record()
    for i in range(reps):
        ctx.sample_and_validate_dev(seed, i * S, S, se3, valid)
    ev1.record()
    torch.cuda.synchronize()
    step_ms = ev0.elapsed_time(ev1) / reps
    ctx.sample_and_validate_dev(seed, 0, S, se3, valid)
    torch.cuda.synchronize()

    pmc, pmc_note = (None, "skipped")
    if N == 1 and not args.no_pmc:
        pmc, pmc_note = collect_pmc_live(args)
        if pmc is None:
            live_note = pmc_note
            pmc, pmc_note = load_committed_pmc()
            pmc_note = f"{pmc_note}; live collection: {live_note}"
        else:
            try:
                os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                json.dump(pmc, open(os.path.join(ROOT, "gpurun_out", "pmc_live.json"), "w"), indent=1)
            except Exception:
                pass
    elif N > 1:
        pmc_note = "N > 1: PMC passes only run at N = 1"
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9
    if pmc is None and N > 1:
        pmc, pmc_note = load_committed_pmc()    # hash-guarded: only a profile of these very kernel sources
        pmc_note = f"{pmc_note} (N > 1: the live PMC passes only run at N = 1)"
    traffic = pmc["validity_hbm_bytes_per_launch"] if pmc else None
    hbm_traffic_frac = None if traffic is None else traffic / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
    # The bound that BINDS, derived from this run's counters: the largest of the three occupancy fractions.  The
    # algorithmic-byte figure of the bench contract (what the reference's window scans read / kernel time) is kept under
    # its own key: the exact range / stride / partner tables answer the scans without reading those bytes, so it exceeds
    # the HBM peak and is a statement about the ALGORITHM, not a fraction of any roofline.
    fracs = binding_fractions(pmc, hbm_traffic_frac)
    bound = max(fracs, key=fracs.get) if fracs else None
    roofline = {
        "bound": bound, "achieved": None if bound is None else fracs[bound], "peak": 1.0 if bound else None,
        "unit": {"valu_issue": "fraction of VALU issue cycles (SQ_ACTIVE_INST_VALU * 4 / (1024 SIMDs * kernel cycles)), "
                               "time-weighted over the pipeline's kernels",
                 "lds": "fraction of LDS issue cycles", "hbm": "fraction of the 8 TB/s HBM peak (PMC traffic / time)",
                 None: None}[bound],
        "frac": roofline_fraction(pmc, bound, fracs),
        "frac_is": "VALU issue occupancy x VALU lane utilisation, time-weighted over the pipeline's kernels (bound valu_issue); "
                   "the occupancy fraction itself for the other bounds",
        "valu_lane_util_time_weighted": None if not pmc else pmc.get("valu_lane_util_time_weighted"),
        "occupancy_fractions": fracs,
        "traffic": traffic, "hbm_traffic_frac": hbm_traffic_frac,
        "kernel": "validity pipeline (artp_validate_states_dev)", "kernel_ms": k_ms,
        "fused_sample_validate_ms": step_ms,
        # the timed region's own rate: `lanes` parts side by side, sampler included (the figures above are ONE batch
        # alone on ONE stream, which is what the rocprofv3 per-kernel durations in profiles/ correspond to)
        "timed_region": {"lanes": lanes, "ms_per_batch": dt / K * 1e3,
                         "algorithmic_GBps": alg_bytes / (dt / K) / 1e9},
        "kernel_launches": "classify_states_kernel + feet_stream_kernel<4> + resolve_boxes_kernel<2,64,0> + 5 "
                           "near-empty fallback launches (profiles/README.md)",
        "validate_only_states_per_s": S / (k_ms * 1e-3),
        "algorithmic_hbm": {
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "ratio_to_peak": achieved / HBM_PEAK_GBS,
            "bytes_per_launch": alg_bytes, "bytes_per_state": alg_bytes / S,
            "note": "SURVEY.md 8d's yardstick: ALGORITHMIC bytes = 4 B x the heightfield vertices of all five index windows "
                    "of a state + 29 B, no credit for early-outs, / kernel time.  Above the HBM peak because the exact "
                    "tables make the scans unnecessary (labels identical): credit as throughput, not as a roofline "
                    "fraction"},
        "note": "bound / frac = the largest occupancy fraction of the pipeline measured by this run's PMC passes "
                "(`binding.per_kernel` has every kernel; classify_states_kernel itself is bound by its L2 requests x "
                "latency / L1 miss concurrency, DESIGN.md 4.1)",
        "binding": None if not pmc else {
            "bound": bound, "valu_busy_time_weighted": pmc["valu_busy_time_weighted"],
            "valu_busy_note": "SQ_ACTIVE_INST_VALU * 4 cycles / (1024 SIMDs * GRBM_GUI_ACTIVE / 8).  A value of 1.0 - 1.1 "
                              "means saturated: instructions that issue with an empty EXEC mask (16-lane groups of a "
                              "wavefront on different branches) retire in fewer than the four cycles the formula charges",
            "lds_busy_time_weighted": pmc["lds_busy_time_weighted"],
            "hbm_traffic_frac_of_peak": hbm_traffic_frac,
            "per_kernel": pmc["kernels"],
            "pmc_kernel_us_sum_vs_hip_events_ms": [pmc["validity_kernel_us_sum"], k_ms]},
        "pmc_source": pmc_note, "csrc_hash": csrc_hash()}

    # sampler alone
    ev0.record()
    for _ in range(reps):
        ctx.sample_states_dev(seed, 0, S, se3)
    ev1.record()
    torch.cuda.synchronize()
    sample_ms = ev0.elapsed_time(ev1) / reps
    ctx.sample_and_validate_dev(seed, 0, S, se3, valid)
    torch.cuda.synchronize()

    # edges: E >= 2^18 pairs of accepted states closer than 2 m
    states = se3.cpu().numpy()
    acc = states[labels != 0]
    ii, jj = pair_edges(acc, args.edges)
    a, b = acc[ii], acc[jj]
    E = 0 if args.skip_extras else len(a)
    edges = {}
    if E > 0:
        s1 = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        s2 = torch.from_numpy(np.ascontiguousarray(b)).to(dev)
        evd = torch.empty(E, dtype=torch.uint8, device=dev)
        lt = torch.empty(E, dtype=torch.float64, device=dev)
        for name, fn in (("check_motion", lambda: ctx.check_motions_dev(s1, s2, evd)),
                         ("interp_0p5m", lambda: ctx.check_edges_interp_dev(s1, s2, evd))):
            fn()
            torch.cuda.synchronize()
            ev0.record()
            for _ in range(3):
                fn()
            ev1.record()
            torch.cuda.synchronize()
            ms = ev0.elapsed_time(ev1) / 3
            edges[name] = {"edges": E, "edges_per_s": E / (ms * 1e-3), "ms": ms,
                           "valid_frac": float(evd.float().mean().item())}
        del lt
        # what binds the edge path: the PMC passes on an edge batch alone (same arithmetic as roofline.binding)
        if N == 1 and not args.no_pmc:
            pmc_e, note_e = collect_pmc_live(args, "check_motion")
            if pmc_e is not None:
                ms_cm = edges["check_motion"]["ms"]
                hbm_e = pmc_e["validity_hbm_bytes_per_launch"] / (ms_cm * 1e-3) / 1e9 / HBM_PEAK_GBS
                fr_e = binding_fractions(pmc_e, hbm_e)
                edges["check_motion"]["binding"] = {
                    "bound": max(fr_e, key=fr_e.get), "occupancy_fractions": fr_e,
                    "valu_busy_time_weighted": pmc_e["valu_busy_time_weighted"],
                    "valu_lane_util_time_weighted": pmc_e.get("valu_lane_util_time_weighted"),
                    "valu_useful_time_weighted": pmc_e.get("valu_useful_time_weighted"),
                    "lds_busy_time_weighted": pmc_e["lds_busy_time_weighted"],
                    "hbm_bytes_per_batch": pmc_e["validity_hbm_bytes_per_launch"],
                    "hbm_traffic_frac_of_peak": pmc_e["validity_hbm_bytes_per_launch"] / (ms_cm * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "pmc_kernel_us_sum_vs_hip_events_ms": [pmc_e["validity_kernel_us_sum"], ms_cm],
                    "per_kernel": pmc_e["kernels"], "pmc_source": note_e,
                    "what": "one artp_check_motions_dev batch of E edges under rocprofv3 --pmc, ALL its dispatches (both "
                            "validity passes): (setup + 4 batches) - (setup alone), per kernel, / 4 (bench.py --pmc-child "
                            "check_motion --pmc-reps 4 / 0): plan + scan + expand, the validity pipeline twice, reduce"}
            else:
                edges["check_motion"]["binding"] = {"error": note_e}

    # ---- C3 / C4 extras: learned motion cost (seeded random weights: the trained ones are git-LFS stubs) ----
    motion_cost = None
    try:
        if args.skip_extras:
            raise RuntimeError("skipped (--skip-extras)")
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import convert_weights
        ctx.cost_load_weights(convert_weights.to_blob(convert_weights.random_params(0)))
        motion_cost = {"weights": "seeded random (tools/convert_weights.random_params(0))"}
        for tag, n_map, g_ in (("c3_400", gm.rows, gm), ("c4_800", 800, None)):
            if g_ is None:
                g_ = raw_map(800, 0.04, seed=77)
            elv = np.ascontiguousarray(g_["elevation"][::-1, ::-1]).astype(np.float32)  # cost_query_server.py:66-74
            ctx.cost_update_map(elv, g_.res, g_.len_x, g_.len_y)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                ctx.cost_update_map(elv, g_.res, g_.len_x, g_.len_y)   # H2D of the map + CNN, synchronous
            wall_ms = (time.perf_counter() - t0) / 5 * 1e3
            elv_t = torch.from_numpy(elv).to(dev)
            ctx.cost_update_map_dev(elv_t, g_.res, g_.len_x, g_.len_y)
            torch.cuda.synchronize()
            ev0.record()
            for _ in range(10):
                ctx.cost_update_map_dev(elv_t, g_.res, g_.len_x, g_.len_y)  # the three launches only (HIP events)
            ev1.record()
            torch.cuda.synchronize()
            kern_ms = ev0.elapsed_time(ev1) / 10
            gf = cnn_flops(n_map) / 1e9
            gfx = cnn_flops_executed(n_map) / 1e9
            motion_cost[tag] = {"cnn_gflop": gf, "cnn_gflop_executed": gfx,
                                "cnn_gflop_note": "cnn_gflop = SURVEY 8a-R8's algorithmic count (six layers); executed = conv1 and "
                                                  "conv2 composed into one 5 x 5 layer on the host; all of it on MFMA",
                                "cnn_ms_incl_h2d": wall_ms, "cnn_kernels_ms": kern_ms,
                                "cnn_launches": "conv345_kernel (conv1 o conv2 + pool inside its patch phase) + conv_ksplit_kernel (15 x 15)",
                                "cnn_kernels_tflops": gf / kern_ms,
                                "cnn_kernels_frac_of_mfma_f16_peak": gf / kern_ms / MFMA_F16_PEAK_TFLOPS,
                                "cnn_kernels_frac_of_mfma_f16_peak_executed_flops": gfx / kern_ms / MFMA_F16_PEAK_TFLOPS,
                                "feature_map": list(ctx.cost_features().shape[:2])}
        probe = mfma_clock_probe()
        motion_cost["mfma_probe"] = probe
        if "error" not in probe:
            for tag in ("c3_400", "c4_800"):
                motion_cost[tag]["cnn_kernels_frac_of_measured_mfma_rate"] = motion_cost[tag]["cnn_kernels_tflops"] / probe["two_per_simd_tflops"]
        # back to the C3 map for the queries
        elv = np.ascontiguousarray(gm["elevation"][::-1, ::-1]).astype(np.float32)
        ctx.cost_update_map(elv, gm.res, gm.len_x, gm.len_y)
        if E > 0:
            em = edge_rows(a, b)
            for Bq in (50000, 1 << 20):
                rows = np.concatenate([em] * ((Bq + E - 1) // E))[:Bq]
                edges_t = torch.from_numpy(np.ascontiguousarray(rows)).to(dev)
                cost_t = torch.empty((Bq, 3), dtype=torch.float32, device=dev)
                n_q = 50 if Bq < (1 << 18) else 10   # a 10 us launch timed five times is at the mercy of one hiccup
                for _ in range(3):
                    ctx.cost_query_dev(edges_t, cost_t)
                torch.cuda.synchronize()
                ev0.record()
                for _ in range(n_q):
                    ctx.cost_query_dev(edges_t, cost_t)
                ev1.record()
                torch.cuda.synchronize()
                q_ms = ev0.elapsed_time(ev1) / n_q
                motion_cost[f"cost_queries_{Bq}"] = {"ms": q_ms, "queries_per_s": Bq / (q_ms * 1e-3)}
    except Exception as ex:  # pragma: no cover
        motion_cost = {"error": repr(ex)}

    # ---- C5 (BASELINE configs[4] as written): params.planner.elevation_layer = "upper_bound" -- body checker, sampler
    # and the whole preprocessing read `upper_bound` (validity_checker_body.cpp:52-55, basic.cpp:45-104) -- on a
    # persistent HBM map; 100 map versions, each changing ~5 % of the cells in 3 rectangles of that layer (body slot)
    # and of the masked layer derived from it (feet slot); per cycle = dirty-rectangle upload + table refresh +
    # feature-map refresh + 2^18 states + 50 000 cost edges.  Labels after the last version are checked
    # against the CPU oracle on the updated layers (outside the timed cycles).
    c5 = None
    try:
        if args.skip_extras or E == 0:
            raise RuntimeError("skipped (--skip-extras)")
        ctx5 = Context(local_rank, "yaml")
        gm5 = map_from_device(ctx5, raw_map(args.map, args.res, seed=1234, with_upper_bound=True), body_layer="upper_bound")
        ctx5.use_torch_stream()
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import convert_weights
        ctx5.cost_load_weights(convert_weights.to_blob(convert_weights.random_params(0)))
        rng5 = np.random.default_rng(55)
        ub5 = gm5["upper_bound"].copy(order="F")
        mk5 = gm5["elevation_masked"].copy(order="F")
        n5, cyc, stg = 1 << 18, [], {"rects": [], "cnn": [], "states": [], "cost": []}
        se5 = torch.empty((n5, 7), dtype=torch.float64, device=dev)
        va5 = torch.empty(n5, dtype=torch.uint8, device=dev)
        rows5 = torch.from_numpy(np.ascontiguousarray(edge_rows(a, b)[:50000])).to(dev)
        cost5 = torch.empty((rows5.shape[0], 3), dtype=torch.float32, device=dev)
        side = int(round(np.sqrt(0.05 * gm5.rows * gm5.cols / 3)))  # 3 squares = 5 % of the cells
        ctx5.cost_update_map(np.ascontiguousarray(ub5[::-1, ::-1]), gm5.res, gm5.len_x, gm5.len_y)
        ctx5.sample_and_validate_dev(seed, 0, n5, se5, va5)
        torch.cuda.synchronize()
        ver0 = ctx5.map_version()
        for c_i in range(100):
            t0 = time.perf_counter()
            org5 = []
            for _ in range(3):
                r0, c0 = int(rng5.integers(0, gm5.rows - side)), int(rng5.integers(0, gm5.cols - side))
                ub5[r0:r0 + side, c0:c0 + side] += np.float32(rng5.normal(0, 0.01))
                m_ = mk5[r0:r0 + side, c0:c0 + side]
                mk5[r0:r0 + side, c0:c0 + side] = np.where(np.isfinite(m_), ub5[r0:r0 + side, c0:c0 + side], m_)
                org5.append((r0, c0))
            # dirty cells -> HBM + tables: the version's rectangles of a slot in one (asynchronous) call
            ctx5.update_layer_rects(0, [ub5[r0:r0 + side, c0:c0 + side] for r0, c0 in org5], org5)
            ctx5.update_layer_rects(1, [mk5[r0:r0 + side, c0:c0 + side] for r0, c0 in org5], org5)
            t1 = time.perf_counter()
            ctx5.cost_update_map(np.ascontiguousarray(ub5[::-1, ::-1]), gm5.res, gm5.len_x, gm5.len_y)  # features
            t2 = time.perf_counter()
            ctx5.sample_and_validate_dev(seed, 7_000_000 + c_i * n5, n5, se5, va5)
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            ctx5.cost_query_dev(rows5, cost5)
            torch.cuda.synchronize()
            t4 = time.perf_counter()
            cyc.append((t4 - t0) * 1e3)
            for k_, v_ in zip(("rects", "cnn", "states", "cost"), (t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
                stg[k_].append(v_ * 1e3)
        c5 = {"versions": 100, "elevation_layer": "upper_bound (body slot + sampler + preprocessing; feet slot = the "
                                                  "masked layer derived from it)",
              "cycle_ms_median": float(np.median(cyc)), "cycle_ms_max": float(np.max(cyc)),
              "stage_ms_median": {k_: float(np.median(v_)) for k_, v_ in stg.items()},
              "states_per_cycle": n5, "cost_edges_per_cycle": int(rows5.shape[0]), "dirty_rects_per_cycle": 3,
              "layers_updated_per_rect": 2, "cells_changed_per_cycle": 3 * side * side, "budget_ms_at_10hz": 100.0,
              "map_versions_seen": ctx5.map_version() - ver0,
              "stage_note": "the rectangle updates are asynchronous: `rects` is their host time, their device time is "
                            "inside the stage that synchronises next (`cnn`, a host-buffer call)",
              "sustained_states_per_s": n5 / (float(np.median(cyc)) * 1e-3)}
        if not args.no_cpu_baseline:  # the checker, outside the timed cycles
            import oracle_py as O
            g5 = type(gm5)(gm5.rows, gm5.cols, gm5.res, gm5.pos_x, gm5.pos_y)
            g5.add("upper_bound", ub5)
            g5.add("elevation_masked", mk5)
            m5 = 16384
            st5 = se5[:m5].cpu().numpy()
            ref5 = O.OracleMap(g5, body_layer="upper_bound").states_valid(O.robot("yaml"), st5)
            c5["labels_match_oracle_after_last_version"] = bool(np.array_equal(ref5, va5[:m5].cpu().numpy()))
            c5["labels_checked"] = m5
        gm5.preprocessed.close()
        ctx5.close()
        ctx.use_torch_stream()
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
        # the reference planners' own insertion orders on the same map and query (include/artp_c.h, construction):
        # 1 = PRMMotionCost::addValidMilestone with the reference's budgets (10 000 vertices -- chain vertices count --
        # / 50 000 edges; sequential host loop, one small device batch per milestone), 2 = LazyPRMStarMinUpdate's
        # predecessor-only graph (one device batch)
        for name, kw in (("prm_motion_cost_order", dict(n_milestones=10_000, max_n_edges=50_000, construction=1)),
                         ("lazy_prm_star_order_10000", dict(n_milestones=10_000, construction=2))):
            # twice: the first build + solve of a kind grows the context's scratch buffers (a 2^18-motion validity batch
            # allocates ~1 GB of PoseRecs and queues, anything between 1 and 300 ms); a planner that replans at 10 Hz is
            # in the second state
            for attempt in ("first", "steady"):
                t0 = time.perf_counter()
                rm = Roadmap(ctx, s_state, g_state, seed=seed, **kw)
                t1 = time.perf_counter()
                path, cost, rep = rm.solve()
                t2 = time.perf_counter()
                st = rm.stats()
                rm.close()
                if attempt == "first":
                    first = {"build_ms": (t1 - t0) * 1e3, "solve_ms": (t2 - t1) * 1e3}
            roadmap[name] = {"build_ms": (t1 - t0) * 1e3, "solve_ms": (t2 - t1) * 1e3, "first_run": first,
                             "vertices": int(st["vertices"]),
                             "edges": int(st["candidate_edges"]), "samples_drawn": int(st["samples_drawn"]),
                             "path_states": None if path is None else int(len(path)), "path_cost_s": cost,
                             "lazy_removals": rep,
                             "solve": "shortest-path tree hung from the far end, repaired per removal + motion verdicts "
                                      "of the informed set in one batch after 3 removals (roadmap.h roadmap_solve_tree, "
                                      "DESIGN 4.5); round 3: an A* and a device call per removal"}
    except Exception as ex:  # pragma: no cover
        roadmap = {"error": repr(ex)}

    # ---- N2 extras: the per-map preprocessing chain on the device ------------------------------------------
    preprocess = None
    try:
        if args.skip_extras:
            raise RuntimeError("skipped (--skip-extras)")
        # three kinds of install: a map the context has not seen (all tables built: alternating between two different
        # maps), the same map with a 40 x 40 patch changed (the install diffs the height fields on the device and takes
        # the rectangle path), the identical map again (nothing to do but the sampler layers and the diff itself)
        pre_ms, inst_new, inst_patch, inst_same = [], [], [], []
        other = raw_map(args.map, args.res, seed=4321)
        e_a, e_b = gm["elevation"], other["elevation"]
        e_p = e_a.copy()
        e_p[180:220, 150:190] += np.float32(0.05)

        def one(elev, trav, sink):
            t0 = time.perf_counter()
            pp = ctx.preprocess_map(elev, gm.len_x, gm.len_y, gm.pos_x, gm.pos_y, traversability=trav)
            t1 = time.perf_counter()
            pp.install()
            ctx.synchronize()
            t2 = time.perf_counter()
            pp.close()
            pre_ms.append((t1 - t0) * 1e3)
            sink.append((t2 - t1) * 1e3)

        for r_ in range(5):
            one(e_b if r_ % 2 == 0 else e_a, other["traversability"] if r_ % 2 == 0 else gm["traversability"], inst_new)
        one(e_a, gm["traversability"], [])
        for r_ in range(5):
            one(e_p if r_ % 2 == 0 else e_a, gm["traversability"], inst_patch)
        one(e_a, gm["traversability"], [])
        for r_ in range(5):
            one(e_a, gm["traversability"], inst_same)
        preprocess = {"preprocess_ms_incl_h2d": float(np.median(pre_ms)),
                      "install_ms_incl_tables": float(np.median(inst_new)),
                      "reinstall_ms_40x40_patch_changed": float(np.median(inst_patch)),
                      "reinstall_ms_identical_map": float(np.median(inst_same)), "map": f"{gm.rows}x{gm.cols}"}
    except Exception as ex:  # pragma: no cover
        preprocess = {"error": repr(ex)}

    # ---- per-call latency of the OMPL seams (one state / one edge per host call), driven from Python ------------
    per_call = None
    try:
        if args.skip_extras or multi:
            raise RuntimeError("skipped")
        cl = Context(local_rank, "yaml")
        cl.upload_map(gm)
        st_l = cl.sample_states(seed, 0, 8192)
        lab_l = cl.validate_states(st_l)
        acc_l = st_l[lab_l != 0]
        acc_l = acc_l[np.argsort(acc_l[:, 0])]
        a_l, b_l = acc_l[:-1], acc_l[1:]
        keep_l = np.hypot(a_l[:, 0] - b_l[:, 0], a_l[:, 1] - b_l[:, 1]) < 2.0
        a_l, b_l = np.ascontiguousarray(a_l[keep_l][:256]), np.ascontiguousarray(b_l[keep_l][:256])
        want_l = cl.check_motions(a_l, b_l)

        def per_call_us(fn, reps):
            for r in range(20):
                fn(r)
            t0 = time.perf_counter()
            for r in range(reps):
                fn(r)
            return (time.perf_counter() - t0) / reps * 1e6

        bad_l = [0]

        def one_state(r):
            i = r % 4096
            bad_l[0] += int(cl.validate_states(st_l[i:i + 1])[0] != lab_l[i])

        def one_edge(r):
            i = r % len(a_l)
            bad_l[0] += int(cl.check_motions(a_l[i:i + 1], b_l[i:i + 1])[0] != want_l[i])

        per_call = {"how": "host-buffer C ABI through ctypes (≈3 us of Python per call included); edges between accepted states "
                           "< 2 m apart; test_host.cpp measures the same seams from C++ (profiles/*_host_latency.json)",
                    "is_valid_1_state_us": per_call_us(one_state, 1000), "check_motion_1_edge_us": per_call_us(one_edge, 1000)}
        cl.set_persistent_latency(True)
        per_call["is_valid_1_state_us_resident"] = per_call_us(one_state, 2000)
        per_call["check_motion_1_edge_us_resident"] = per_call_us(one_edge, 2000)
        cl.set_persistent_latency(False)
        per_call["label_mismatches"] = bad_l[0]
        cl.close()
    except Exception as ex:  # pragma: no cover
        per_call = {"error": repr(ex)}

    # ---- C4 extras: 800 x 800 @ 0.04 m map, Params-default robot ---------------------------------------------
    c4 = None
    try:
        if args.skip_extras:
            raise RuntimeError("skipped (--skip-extras)")
        ctx4 = Context(local_rank, "defaults")
        gm4 = map_from_device(ctx4, raw_map(800, 0.04, seed=77))
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
              "map": "800x800@0.04", "robot": "Params defaults",
              "motion_cost_cnn": None if not motion_cost else motion_cost.get("c4_800")}
        gm4.preprocessed.close()
        ctx4.close()
        # secondary robot of SURVEY.md 8d on the C2 map
        ctx2d = Context(local_rank, "defaults")
        gm2d = map_from_device(ctx2d, raw_map(args.map, args.res, seed=1234))
        ctx2d.use_torch_stream()
        ctx2d.sample_and_validate_dev(seed, 0, S, se3, valid)
        ev0.record()
        for i4 in range(3):
            ctx2d.sample_and_validate_dev(seed, (i4 + 1) * S, S, se3, valid)
        ev1.record()
        torch.cuda.synchronize()
        c4["c2_map_defaults_robot"] = {"states_per_s": S / (ev0.elapsed_time(ev1) / 3 * 1e-3),
                                       "valid_frac": float(valid.float().mean().item())}
        gm2d.preprocessed.close()
        ctx2d.close()
        ctx.use_torch_stream()
    except Exception as ex:  # pragma: no cover
        c4 = {"error": repr(ex)}

    cpu = None
    if N == 1 and not args.no_cpu_baseline:
        cpu, cpu_labels, _ = cpu_baseline(gm, states)
        n_cpu = len(cpu_labels)
        cpu["labels_match_gpu"] = bool(np.array_equal(cpu_labels, labels[:n_cpu]))
        # the real patched ODE (kind "reference") when oracle/_ref travelled here: one thread and the sweep's best count
        try:
            import oracle_py as O
            if O.have_ref():
                aff_ = cpu["cores_how"]["sched_affinity"]
                tcs = sorted({t for t in (1, 4, 8, 16, 32, cpu["threads_at_best"]) if t <= max(aff_, 1)})
                ref = reference_ode_rates(gm, states, labels, tcs)   # 2e5 states at every thread count
                cpu["reference_ode"] = ref
                cpu["reference_ode_states"] = ref["states"]
                cpu["reference_ode_single_core_states_per_s"] = ref["threads"]["1"]
                cpu["reference_ode_best_states_per_s"] = max(ref["threads"].values())
                cpu["reference_ode_threads_at_best"] = int(max(ref["threads"], key=ref["threads"].get))
                cpu["reference_ode_labels_match_gpu"] = ref["labels_match_gpu"]
        except Exception as e:  # pragma: no cover
            cpu["reference_ode_error"] = repr(e)
        # edges/s of the CPU path on a bounded sample of the same edges (single thread)
        try:
            import oracle_py as O
            if E > 0:
                rob, om = O.robot("yaml"), O.OracleMap(gm)
                m_e = min(E, 1500)
                t0 = time.perf_counter()
                okm, _ = om.check_motions(rob, a[:m_e], b[:m_e])
                t1 = time.perf_counter()
                oki, _ = om.edges_interp_valid(rob, a[:m_e], b[:m_e])
                t2 = time.perf_counter()
                g_cm = ctx.check_motions(a[:m_e], b[:m_e])
                g_ci, _ = ctx.check_edges_interp(a[:m_e], b[:m_e])
                cpu["edges"] = {"sample": f"first {m_e} bench edges, single thread",
                                "check_motion_edges_per_s": m_e / (t1 - t0), "interp_0p5m_edges_per_s": m_e / (t2 - t1),
                                "verdicts_match_gpu": bool(np.array_equal(okm, g_cm) and np.array_equal(oki, g_ci))}
        except Exception as e:  # pragma: no cover
            cpu["edges"] = {"error": repr(e)}
        try:
            cpu["c1"] = c1_leg(local_rank)
        except Exception as e:  # pragma: no cover
            cpu["c1"] = {"error": repr(e)}

    # a roofline fraction for every kernel that has one of its own
    kernel_rooflines = {
        "sampler": {"bound": "hbm", "bytes_per_batch": 56.0 * S, "ms": sample_ms,
                    "achieved_GBps": 56.0 * S / (sample_ms * 1e-3) / 1e9, "frac": 56.0 * S / (sample_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "what": "artp_sample_states_dev alone: 7 f64 written per state (algorithmic bytes; the CDF tables stay in L2)"}}
    if motion_cost and "error" not in motion_cost:
        for Bq in (50000, 1 << 20):
            q = motion_cost.get(f"cost_queries_{Bq}")
            if q:
                kernel_rooflines[f"cost_query_{Bq}"] = {
                    "bound": "hbm", "bytes_per_batch": 132.0 * Bq, "ms": q["ms"], "achieved_GBps": 132.0 * Bq / (q["ms"] * 1e-3) / 1e9,
                    "frac": 132.0 * Bq / (q["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "what": "fc_cost_mfma_kernel: 24 B edge row + 96 B gathered features + 12 B costs per edge (SURVEY 8d)"}
        for tag in ("c3_400", "c4_800"):
            m_ = motion_cost.get(tag)
            if m_:
                kernel_rooflines[f"cnn_{tag}"] = {"bound": "mfma", "gflop": m_["cnn_gflop"], "ms": m_["cnn_kernels_ms"],
                                                  "achieved_TFLOPs": m_["cnn_kernels_tflops"], "peak_TFLOPs": MFMA_F16_PEAK_TFLOPS,
                                                  "frac": m_["cnn_kernels_frac_of_mfma_f16_peak"],
                                                  "frac_of_measured_mfma_rate": m_.get("cnn_kernels_frac_of_measured_mfma_rate"),
                                                  "measured_mfma_rate_TFLOPs": (motion_cost.get("mfma_probe") or {}).get("two_per_simd_tflops")}

    out = {
        "metric": "validated states/sec on 400x400@0.04m map (sample + validity check)",
        "value": value, "unit": "states/s", "n_gpus": N, "steps": K, "warmup": W,
        "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "C2: lazy_prm_star_min_update front end, 400x400@0.04m Perlin terrain "
                               "(seed 1234) + 12 obstacles, YAML robot, batch sampler + validity checker",
                   "states_per_gpu_per_step": S, "map": f"{args.map}x{args.map}@{args.res}",
                   # the regime of `value`: steady state, reached by untimed spin-up batches in front of the timed region
                   # (on top of the W warm-up steps); a 10 Hz replanner that idles between cycles sees the cold figure
                   "spin_up_batches": args.spin_up, "regime": "steady state (after the untimed spin-up batches)",
                   "ms_per_step_cold_after_2s_idle": cold_ms,
                   "lanes": f"{lanes} (the step's batch as {lanes} contiguous part(s) on {lanes} HIP stream(s) of one "
                            "context; roofline.kernel_ms is one part-free batch on one stream)",
                   "sharding": f"sample-index ranges over {N} GPU(s)" +
                               ((" [exchange: artp_group_* of the C ABI]" if use_grp is not None else " [exchange: torch.distributed]")
                                if do_gather else "") +
                               (", validity bitmaps (1 bit per candidate) all-gathered over RCCL + the first "
                                f"{'all' if args.materialise < 0 else args.materialise} accepted states of every rank per step "
                                "re-materialised on every rank" if do_gather else "")},
        # the metric's second half (BASELINE.json: "validated states/sec + edges/sec"): E edges per batch, HIP events
        # the figures a reader wants first, flat and near the front (the tail of a long line gets truncated in records)
        "headline": {
            "value_edges": edges.get("check_motion", {}).get("edges_per_s"),
            "value_edges_interp": edges.get("interp_0p5m", {}).get("edges_per_s"),
            "ms_per_step_cold_after_2s_idle": cold_ms,
            "roofline_bound": roofline["bound"], "roofline_frac": roofline["frac"],
            "valu_issue_occupancy": roofline["occupancy_fractions"].get("valu_issue") if roofline["occupancy_fractions"] else None,
            "valu_lane_util": roofline["valu_lane_util_time_weighted"],
            "hbm_traffic_frac": roofline["hbm_traffic_frac"],
            "cnn_c3_frac_of_mfma_f16_peak": ((motion_cost or {}).get("c3_400") or {}).get("cnn_kernels_frac_of_mfma_f16_peak"),
            "cnn_c4_frac_of_mfma_f16_peak": ((motion_cost or {}).get("c4_800") or {}).get("cnn_kernels_frac_of_mfma_f16_peak"),
            "cost_queries_per_s_2e20": ((motion_cost or {}).get(f"cost_queries_{1 << 20}") or {}).get("queries_per_s"),
            "c5_cycle_ms_median": (c5 or {}).get("cycle_ms_median"), "c5_sustained_states_per_s": (c5 or {}).get("sustained_states_per_s"),
            "cpu_port_states_per_s": (cpu or {}).get("value"), "cpu_cores": (cpu or {}).get("cores"),
            "cpu_threads_at_best": (cpu or {}).get("threads_at_best"),
            "cpu_reference_ode_best_states_per_s": (cpu or {}).get("reference_ode_best_states_per_s"),
            "labels_match_cpu": (cpu or {}).get("labels_match_gpu"),
        },
        "kernel_rooflines": kernel_rooflines,
        "metric_edges": "validated edges/sec on the same map: OMPL DiscreteMotionValidator::checkMotion (value_edges) and "
                        "the 0.5 m interpolation rule of PRMMotionCost::addValidMilestone (value_edges_interp)",
        "value_edges": edges.get("check_motion", {}).get("edges_per_s"),
        "value_edges_interp": edges.get("interp_0p5m", {}).get("edges_per_s"), "unit_edges": "edges/s",
        "roofline": roofline, "cpu_baseline": cpu,
        "valid_fraction": valid_frac, "label_hash_batch0": label_hash,
        "sampler_ms_per_batch": sample_ms, "edges": edges, "pipeline_counts_batch0": pipeline_counts,
        "motion_cost_c3": motion_cost, "replan_cycle_c5": c5, "roadmap_n1": roadmap, "preprocess_n2": preprocess,
        "c4_800_defaults": c4, "per_call_latency": per_call, "distributed": dist_extras,
        "device": ctx.arch, "gather_error": gather_error,
    }
    try:  # RCCL prints a version banner through C stdio; push it out BEFORE the JSON so that the line is the last one
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    if wd.fired:   # the watchdog is printing its own line: never two
        time.sleep(5)
        return
    wd.done()
    emit(out, partial=bool(args.no_pmc or args.skip_extras or args.no_cpu_baseline or N > 1))
    if dist is not None:
        wd.line_printed = True
        wd.stage("final barrier", args.watchdog)
        try:
            dist.barrier()
            if use_grp is not None:
                use_grp.close()
            dist.destroy_process_group()
        except Exception:  # pragma: no cover
            pass
        wd.done()


if __name__ == "__main__":
    main()
