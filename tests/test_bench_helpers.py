"""CPU suite: the host-side logic of bench.py that does not need a GPU -- the PMC summary arithmetic, the
staleness rule for the committed PMC fallback, the edge pairing and the CNN flop count of SURVEY.md 8a-R8."""
import json
import pytest
import os

import numpy as np

import common  # noqa: F401  (sys.path)
import bench


def test_cnn_flops_match_the_survey_figures():
    assert abs(bench.cnn_flops(400) / 1e9 - 37.66) < 0.02      # 37.66 GFLOP at C2
    assert 165 < bench.cnn_flops(800) / 1e9 < 175              # "~170 GFLOP" at C4
    # conv1 o conv2 composed: 0.068 + 1.626 GFLOP become 2 * 25 * 24 * 396^2 = 0.188 GFLOP at C2
    assert abs((bench.cnn_flops(400) - bench.cnn_flops_executed(400)) / 1e9 - (0.068 + 1.626 - 0.188)) < 0.01


def test_pair_edges_follow_the_8d_rule():
    rng = np.random.default_rng(0)
    acc = np.zeros((5000, 7))
    acc[:, :2] = rng.uniform(-8, 8, (5000, 2))
    ii, jj = bench.pair_edges(acc, 400)
    assert len(ii) == 400 and (jj > ii).all() and (jj - ii <= 3).all()
    assert (np.hypot(*(acc[ii, :2] - acc[jj, :2]).T) < 2.0).all()


def test_pmc_summary_arithmetic_and_corrections():
    cyc = 2.0e6  # per-XCD kernel cycles
    per_kernel = {
        "void artp::classify_states_kernel(...)": {
            "GRBM_GUI_ACTIVE": 8 * cyc, "SQ_ACTIVE_INST_VALU": 0.55 * bench.N_SIMD * cyc / 4,
            "SQ_ACTIVE_INST_LDS": 0.10 * bench.N_SIMD * cyc / 4, "SQ_INSTS_LDS": 1.0e6, "SQ_WAVE_CYCLES": 100.0,
            "SQ_WAIT_ANY": 70.0, "FETCH_SIZE": 1000.0, "WRITE_SIZE": 500.0, "TCC_HIT_sum": 80.0, "TCC_MISS_sum": 20.0,
            "max_us_by_pass": [900.0, 905.0, 910.0]},
        "void artp::some_other_kernel()": {"GRBM_GUI_ACTIVE": 1.0, "max_us_by_pass": [1.0]},
    }
    s = bench.summarise_pmc(per_kernel)
    k = s["kernels"]["classify_states_kernel"]
    assert abs(k["valu_busy"] - 0.55) < 1e-12 and abs(k["lds_busy"] - 0.10) < 1e-12
    assert k["hbm_fetch_bytes"] == 2 * 1024 * 1000.0          # gfx950: FETCH_SIZE doubled
    assert k["hbm_write_bytes"] == 1024 * 500.0
    assert abs(k["l2_hit"] - 0.8) < 1e-12 and abs(k["wait_frac"] - 0.7) < 1e-12 and k["us"] == 905.0
    assert s["validity_hbm_bytes_per_launch"] == 2 * 1024 * 1000.0 + 1024 * 500.0
    assert abs(s["valu_busy_time_weighted"] - 0.55) < 1e-12
    assert s["csrc_hash"] == bench.csrc_hash()


def test_committed_pmc_profile_is_only_used_for_the_same_kernel_sources(tmp_path, monkeypatch):
    """`roofline.traffic` must not silently go stale: the committed fallback is refused unless it
    was measured on exactly this checkout's art_planner_amd/csrc."""
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    os.makedirs(tmp_path / "profiles")
    os.makedirs(tmp_path / "art_planner_amd" / "csrc")
    (tmp_path / "art_planner_amd" / "csrc" / "k.h").write_text("// v1\n")
    h = bench.csrc_hash()
    json.dump({"csrc_hash": h, "validity_hbm_bytes_per_launch": 1.0}, open(tmp_path / "profiles" / "pmc_r03.json", "w"))
    d, note = bench.load_committed_pmc()
    assert d is not None and "same kernel sources" in note
    (tmp_path / "art_planner_amd" / "csrc" / "k.h").write_text("// v2\n")
    d, note = bench.load_committed_pmc()
    assert d is None and "STALE" in note


def test_bench_multi_gpu_launch_fails_with_a_clear_message_not_an_assertion():
    """`python bench.py --gpus N` spawns its own ranks.  On a box without N GPUs it must say so; under
    a launcher with another world size it must say how to start it."""
    import subprocess
    import sys
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(common.ROOT, "bench.py"), "--gpus", str(have + 2)], env=env,
                       capture_output=True, text=True)
    assert r.returncode != 0 and f"needs {have + 2} GPUs on this node, found {have}" in r.stderr, r.stderr
    assert "AssertionError" not in r.stderr and "Traceback" not in r.stderr
    r = subprocess.run([sys.executable, os.path.join(common.ROOT, "bench.py"), "--gpus", "4"],
                       env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr and "torch.distributed.run" in r.stderr
    assert "Traceback" not in r.stderr


def test_watchdog_prints_the_partial_line_when_a_multi_gpu_stage_hangs():
    """A multi-GPU run must not end without its JSON line: when a stage overruns -- a collective
    whose peer never arrived -- rank 0 prints the line from what was measured so far with `gather_error` naming the
    stage and exits 0; the other ranks leave without printing; a stage that finishes in time prints nothing."""
    import subprocess
    import sys
    prog = ("import sys, time; sys.path.insert(0, %r); sys.path.insert(0, %r); import bench\n"
            "wd = bench.Watchdog(int(sys.argv[1]))\n"
            "wd.stage('quick', 5.0); wd.done()\n"
            "wd.partial = {'metric': 'm', 'value': 123.0, 'headline_includes_exchange': False}\n"
            "wd.stage('bitmap all-gather', 0.6)\n"
            "time.sleep(30)\n" % (common.ROOT, os.path.join(common.ROOT, "tests")))
    r0 = subprocess.run([sys.executable, "-c", prog, "0"], capture_output=True, text=True, timeout=60)
    assert r0.returncode == 0, r0.stderr
    lines = [l for l in r0.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["value"] == 123.0 and "bitmap all-gather" in d["gather_error"] and d["headline_includes_exchange"] is False
    r1 = subprocess.run([sys.executable, "-c", prog, "1"], capture_output=True, text=True, timeout=60)
    assert r1.returncode == 3 and not [l for l in r1.stdout.splitlines() if l.startswith("{")]
    assert "bitmap all-gather" in r1.stderr


def test_the_reported_bound_is_derived_from_the_counters():
    """`roofline.bound` / `frac` name the LARGEST occupancy fraction the PMC passes measured (no literal),
    never the algorithmic-byte figure; without counters there is nothing to choose from."""
    assert bench.binding_fractions(None, None) == {}
    pmc = {"valu_busy_time_weighted": 0.86, "lds_busy_time_weighted": 0.04}
    fr = bench.binding_fractions(pmc, 0.10)
    assert max(fr, key=fr.get) == "valu_issue" and fr["hbm"] == 0.10
    fr = bench.binding_fractions({"valu_busy_time_weighted": 0.2, "lds_busy_time_weighted": 0.1}, 0.7)
    assert max(fr, key=fr.get) == "hbm"
    assert set(bench.binding_fractions(pmc, None)) == {"valu_issue", "lds"}


def test_host_core_count_is_the_affinity_mask_capped_by_the_cgroup_quota():
    """`cores` must be what the process may use, not the machine's logical CPU count."""
    assert bench.parse_cpu_max("max 100000\n") is None
    assert bench.parse_cpu_max("800000 100000") == 8.0
    assert bench.parse_cpu_max("150000 100000") == 1.5
    assert bench.parse_cpu_max("garbage") is None
    assert bench.host_cores(affinity=256, quota=8.0)[0] == 8
    assert bench.host_cores(affinity=256, quota=None)[0] == 256
    assert bench.host_cores(affinity=4, quota=8.0)[0] == 4
    assert bench.host_cores(affinity=16, quota=0.5)[0] == 1
    n, how = bench.host_cores()
    assert 1 <= n <= (os.cpu_count() or 1) and how["sched_affinity"] >= n
    assert bench.sweep_thread_counts(256) == [1, 8, 32, 64, 128, 256]
    assert bench.sweep_thread_counts(8) == [1, 8]
    assert bench.sweep_thread_counts(48) == [1, 8, 32, 48]
    assert bench.sweep_thread_counts(1) == [1]


def test_cgroup_quota_is_read_from_v2_and_v1_files(tmp_path):
    (tmp_path / "cpu.max").write_text("400000 100000\n")
    assert bench.cgroup_cpu_quota(str(tmp_path)) == 4.0
    (tmp_path / "cpu.max").write_text("max 100000\n")
    assert bench.cgroup_cpu_quota(str(tmp_path)) is None
    v1 = tmp_path / "v1"
    os.makedirs(v1 / "cpu")
    (v1 / "cpu" / "cpu.cfs_quota_us").write_text("250000\n")
    (v1 / "cpu" / "cpu.cfs_period_us").write_text("100000\n")
    assert bench.cgroup_cpu_quota(str(v1)) == 2.5
    (v1 / "cpu" / "cpu.cfs_quota_us").write_text("-1\n")
    assert bench.cgroup_cpu_quota(str(v1)) is None
    assert bench.cgroup_cpu_quota(str(tmp_path / "nowhere")) is None


def test_cpu_thread_legs_run_the_work_of_every_thread():
    hits = []
    dt = bench.run_threads(5, lambda k: hits.append(k))
    assert sorted(hits) == [0, 1, 2, 3, 4] and dt >= 0.0


def test_per_batch_pmc_figures_count_every_dispatch_of_a_batch():
    """a two-pass checkMotion launches each pipeline kernel twice per batch; the summary must carry
    both (it kept the largest dispatch only: 1.44 of 1.98 ms).  (setup + 4 batches) - (setup alone), per kernel, / 4."""
    setup = {"classify": {"n": 1, "sum_us": 400.0, "sums": {"GRBM_GUI_ACTIVE": 8e5, "SQ_ACTIVE_INST_VALU": 1e5}},
             "preproc": {"n": 7, "sum_us": 90.0, "sums": {"GRBM_GUI_ACTIVE": 1e5}}}
    full = {"classify": {"n": 1 + 4 * 2, "sum_us": 400.0 + 4 * (170.0 + 730.0),
                         "sums": {"GRBM_GUI_ACTIVE": 8e5 + 4 * (3e5 + 13e5), "SQ_ACTIVE_INST_VALU": 1e5 + 4 * 6e5}},
            "preproc": {"n": 7, "sum_us": 90.0, "sums": {"GRBM_GUI_ACTIVE": 1e5}},
            "expand": {"n": 8, "sum_us": 4 * (55.0 + 211.0), "sums": {"GRBM_GUI_ACTIVE": 4 * 5e5}}}
    pb = bench.per_batch_from_two_runs(full, setup, 4)
    assert set(pb) == {"classify", "expand"}                       # the setup's own kernels cancel
    assert pb["classify"]["dispatches_per_batch"] == 2.0 and abs(pb["classify"]["max_us"] - 900.0) < 1e-9
    assert abs(pb["classify"]["GRBM_GUI_ACTIVE"] - 16e5) < 1e-6 and abs(pb["classify"]["SQ_ACTIVE_INST_VALU"] - 6e5) < 1e-6
    assert abs(pb["expand"]["max_us"] - 266.0) < 1e-9


def test_lane_utilisation_and_the_roofline_fraction():
    """issue occupancy alone counts an instruction issued for 16 lanes like a full one."""
    assert bench.lane_utilisation(64.0 * 1000, 1000) == 1.0
    assert bench.lane_utilisation(16.0 * 1000, 1000) == 0.25
    assert bench.lane_utilisation(1.0, 0) is None
    cyc = 2.0e6
    per_kernel = {
        "void artp::feet_stream_kernel<4>(...)": {
            "GRBM_GUI_ACTIVE": 8 * cyc, "SQ_ACTIVE_INST_VALU": 1.0 * bench.N_SIMD * cyc / 4, "SQ_ACTIVE_INST_LDS": 0.0,
            "valu_lane_util": 0.5, "max_us_by_pass": [300.0]},
        "void artp::classify_states_kernel(...)": {
            "GRBM_GUI_ACTIVE": 8 * cyc, "SQ_ACTIVE_INST_VALU": 0.6 * bench.N_SIMD * cyc / 4, "SQ_ACTIVE_INST_LDS": 0.0,
            "valu_lane_util": 0.75, "max_us_by_pass": [100.0]},
    }
    s = bench.summarise_pmc(per_kernel)
    assert abs(s["kernels"]["feet_stream_kernel"]["valu_useful"] - 0.5) < 1e-12
    assert abs(s["kernels"]["classify_states_kernel"]["valu_useful"] - 0.45) < 1e-12
    assert abs(s["valu_busy_time_weighted"] - 0.9) < 1e-12
    assert abs(s["valu_lane_util_time_weighted"] - (0.5 * 300 + 0.75 * 100) / 400) < 1e-12
    assert abs(s["valu_useful_time_weighted"] - (0.5 * 300 + 0.45 * 100) / 400) < 1e-12
    fr = bench.binding_fractions(s, 0.1)
    assert max(fr, key=fr.get) == "valu_issue"
    assert abs(bench.roofline_fraction(s, "valu_issue", fr) - s["valu_useful_time_weighted"]) < 1e-12
    assert bench.roofline_fraction(s, "hbm", {"hbm": 0.7}) == 0.7
    assert bench.roofline_fraction(None, None, {}) is None


def test_mfma_clock_probe_reports_an_error_instead_of_failing_the_bench_without_a_gpu():
    """bench.mfma_clock_probe compiles tests/cpp/mfma_clock_probe.hip and runs it; where there is no device (this container) or no
    hipcc it returns {"error": ...} -- a reported figure is never a reason for the bench line to be missing."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the probe's figures are checked by tests/test_mfma_hazard.py")
    r = bench.mfma_clock_probe(timeout_s=120)
    assert set(r) == {"error"} and r["error"], r


def _canned_full_result(pad=1):
    """A full bench result of round 5's shape and worse: long prose, per-kernel tables, NaN / inf in the extras."""
    prose = "x" * 700
    per_kernel = {f"kernel_{i}": {"valu_busy": 0.7, "lds_busy": 0.1, "us": 400.0 + i, "note": prose} for i in range(12 * pad)}
    return {
        "metric": "validated states/sec on 400x400@0.04m map (sample + validity check)", "value": 3.667e9, "unit": "states/s",
        "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 1.1437, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "C2: " + prose, "states_per_gpu_per_step": 1 << 22, "map": "400x400@0.04", "lanes": prose,
                   "sharding": prose, "spin_up_batches": 32},
        "headline": {"value_edges": 1.33e8, "value_edges_interp": 2.0e8, "ms_per_step_cold_after_2s_idle": 1.264,
                     "roofline_frac": 0.5, "c5_cycle_ms_median": float("nan")},
        "kernel_rooflines": {f"k{i}": {"what": prose, "frac": 0.1} for i in range(8 * pad)},
        "value_edges": 1.33e8, "value_edges_interp": 2.0e8, "unit_edges": "edges/s",
        "roofline": {"bound": "valu_issue", "achieved": 0.85, "peak": 1.0, "unit": prose, "frac": 0.52, "frac_is": prose,
                     "traffic": 8.75e8, "hbm_traffic_frac": 0.0957, "kernel": "validity pipeline", "kernel_ms": 1.12,
                     "occupancy_fractions": {"valu_issue": 0.85, "lds": 0.2, "hbm": 0.1}, "note": prose,
                     "algorithmic_hbm": {"achieved": 20000.0, "peak": 8000.0, "unit": "GB/s", "ratio_to_peak": 2.5,
                                         "bytes_per_launch": 2.292e10, "bytes_per_state": 5464.0, "note": prose},
                     "binding": {"per_kernel": per_kernel, "valu_busy_note": prose}, "pmc_source": "live",
                     "csrc_hash": "f06808b994e8a4a7"},
        "cpu_baseline": {"value": 9.66e5, "unit": "states/s", "cores": 16, "kind": "port", "sample": prose,
                         "threads_at_best": 32, "single_core_value": 6.0e4, "labels_match_gpu": True,
                         "thread_sweep": [{"threads": t, "states_per_s": 1e5 * t, "note": prose} for t in range(1, 40 * pad)],
                         "reference_ode": {"threads": {str(t): 5e4 * t for t in range(64)}},
                         "reference_ode_single_core_states_per_s": 5.5e4, "reference_ode_best_states_per_s": 2.9e5,
                         "reference_ode_labels_match_gpu": True,
                         "edges": {"check_motion_edges_per_s": 6900.0, "verdicts_match_gpu": True, "sample": prose},
                         "c1": {"cpu_lazy_prm_star": {"x": [prose] * 20}}},
        "valid_fraction": 0.31, "label_hash_batch0": "0123456789abcdef", "sampler_ms_per_batch": 0.05,
        "edges": {"check_motion": {"binding": {"per_kernel": per_kernel}}}, "pipeline_counts_batch0": list(range(400)),
        "motion_cost_c3": {"c3_400": {"cnn_kernels_ms": float("inf")}, "blob": prose * 5},
        "replan_cycle_c5": {"cycles": [1.0] * 300}, "roadmap_n1": {"a": prose}, "preprocess_n2": {"a": prose},
        "c4_800_defaults": {"a": prose}, "distributed": None, "device": "gfx950:sramecc+:xnack-", "gather_error": None,
    }


@pytest.mark.parametrize("pad", [1, 20])
def test_the_contract_line_is_one_short_line_whatever_the_extras_hold(pad):
    """round 5's final line grew to 25.5 kB and the driver's record parsed nothing.  The last line
    bench.py prints carries the contract keys + config + headline + roofline + cpu_baseline and nothing bulky."""
    full = _canned_full_result(pad)
    line = bench.contract_line(full)
    assert "\n" not in line and len(line.encode()) < 6144 < 8192
    for tok in ("NaN", "Infinity"):
        assert tok not in line
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["value"] == full["value"] and d["ms_per_step"] == full["ms_per_step"] and d["config"]["states_per_gpu_per_step"] == 1 << 22
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d["roofline"], k
    assert d["roofline"]["algorithmic_hbm"]["ratio_to_peak"] == 2.5 and "binding" not in d["roofline"]
    for k in ("value", "unit", "cores", "kind", "sample", "reference_ode_best_states_per_s", "labels_match_gpu"):
        assert k in d["cpu_baseline"], k
    assert "thread_sweep" not in d["cpu_baseline"] and "c1" not in d["cpu_baseline"]
    assert d["headline"]["c5_cycle_ms_median"] is None        # NaN became null
    for bulky in ("kernel_rooflines", "edges", "motion_cost_c3", "replan_cycle_c5", "roadmap_n1", "pipeline_counts_batch0"):
        assert bulky not in d


def test_emit_prints_the_detail_line_first_and_the_contract_line_last(tmp_path, monkeypatch):
    import io
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    buf = io.StringIO()
    bench.emit(_canned_full_result(), buf)
    lines = buf.getvalue().splitlines()
    assert len(lines) == 2 and lines[0].startswith("BENCH_DETAIL ")
    detail = json.loads(lines[0][len("BENCH_DETAIL "):])
    assert "kernel_rooflines" in detail and "roadmap_n1" in detail
    last = json.loads(lines[-1])
    assert last["metric"].startswith("validated states/sec") and len(lines[-1]) < 6144
    assert json.load(open(tmp_path / "gpurun_out" / "bench_detail.json"))["value"] == last["value"]


def test_watchdog_partial_line_is_short_too():
    part = {"metric": "m", "value": 1.0, "unit": "states/s", "n_gpus": 2, "distributed": {"exchange": "x" * 5000, "blob": ["y" * 100] * 100}}
    line = bench.contract_line(part)
    assert len(line) < 6144 and json.loads(line)["value"] == 1.0
