"""The C++ host mirror (art_planner_amd/host: reference class names/signatures over the C ABI)."""
import os
import sys
import struct
import subprocess

import numpy as np
import pytest

import common
import golden_io

HOST = os.path.join(common.ROOT, "art_planner_amd", "host")
BIN = os.path.join(HOST, "test_host")
BIN_PLANNER = os.path.join(HOST, "test_planner")
BIN_GROUP = os.path.join(HOST, "test_group")


def _build():
    subprocess.check_call(["make", "-s", "-C", HOST])
    assert os.path.exists(BIN) and os.path.exists(BIN_PLANNER) and os.path.exists(BIN_GROUP)


def test_host_mirror_builds_and_refuses_without_gpu():
    _build()
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    r = subprocess.run([BIN], capture_output=True, text=True)
    assert r.returncode == 3, r.stdout + r.stderr   # context creation throws: no CPU fallback
    r = subprocess.run([BIN_PLANNER], capture_output=True, text=True)
    assert r.returncode == 3, r.stdout + r.stderr
    r = subprocess.run([BIN_GROUP], capture_output=True, text=True)   # artp_group_create refuses too
    assert r.returncode == 3, r.stdout + r.stderr


def test_shard_first_index_is_the_same_function_on_both_sides_of_the_abi():
    from art_planner_amd import _capi
    from art_planner_amd.distributed import shard_first_index
    L = _capi.load()
    for step, rank, world, batch in ((0, 0, 1, 1), (3, 2, 8, 1 << 22), (1000003, 7, 8, 4194304), (5, 15, 16, 12345)):
        assert L.artp_shard_first_index(step, rank, world, batch) == shard_first_index(step, rank, world, batch)


def _write_map_part(f, gm, zb):
    hack = np.zeros((gm.rows, gm.cols), np.float32, order="F")
    hack[:, 0] = gm["cum_prob_rowwise"]
    f.write(struct.pack("<ii", gm.rows, gm.cols))
    f.write(struct.pack("<dddd", gm.len_x, gm.len_y, gm.pos_x, gm.pos_y))
    for layer in (gm["elevation"], gm["elevation_masked"], gm["cum_prob"], hack, gm["normal_x"], gm["normal_y"],
                  gm["normal_z"], gm["plane_fit_std_dev"]):
        f.write(np.asfortranarray(layer, np.float32).tobytes(order="F"))
    f.write(struct.pack("<dd", *zb))


@pytest.mark.gpu
def test_device_group_through_the_c_abi(tmp_path):
    """test_group.cpp: artp_group_* (include/artp_c.h "multi-GPU") from a C++ host, no interpreter and no torch in the
    process: an RCCL group over every visible GPU and a three-rank peer-copy group on device 0 -- gathered bitmaps,
    accepted counts, re-materialised states and edge records of EVERY rank bit-identical to what one plain context
    computes for that rank's shard; steps in flight without host waits; the group's throughput as a C++ host sees it
    (gpurun_out/group_test.json)."""
    _build()
    import oracle_py as O
    from synthetic import make_map
    gm = make_map(160, 0.04, seed=7)
    rob = O.robot("yaml")
    elev = gm["elevation"]
    fin = elev[np.isfinite(elev)]
    zb = (float(fin.min()) - rob.reach_z / 2, float(fin.max()) + rob.reach_z / 2)
    path = tmp_path / "map.bin"
    with open(path, "wb") as f:
        _write_map_part(f, gm, zb)
    out_dir = os.path.join(common.ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    r = subprocess.run([BIN_GROUP, str(path), os.path.join(out_dir, "group_test.json")], capture_output=True,
                       text=True, timeout=600)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "test_group ok" in r.stdout
    assert "art_planner::DeviceGroup (class): ok" in r.stdout   # the same through host/include/art_planner/device_group.h


def _build_loopback_rccl(out_dir):
    """tests/cpp/loopback_rccl.cpp -> libloopback_rccl.so (g++; a TEST DOUBLE for librccl.so, see its header)."""
    so = os.path.join(str(out_dir), "libloopback_rccl.so")
    subprocess.check_call(["g++", "-O2", "-fPIC", "-shared", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-o", so,
                           os.path.join(common.ROOT, "tests", "cpp", "loopback_rccl.cpp"), "-L/opt/rocm/lib", "-lamdhip64", "-lrt"])
    return so


def test_loopback_rccl_double_builds_and_exports_what_group_h_binds(tmp_path):
    import ctypes as C
    so = _build_loopback_rccl(tmp_path)
    L = C.CDLL(so)
    for name in ("ncclGetUniqueId", "ncclCommInitRank", "ncclCommInitAll", "ncclCommDestroy", "ncclCommAbort", "ncclCommGetAsyncError",
                 "ncclAllGather", "ncclAllReduce", "ncclGroupStart", "ncclGroupEnd", "ncclGetErrorString"):
        assert hasattr(L, name), name     # csrc/group.h ARTP_RCCL_SYM: a missing one disables the binding


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_bench_n_ranks_rehearsal_on_one_gpu(tmp_path, world):
    """bench.py's N > 1 path end to end on a ONE-GPU box, launched the way the driver launches it (torch.distributed.run, one
    process per rank): the RCCL id made by rank 0 and broadcast, artp_group_create_rank in every rank, the bitmap all-gather
    and re-materialisation inside the timed steps, the barrier + max-over-ranks clock, ONE contract line from rank 0 with
    world_size = rccl_ranks_seen = N.  --rehearse-on-one-gpu: every rank uses GPU 0, torch.distributed over gloo, the group
    over the loopback test double (RCCL refuses two ranks on a GPU) -- a rehearsal of the code path, not a measurement, and
    the line says so."""
    import json
    import socket
    so = _build_loopback_rccl(tmp_path)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, ARTP_RCCL_LIB=so, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(common.ROOT, "bench.py"), "--gpus", str(world), "--steps", "4", "--warmup", "2",
           "--batch", str(1 << 20), "--rehearse-on-one-gpu"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=common.ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = r.stdout.strip().splitlines()[-1]
    d = json.loads(line)
    assert len(line) <= 6144
    assert d["n_gpus"] == world and d["steps"] == 4 and d["scaling"] == "weak" and d["value"] > 0
    dd = d["distributed"]
    assert dd["world_size"] == world and dd["rccl_ranks_seen"] == world and dd["headline_includes_exchange"] is True
    assert "artp_group" in dd["exchange"] and "NOT a measurement" in dd["rehearsal"]
    assert d.get("gather_error") is None
    out_dir = os.path.join(common.ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, f"bench_rehearsal_{world}_ranks_one_gpu.json"), "w") as f:
        f.write(line + "\n")


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_device_group_one_process_per_rank_on_one_gpu(tmp_path, world):
    """artp_group_create_rank with W > 1 -- W PROCESSES, the id made by rank 0 and handed to the others
    through a file, each rank's own communicator, the all-gathers of the bitmap / edge blocks at the rank offsets, steps in
    flight on the double buffers -- on ONE GPU.  The real RCCL refuses two ranks on a device ("Duplicate GPU detected",
    scripts/rccl_same_gpu_probe.py), so $ARTP_RCCL_LIB points libartp.so at tests/cpp/loopback_rccl.cpp, a test double
    behind the same ncclAllGather / ncclAllReduce call sites (staging buffers shared by hipIpcMemHandle).  Every process
    compares ALL W ranks' blocks with what a plain context computes for those shards (test_group.cpp exercise_group).
    What this does NOT execute: ncclCommInitRank of the real library across devices."""
    _build()
    so = _build_loopback_rccl(tmp_path)
    import oracle_py as O
    from synthetic import make_map
    gm = make_map(160, 0.04, seed=7)
    rob = O.robot("yaml")
    fin = gm["elevation"][np.isfinite(gm["elevation"])]
    zb = (float(fin.min()) - rob.reach_z / 2, float(fin.max()) + rob.reach_z / 2)
    path = tmp_path / "map.bin"
    with open(path, "wb") as f:
        _write_map_part(f, gm, zb)
    env = dict(os.environ, ARTP_RCCL_LIB=so, HSA_ENABLE_IPC_MODE_LEGACY="0")
    id_file = str(tmp_path / f"id_{world}.bin")
    procs = [subprocess.Popen([BIN_GROUP, str(path), "--rank", str(r), str(world), id_file, str(tmp_path / f"rank{r}.json")],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o)
    for r, (p, o) in enumerate(zip(procs, outs)):
        print(o)
        assert p.returncode == 0 and f"test_group rank {r} of {world} ok" in o, o
    out_dir = os.path.join(common.ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, f"group_ranks_{world}.json"), "w") as f:
        f.write("[" + ", ".join(open(tmp_path / f"rank{r}.json").read().strip() for r in range(world)) + "]\n")


def test_real_library_branches_compile():
    """The preprocessor branches INTEGRATION.md tells a maintainer to build -- ARTP_HAVE_OMPL (`#include
    <ompl/base/...>`) and ARTP_HAVE_EIGEN (`EdgeMatrix` = the Eigen row-major matrix) -- through a compiler:
    tests/fake_include presents the stand-ins under the real include names (neither library is in this image).
    Catches scope / namespace / missing-include regressions in exactly that branch (round 1's Eigen include bug)."""
    fake = os.path.join(common.ROOT, "tests", "fake_include")
    for src in ("test_host.cpp", "test_planner.cpp"):
        r = subprocess.run(["g++", "-std=c++14", "-Wall", "-Werror", "-fsyntax-only", "-DARTP_HAVE_OMPL", "-DARTP_HAVE_EIGEN",
                            "-I" + fake, "-I" + os.path.join(HOST, "include"), "-I" + os.path.join(common.ROOT, "include"),
                            os.path.join(HOST, src)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    # with grid_map too, Planner carries the reference's exact signatures and protected members (planner.h:41-70): a
    # PlannerRos-shaped subclass (`class ... : protected Planner`, planner_ros.h:24) must build against them -- and the
    # plain mirror programs must keep building in that configuration
    for src in ("test_planner_ros_shape.cpp", "test_planner.cpp", "test_host.cpp"):
        r = subprocess.run(["g++", "-std=c++14", "-Wall", "-Werror", "-fsyntax-only", "-DARTP_HAVE_OMPL", "-DARTP_HAVE_EIGEN",
                            "-DARTP_HAVE_GRID_MAP", "-I" + fake, "-I" + os.path.join(HOST, "include"),
                            "-I" + os.path.join(common.ROOT, "include"), os.path.join(HOST, src)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    r = subprocess.run(["g++", "-std=c++14", "-fsyntax-only", "-DARTP_HAVE_OMPL", "-DARTP_HAVE_EIGEN", "-I" + fake,
                        "-I" + os.path.join(HOST, "include"), "-I" + os.path.join(common.ROOT, "include"),
                        os.path.join(HOST, "test_planner_ros_shape.cpp")], capture_output=True, text=True)
    assert r.returncode != 0 and "ARTP_HAVE_GRID_MAP" in r.stderr     # the exact surface needs both libraries
    # and the branch really is taken: without the fake tree the same flags must fail on the missing headers
    r = subprocess.run(["g++", "-std=c++14", "-fsyntax-only", "-DARTP_HAVE_OMPL", "-DARTP_HAVE_EIGEN",
                        "-I" + os.path.join(HOST, "include"), "-I" + os.path.join(common.ROOT, "include"),
                        os.path.join(HOST, "test_host.cpp")], capture_output=True, text=True)
    assert r.returncode != 0 and ("ompl/" in r.stderr or "Eigen/" in r.stderr)


@pytest.mark.gpu
def test_host_mirror_matches_oracle_through_the_ompl_shaped_interfaces(tmp_path):
    """test_host.cpp, compiled against the strict OMPL-1.4.2-shaped stand-ins: isValid (batch, arbitrary single
    states through the latency path, sampler-issued states through the label lookup), both checkMotion overloads
    incl. the lastValid pair, sampleUniformNear / sampleGaussian -- labels, verdicts and lastValid against the
    CPU oracle.  Also records the per-call latencies INTEGRATION.md quotes (gpurun_out/host_latency.json)."""
    _build()
    import oracle_py as O
    from art_planner_amd.context import Context
    from synthetic import make_map
    gm = make_map(160, 0.04, seed=7)
    rob = O.robot("yaml")
    om = O.OracleMap(gm)
    ctx = Context(0, "yaml")
    ctx.upload_map(gm)
    se3 = ctx.sample_states(11, 0, 6000)
    # stale-label case: the states a fresh sampler mirror with seed 4242 will issue, their oracle
    # labels on this map and on the map after a rectangle of the BODY layer was raised by 0.5 m
    nb, seed_b = 2048, 4242
    blk = ctx.sample_states(seed_b, 0, nb)
    ctx.close()
    import copy
    r0, c0, nr, nc = 30, 35, 90, 80
    gm_new = copy.copy(gm)
    gm_new.layers = dict(gm.layers)
    raised = gm["elevation"].copy(order="F")
    raised[r0:r0 + nr, c0:c0 + nc] += np.float32(0.5)
    gm_new.layers["elevation"] = raised
    old_labels = om.states_valid(rob, blk)
    new_labels = O.OracleMap(gm_new).states_valid(rob, blk)
    assert (old_labels != new_labels).sum() > 40
    se3[::7, 2] += 0.3                      # some states off the terrain
    expected = om.states_valid(rob, se3)
    acc = se3[expected != 0]
    assert 200 < len(acc) < len(se3)
    # motions: accepted state i -> accepted state i+1 (any length), plus some that end on an invalid state
    m = 300
    s1, s2 = acc[:m].copy(), acc[1:m + 1].copy()
    # every other motion PRM-sized: to one of the three nearest accepted states (the reference connects milestones to their
    # nearest neighbours; test_host.cpp times the motions shorter than 2 m apart from the map-spanning ones)
    dd = np.hypot(s1[:, None, 0] - acc[None, :, 0], s1[:, None, 1] - acc[None, :, 1])
    dd[np.arange(m), np.arange(m)] = np.inf
    near = acc[np.argsort(dd, axis=1)[np.arange(m), np.random.default_rng(3).integers(0, 3, m)]]
    s2[1::2] = near[1::2]
    s2[::9] = se3[expected == 0][:len(s2[::9])]
    ok, last_t, last_state = om.check_motions_last_valid(rob, s1, s2)
    ok0, _ = om.check_motions(rob, s1, s2)
    assert np.array_equal(ok, ok0)          # both overloads of the restated validator agree
    assert 20 < ok.sum() < m - 20
    elev = gm["elevation"]
    fin = elev[np.isfinite(elev)]
    zb = (float(fin.min()) - rob.reach_z / 2, float(fin.max()) + rob.reach_z / 2)
    path = tmp_path / "fixture.bin"
    hack = np.zeros((gm.rows, gm.cols), np.float32, order="F")
    hack[:, 0] = gm["cum_prob_rowwise"]
    with open(path, "wb") as f:
        f.write(struct.pack("<ii", gm.rows, gm.cols))
        f.write(struct.pack("<dddd", gm.len_x, gm.len_y, gm.pos_x, gm.pos_y))
        for layer in (gm["elevation"], gm["elevation_masked"], gm["cum_prob"], hack, gm["normal_x"], gm["normal_y"],
                      gm["normal_z"], gm["plane_fit_std_dev"]):
            f.write(np.asfortranarray(layer, np.float32).tobytes(order="F"))
        f.write(struct.pack("<dd", *zb))
        f.write(struct.pack("<i", len(se3)))
        f.write(np.ascontiguousarray(se3, np.float64).tobytes())
        f.write(np.ascontiguousarray(expected, np.uint8).tobytes())
        f.write(struct.pack("<i", m))
        f.write(np.ascontiguousarray(s1, np.float64).tobytes())
        f.write(np.ascontiguousarray(s2, np.float64).tobytes())
        f.write(np.ascontiguousarray(ok, np.uint8).tobytes())
        f.write(np.ascontiguousarray(last_t, np.float64).tobytes())
        f.write(np.ascontiguousarray(last_state, np.float64).tobytes())
        f.write(struct.pack("<iQ", nb, seed_b))
        f.write(np.ascontiguousarray(old_labels, np.uint8).tobytes())
        f.write(np.ascontiguousarray(new_labels, np.uint8).tobytes())
        f.write(struct.pack("<iiii", r0, c0, nr, nc))
        f.write(np.asfortranarray(raised[r0:r0 + nr, c0:c0 + nc], np.float32).tobytes(order="F"))
    out_dir = os.path.join(common.ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    lat = os.path.join(out_dir, "host_latency.json")
    r = subprocess.run([BIN, str(path), lat], capture_output=True, text=True)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr
    import json
    t = json.load(open(lat))
    # sampler-issued states amortise below 1 us per sampleUniform + isValid, an arbitrary single
    # state costs less than 25 us (one launch through mapped host memory)
    assert t["sampler_loop_us_per_state"] < 1.0, t
    assert t["isvalid_arbitrary_state_us"] < 25.0, t
    # one checkMotion call = one launch (check_motions_few_kernel), not the batch pipeline; the CPU
    # oracle's per-edge time on the same edges goes into the record next to it
    import time
    t0 = time.perf_counter()
    om.check_motions(rob, s1, s2)
    t["cpu_oracle_check_motion_us_per_edge"] = (time.perf_counter() - t0) / m * 1e6
    json.dump(t, open(lat, "w"))
    assert t["isvalid_arbitrary_state_us_persistent_service"] < t["isvalid_arbitrary_state_us"], t   # no launch per call
    assert t["check_motion_1_edge_us"] < 40.0, t
    assert t["check_motion_1_edge_us"] < t["check_motion_1_edge_us_batch_pipeline"], t


@pytest.mark.gpu
def test_planner_mirror_plans_on_a_perlin_map(tmp_path):
    """art_planner::Planner of the host mirror (setMap / plan / getSolutionPath, planner.h:31-71): NO_MAP, the
    ignored map without elevation layer, SOLVED with valid states and motions (original and simplified path),
    re-query on the kept roadmap after another setMap, INVALID_START / INVALID_GOAL."""
    _build()
    from art_planner_amd.context import Context
    from synthetic import make_map
    gm = make_map(250, 0.04, seed=77)
    # start / goal: two valid samples of the same map under the same preprocessing (Params defaults), far apart
    ctx = Context(0, "yaml")
    pm = ctx.preprocess_map(gm["elevation"], gm.len_x, gm.len_y, traversability=gm["traversability"], kind="defaults")
    pm.install()
    se3 = ctx.sample_states(5, 0, 4000)
    acc = se3[ctx.validate_states(se3) != 0]
    assert len(acc) > 100
    d = np.hypot(acc[:, None, 0] - acc[None, :200, 0], acc[:, None, 1] - acc[None, :200, 1])
    i, j = np.unravel_index(np.argmax(d), d.shape)
    assert d[i, j] > 5.0
    pm.close()
    ctx.close()
    path = tmp_path / "planner.bin"
    with open(path, "wb") as f:
        f.write(struct.pack("<ii", gm.rows, gm.cols))
        f.write(struct.pack("<dddd", gm.len_x, gm.len_y, gm.pos_x, gm.pos_y))
        f.write(np.asfortranarray(gm["elevation"], np.float32).tobytes(order="F"))
        f.write(np.asfortranarray(gm["traversability"], np.float32).tobytes(order="F"))
        f.write(np.ascontiguousarray(np.stack([acc[i], acc[j]]), np.float64).tobytes())
    r = subprocess.run([BIN_PLANNER, str(path)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    # the same query through the reference's EXACT signatures: a PlannerRos-shaped subclass -- protected
    # ss_ / space_ / checker_ / map_->getMap(), setMap(std::unique_ptr<grid_map::GridMap>&&), plan(ScopedState, ScopedState),
    # og::PathGeometric getSolutionPath -- compiled with -DARTP_HAVE_OMPL -DARTP_HAVE_EIGEN -DARTP_HAVE_GRID_MAP against the
    # scaffold tests/fake_include and RUN on the GPU: SOLVED, every path state and motion valid through the OMPL objects
    subprocess.check_call(["make", "-s", "-C", HOST, "test_planner_ros_shape"])
    r = subprocess.run([os.path.join(HOST, "test_planner_ros_shape"), str(path)], capture_output=True, text=True)
    print(r.stdout)
    assert r.returncode == 0 and "0 failed checks" in r.stdout, r.stdout + r.stderr
