"""Parity of the forms that were built, measured and did NOT become the default (out of the product):
they live in libartp_variants.so (make -C art_planner_amd/csrc variants: cost_kernels_variants.h, pipeline_variants.h, the
roadmap's search-per-round loop) behind environment switches that libartp.so does not read.  Every one of them must give the
default's answers -- that is what makes the measurements in profiles/ comparisons of like with like."""
import os
import sys

import numpy as np
import pytest

import common
import golden_io
import oracle_py as O

sys.path.insert(0, os.path.join(common.ROOT, "oracle"))
sys.path.insert(0, os.path.join(common.ROOT, "tools"))
import motion_cost_oracle as mo  # noqa: E402
import convert_weights  # noqa: E402
from art_planner_amd import _capi  # noqa: E402
from test_motion_cost import _assert_features_close, _gpu_features  # noqa: E402
from test_roadmap import _directional_cost  # noqa: E402


def _vctx(kind):
    from art_planner_amd.context import Context
    import subprocess   # always through make: a no-op when libartp_variants.so is current, a rebuild when a source changed
    subprocess.check_call(["make", "-s", "-C", os.path.dirname(_capi.VARIANTS_LIB_PATH), "variants"])
    return Context(0, kind, lib=_capi.VARIANTS_LIB_PATH)


def test_the_product_library_reads_three_environment_variables_and_carries_no_variant_kernels():
    """`strings libartp.so | grep ARTP_`: the RCCL library path, the group's configure timeout, the latency paths' polling
    switch -- and nothing that selects a kernel (<= 4 names)."""
    import re
    import subprocess
    subprocess.check_call(["make", "-s", "-C", os.path.dirname(_capi.LIB_PATH)])
    blob = open(_capi.LIB_PATH, "rb").read()
    names = set(m.decode() for m in re.findall(rb"ARTP_[A-Z0-9_]{3,}", blob))
    env = {n for n in names if not n.startswith(("ARTP_ERR", "ARTP_OK", "ARTP_WAVES", "ARTP_GROUP_PEER", "ARTP_GROUP_RCCL", "ARTP_INPAINT"))}
    assert env == {"ARTP_RCCL_LIB", "ARTP_GROUP_CONFIGURE_TIMEOUT_MS", "ARTP_NO_POLL"}, env
    assert len(names) <= 4, names
    for kernel in (b"conv_kwalk_kernel", b"conv12_pool_kernel", b"conv12_mfma_kernel", b"feet_stream2_kernel", b"conv15_pair32_kernel"):
        assert kernel not in blob, kernel


@pytest.mark.gpu
@pytest.mark.parametrize("name", golden_io.MAPS)
def test_dense_feet_stream_variant_gives_the_same_labels(name, big_map, monkeypatch):
    """$ARTP_FEET_DENSE=1 selects feet_stream2_kernel (pipeline_variants.h) (the corner stage's plane / contact arithmetic on dense lanes: round 5's
    lane-utilisation experiment, kept although it is no faster): same labels as the real ODE's on the bulk states and as
    the default kernel's on 2^19 sampler states of the C2 map."""
    gm, _, states, _ = golden_io.load_bulk(name)
    monkeypatch.setenv("ARTP_FEET_DENSE", "1")
    dense = {r: _vctx(r) for r in ("yaml", "defaults")}
    monkeypatch.delenv("ARTP_FEET_DENSE")
    for rname, ctx in dense.items():
        ctx.upload_map(gm, sampler=False)
        assert np.array_equal(ctx.validate_states(states[rname]["se3"]), states[rname]["valid"]), f"{name}/{rname}"
    if name == golden_io.MAPS[0]:
        from art_planner_amd.context import Context
        ref = Context(0, "yaml")                       # the product library's default kernel
        ref.upload_map(big_map)
        se3 = ref.sample_states(42, 0, 1 << 19)
        want = ref.validate_states(se3)
        dense["yaml"].upload_map(big_map)
        assert np.array_equal(dense["yaml"].validate_states(se3), want)
        ref.close()
    for ctx in dense.values():
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"ARTP_POOL_WGS": "5"}, {"ARTP_POOL_WGS": "64"}])
def test_resident_edge_pool_of_other_sizes(env, monkeypatch):
    """The resident edge pool with other numbers of workgroups ($ARTP_POOL_WGS, variants build: 5 = many rounds per edge and the
    cross-workgroup early exit on every edge; 64): the golden edges' verdicts, lastValid pairs and interpolation counts, one
    and two edges per call."""
    name = golden_io.MAPS[0]
    gm, _ = golden_io.load_boxes(name)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    for rname, e in golden_io.load_edges(name).items():
        ctx = _vctx(rname)
        ctx.upload_map(gm, sampler=False)
        m = min(len(e["s1"]), 400)
        s1, s2 = e["s1"][:m], e["s2"][:m]
        b_ok, b_t, b_st = ctx.check_motions_last_valid(s1, s2)
        ctx.set_persistent_latency(True)
        i, k, bad = 0, 1, 0
        while i < m:
            j = min(i + k, m)
            bad += int((ctx.check_motions(s1[i:j], s2[i:j]) != e["check_motion"][i:j]).sum())
            ok, t, st = ctx.check_motions_last_valid(s1[i:j], s2[i:j])
            bad += int((ok != b_ok[i:j]).sum()) + int((t != b_t[i:j]).sum())
            bad += int((~((st == b_st[i:j]) | (np.isnan(st) & np.isnan(b_st[i:j])))).any(axis=1).sum())
            oki, ni = ctx.check_edges_interp(s1[i:j], s2[i:j])
            bad += int((oki != e["interp_valid"][i:j]).sum()) + int((ni != e["n_interp"][i:j]).sum())
            i, k = j, k % 2 + 1
        assert bad == 0, f"{rname} {env}: {bad} mismatches"
        assert ctx.persistent_latency_stats()["requests"] > 0
        ctx.set_persistent_latency(False)
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n", [400, 800, 141])
def test_kernel_variants_of_the_15x15_layer_and_the_tile_order_agree(n, monkeypatch):
    """Kept behind environment switches of the VARIANTS build (cost_kernels_variants.h; profiles/r05_cnn_variants.txt, r06_cnn_variants.txt): conv1 o conv2 as its own launch or fused (ARTP_CONV12_FUSED=0 / 1), its
    VALU form (ARTP_CONV12_MFMA=0), the 15 x 15 layer as K slice x row half on 18-row tiles (ARTP_KSPLIT_MS=2, 800^2 only), and two things (read at every feature-map update): the persistent strip-walking
    form of the 15 x 15 layer (conv_kwalk_kernel, ARTP_KWALK=1, three variants and two tile heights: built, measured slower) and
    the launch-order tile numbering (ARTP_CNN_XCD=0).  Every one of them must produce the default's features: the same
    products in fp32 accumulators, only the summation order of the K slices differs (one fp16 ulp of the stored feature), and equal
    the numpy oracle like the default does."""
    from synthetic import make_map
    gm = make_map(n, 0.04, seed=1234 if n == 400 else 77)
    elv = np.ascontiguousarray(gm["elevation"][::-1, ::-1]).astype(np.float16).astype(np.float32)
    p = mo.random_params(0)
    ctx = _vctx("yaml")
    ctx.cost_load_weights(convert_weights.to_blob(p))
    base = _gpu_features(ctx, elv, gm.res)
    # the variants library's default is the product's default: bit for bit the product library's features
    from art_planner_amd.context import Context
    prod = Context(0, "yaml")
    prod.cost_load_weights(convert_weights.to_blob(p))
    assert np.array_equal(_gpu_features(prod, elv, gm.res), base)
    prod.close()
    _assert_features_close(base, mo.cnn_features(p, elv), f"default {n}")
    settings = [{"ARTP_CONV12_FUSED": "0"}, {"ARTP_CONV12_FUSED": "1"}, {"ARTP_CONV12_MFMA": "0"}, {"ARTP_CNN_XCD": "0"}, {"ARTP_KSPLIT_MS": "2"}, {"ARTP_KWALK": "1"}, {"ARTP_KWALK": "1", "ARTP_KWALK_VARIANT": "1"},
                {"ARTP_KWALK": "1", "ARTP_KWALK_VARIANT": "2"}, {"ARTP_KWALK": "1", "ARTP_KWALK_TR": "8"},
                {"ARTP_KWALK": "1", "ARTP_KWALK_TR": "10", "ARTP_CNN_XCD": "0"},
                {"ARTP_CONV15_PAIR32": "1"}]   # round 6: the row-pair form on v_mfma_f32_32x32x16_f16 (launches with more tiles than CUs)
    for env in settings:
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        f = _gpu_features(ctx, elv, gm.res)
        for k in env:
            monkeypatch.delenv(k)
        d = np.abs(f - base)
        if "ARTP_CONV12_FUSED" in env:
            # conv1 o conv2 as a launch of its own or inside conv345's patch phase: the same MFMA tiles, the same bits
            assert np.array_equal(f, base), (env, float(d.max()))
            continue
        if "ARTP_CONV12_MFMA" in env:
            # the VALU form of conv1 o conv2 adds its 25 products in another order: single half-float ulps of the FIRST
            # activation, carried through four more layers -- held to the oracle like the default, and close to it
            _assert_features_close(f, mo.cnn_features(p, elv), f"{env} {n}")
            assert d.max() < 2e-2 and d.mean() < 2e-4, (env, float(d.max()), float(d.mean()))
            continue
        # one fp16 unit in the last place of the stored feature at most (another order of the same fp32 partial sums)
        assert (d <= 1.0e-3 + np.abs(base) * 2.0 ** -9).all() and d.mean() < 1e-4, (env, float(d.max()), float(d.mean()))
        # (not bit-equal even for the tile order alone: conv_ksplit_kernel rotates the order of its K slices with the
        # workgroup index, so another workgroup sums a tile's partial products in another order)
    ctx.close()


@pytest.mark.gpu
def test_lazy_path_check_breaks_cost_ties_like_the_search_from_scratch():
    """With the directional objective equal path costs are REAL: a chain of edges priced by their yaw differences costs
    exactly |yaw_end - yaw_start| / max_ang_vel whichever way it goes, so the lazy path check's removal sequence depends on
    which of several equally cheap paths the search returns.  The shortest-path tree that replaces the per-round search
     must return the one a search from scratch returns (roadmap.h LazyTree::before): same removed SET and
    same number of removals as the round-per-search loop (ARTP_SOLVE_ASTAR) and as the oracle's restatement of the
    reference loop -- on a query WITHOUT a valid path, where the loop runs until start and goal fall apart (hundreds of
    rounds; the first build of the tree got 273 removals here, the reference order has 275)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(common.ROOT, "oracle"))
    import oracle_py as O
    import prm_incremental as PI
    from art_planner_amd.context import Context
    from art_planner_amd.roadmap import Roadmap
    from synthetic import make_map
    gm = make_map(200, 0.04, seed=77)
    ctx = Context(0, "yaml", lib=_capi.VARIANTS_LIB_PATH)   # $ARTP_SOLVE_ASTAR is read by the variants build only
    ctx.upload_map(gm)
    se3 = ctx.sample_states(42, 0, 1 << 15)
    acc = se3[ctx.validate_states(se3) != 0]
    near = lambda xy: acc[np.argmin(np.hypot(acc[:, 0] - xy[0], acc[:, 1] - xy[1]))]
    q = 0.3 * gm.len_x
    s, g = near((gm.pos_x - q, gm.pos_y - q)), near((gm.pos_x + q, gm.pos_y + q))
    n_ms = min(1200, len(acc))
    got = {}
    for tag, env in (("tree", None), ("search per round", "1")):
        if env is None:
            os.environ.pop("ARTP_SOLVE_ASTAR", None)
        else:
            os.environ["ARTP_SOLVE_ASTAR"] = env
        try:
            rm = Roadmap(ctx, s, g, n_milestones=n_ms, seed=42, construction=2, objective=1, max_replans=100000)
            p, c, removed = rm.solve()
            ex = rm.export()
            got[tag] = (p is None, removed, np.asarray(ex["edge_removed"], np.uint8).copy(), ex["edges"].copy())
            rm.close()
        finally:
            os.environ.pop("ARTP_SOLVE_ASTAR", None)
    assert got["tree"][0] and got["search per round"][0], "this query has no valid path on this map"
    assert got["tree"][1] == got["search per round"][1] and got["tree"][1] > 100
    assert np.array_equal(got["tree"][2], got["search per round"][2])
    om, rob = O.OracleMap(gm), O.robot("yaml")
    ref = PI.lazy_prm_star_min_update(om, rob, acc, s, g, n_ms, cost_fn=_directional_cost, max_replans=100000)
    assert ref["path"] is None and ref["lazy_removals"] == got["tree"][1]
    left = {(int(u), int(v)) for (u, v), r in zip(got["tree"][3], got["tree"][2]) if not r}
    assert left == set(ref["graph"].edges.keys())
    ctx.close()
