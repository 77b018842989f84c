"""Learned motion cost: numpy oracle vs reference golden vectors (CPU suite) and
the HIP MFMA conv stack + per-edge MLP vs the same vectors (GPU suite)."""
import os
import sys

import numpy as np
import pytest

import common

sys.path.insert(0, os.path.join(common.ROOT, "oracle"))
sys.path.insert(0, os.path.join(common.ROOT, "tools"))
import motion_cost_oracle as mo  # noqa: E402
import convert_weights  # noqa: E402

GOLD = np.load(os.path.join(common.GOLDEN_DIR, "motion_cost.npz"))
# The reference network's OWN torch.half-vs-float32 error on the inputs of these tests (the reference runs in half,
# predictor.py:22,34,44), measured with the imported reference class by tests/golden/make_golden_cost_anchor.py.
# The HIP path (fp16 activations, fp32 accumulation) must stay within ANCHOR_FACTOR x these figures of the float32
# result: the survey's 2e-3 rel / 1e-3 abs (8c) is tighter than the reference's own half evaluation achieves
# (C3: cost errors up to 8e-3, 2.5e-3 beyond the relative part for 1 % of the edges), so it cannot be the bar.
# Factor 1.0 since round 4 (1.5 before): the HIP path may not lose more than the reference's own half evaluation does;
# measured it loses 0.3 - 0.55 x that (gpurun_out/motion_cost_err_*.json).
import json  # noqa: E402
ANCHOR = json.load(open(os.path.join(common.GOLDEN_DIR, "motion_cost_fp16_anchor.json")))["cases"]
ANCHOR_FACTOR = 1.0
CQ_REF = np.load(os.path.join(common.GOLDEN_DIR, "cost_query_ref.npz"))


def _assert_within_reference_half_error(f_hwc, ref_chw, c, c_ref, case, report=None):
    """features [F,F,48] / costs [B,3] of the HIP path vs the float32 reference values, against the anchor `case`."""
    a = ANCHOR[case]
    ref = np.transpose(ref_chw, (1, 2, 0))
    assert f_hwc.shape == ref.shape, (case, f_hwc.shape, ref.shape)
    fe = np.abs(f_hwc - ref)
    ce = np.abs(c - c_ref)
    cex = (ce - 2e-3 * np.abs(c_ref)).max(axis=1)
    got = {"feat_err_max": float(fe.max()), "feat_err_mean": float(fe.mean()), "feat_err_q999": float(np.quantile(fe, 0.999)),
           "cost_err_max": float(ce.max()), "cost_err_mean": float(ce.mean()),
           "cost_excess_over_2e-3_rel_q99": float(np.quantile(cex, 0.99)), "cost_excess_over_2e-3_rel_max": float(cex.max())}
    if report is not None:
        report[case] = {k: [got[k], a[k]] for k in got}
    for k, v in got.items():
        assert v <= ANCHOR_FACTOR * a[k], (case, k, v, "reference half-vs-float32:", a[k])


def test_oracle_matches_reference_network_golden():
    p = mo.random_params(0)
    crop = GOLD["crop"].astype(np.float32)
    res = float(GOLD["res"])
    L = crop.shape[0] * res
    f = mo.cnn_features(p, crop)
    assert f.shape == (48, 32, 32)
    assert np.abs(f - GOLD["features"]).max() < 1e-3
    c = mo.fc_costs(p, GOLD["features"], GOLD["edges"], res, L, L)
    assert np.abs(c - GOLD["costs"]).max() < 1e-4


@pytest.mark.parametrize("tag", ["sq", "rect"])
def test_oracle_gather_and_costs_equal_the_reference_cost_query(tag):
    """R9 pinned on the reference's OWN CostQuery (cost_query.py:26-35,39-69, imported where it lies by
    tests/golden/make_golden_cost.py): rows / cols of the gather -- starts far outside the map (both clamps), starts
    exactly on feature-cell borders +- one float32 ulp (the .long() truncation decides), a non-square map -- are
    IDENTICAL; the costs equal the reference FCpart's on the gathered features."""
    g = CQ_REF
    crop = g[f"{tag}_crop"].astype(np.float32)
    res = float(g["res"])
    Lx, Ly = crop.shape[0] * res, crop.shape[1] * res
    p = mo.random_params(0)
    f = mo.cnn_features(p, crop)
    assert (tag == "rect") == (f.shape[1] != f.shape[2])
    rows, cols = mo.query_cells(g[f"{tag}_edges"], res, Lx, Ly, (f.shape[1], f.shape[2]))
    assert np.array_equal(rows, g[f"{tag}_rows"]) and np.array_equal(cols, g[f"{tag}_cols"])
    assert rows.min() == 1 and rows.max() == f.shape[1] - 2 and cols.min() == 1 and cols.max() == f.shape[2] - 2
    c = mo.fc_costs(p, f, g[f"{tag}_edges"], res, Lx, Ly)
    assert np.abs(c - g[f"{tag}_costs"]).max() < 1e-4


def test_blob_layout():
    blob = convert_weights.to_blob(mo.random_params(0))
    n_conv = sum(int(np.prod(mo.SHAPES[k])) + mo.SHAPES[k][0] for k in convert_weights.CONVS)
    n_fc = sum(int(np.prod(mo.SHAPES[k])) + mo.SHAPES[k][0] for k in convert_weights.FC_BN)
    n_out = sum(int(np.prod(mo.SHAPES[k])) + 1 for k in convert_weights.FC_OUT)
    assert len(blob) == 8 + 4 * (n_conv + n_fc + n_out)
    assert blob[:4] == b"ARMC" and blob[4] == 1
    from art_planner_amd import _capi
    assert _capi.load().artp_cost_blob_bytes() == len(blob)


def test_oracle_matches_reference_network_partial_tile_size():
    """The second reference fixture: 120 x 120 -> 36 x 36 features (partial tiles in the HIP kernels)."""
    g = np.load(os.path.join(common.GOLDEN_DIR, "motion_cost_120.npz"))
    f = mo.cnn_features(mo.random_params(0), g["crop"].astype(np.float32))
    assert f.shape == (48, 36, 36)
    assert np.abs(f - g["features"]).max() < 1e-3


def _check_blob_from_checkpoint(path, tmp_path, monkeypatch):
    out = tmp_path / "model.armc"
    monkeypatch.setattr(sys, "argv", ["convert_weights.py", str(path), str(out)])
    convert_weights.main()
    assert open(out, "rb").read() == convert_weights.to_blob(mo.random_params(0))


def test_convert_weights_main_on_a_torch_saved_state_dict(tmp_path, monkeypatch):
    """N3: tools/convert_weights.py main() on a torch.save()d state_dict with the reference's key set (incl. the
    BatchNorm num_batches_tracked entries a real checkpoint carries) gives the blob of the same parameters."""
    import torch
    p = mo.random_params(0)
    sd = {}
    for name in convert_weights.SHAPES:           # the order network_light.py registers its modules in
        sd[name + ".weight"] = torch.from_numpy(p[name + ".weight"].copy())
        if name in convert_weights.WITH_BIAS:
            sd[name + ".bias"] = torch.from_numpy(p[name + ".bias"].copy())
        else:
            for k in ("weight", "bias", "running_mean", "running_var"):
                sd[f"{name}_bn.{k}"] = torch.from_numpy(p[f"{name}_bn.{k}"].copy())
            sd[f"{name}_bn.num_batches_tracked"] = torch.tensor(0)
    ck = tmp_path / "model.pt"
    torch.save(sd, ck)
    _check_blob_from_checkpoint(ck, tmp_path, monkeypatch)


@pytest.mark.skipif(not os.path.isdir("/root/reference/art_planner_motion_cost"),
                    reason="the reference network class only exists in the build container")
def test_convert_weights_main_on_the_reference_networks_checkpoint(tmp_path, monkeypatch):
    """The same through the REFERENCE class: network_light.network().state_dict() torch.save()d by
    tests/golden/make_reference_state_dict.py (the format predictor.py:20 loads)."""
    import subprocess
    ck = tmp_path / "ref_model.pt"
    subprocess.check_call([sys.executable, os.path.join(common.GOLDEN_DIR, "make_reference_state_dict.py"), str(ck)])
    _check_blob_from_checkpoint(ck, tmp_path, monkeypatch)


def _gpu_features(ctx, elv, res):
    ctx.cost_update_map(np.ascontiguousarray(elv, np.float32), res, elv.shape[0] * res, elv.shape[1] * res)
    return ctx.cost_features()


def _assert_features_close(f, ref_chw, what):
    """fp16 activations / fp32 accumulate vs the float32 reference: 2e-2 absolute per element + 2e-3 mean
    (values reach +-5; six fp16-rounded layers, K up to 10800)."""
    ref = np.transpose(ref_chw, (1, 2, 0))
    assert f.shape == ref.shape, (what, f.shape, ref.shape)
    err = np.abs(f - ref)
    assert err.max() < 2e-2 + 4e-3 * np.abs(ref).max(), (what, float(err.max()), float(err.mean()))
    assert err.mean() < 3e-3, (what, float(err.mean()))


@pytest.mark.gpu
def test_gpu_features_partial_tiles_match_reference_golden():
    """120 x 120 -> 36 x 36: partial tiles in every MFMA kernel (36 = 2 * 16 + 4 pixels, 4 * 8 + 4 rows) against
    the REFERENCE network's features."""
    from art_planner_amd.context import Context
    g = np.load(os.path.join(common.GOLDEN_DIR, "motion_cost_120.npz"))
    ctx = Context(0, "yaml")
    ctx.cost_load_weights(convert_weights.to_blob(mo.random_params(0)))
    f = _gpu_features(ctx, g["crop"].astype(np.float32), float(g["res"]))
    _assert_features_close(f, g["features"], "120x120")
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n,F", [(400, 176), (800, 376), (141, 46), (97, 24)])
def test_gpu_features_match_oracle_at_c3_c4_and_odd_sizes(n, F):
    """C3 (400^2 -> 176^2), C4 (800^2 -> 376^2: 23.5 16-pixel tiles per row) and two odd sizes against the numpy
    oracle (pinned on the reference network at 112^2 and 120^2), then the per-edge costs of 20 000 edges against
    the oracle fed with the ORACLE's features (whole-path parity, not the GPU's own features)."""
    from art_planner_amd.context import Context
    from synthetic import make_map
    gm = make_map(n, 0.04, seed=1234 if n == 400 else 77)
    elv = np.ascontiguousarray(gm["elevation"][::-1, ::-1]).astype(np.float16).astype(np.float32)
    p = mo.random_params(0)
    ref = mo.cnn_features(p, elv)
    assert ref.shape == (48, F, F)
    ctx = Context(0, "yaml")
    ctx.cost_load_weights(convert_weights.to_blob(p))
    f = _gpu_features(ctx, elv, gm.res)
    rng = np.random.default_rng(n)
    B = 20000
    s = rng.uniform(-0.55 * gm.len_x, 0.55 * gm.len_x, (B, 2))
    d = rng.uniform(-0.6, 0.6, (B, 2))
    e = np.stack([s[:, 0] + d[:, 0], s[:, 1] + d[:, 1], rng.uniform(-np.pi, np.pi, B), s[:, 0], s[:, 1],
                  rng.uniform(-np.pi, np.pi, B)], 1).astype(np.float32)
    c = ctx.cost_query(e)
    co = mo.fc_costs(p, ref, e, gm.res, gm.len_x, gm.len_y)
    # against the float32 oracle (== the reference network to 2e-5), within 1.5x of what the reference's own half
    # evaluation loses on the same map and the same edges
    report = {}
    try:
        _assert_within_reference_half_error(f, ref, c, co, f"map_{n}", report)
    finally:
        out = os.path.join(common.ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        json.dump(report, open(os.path.join(out, f"motion_cost_err_{n}.json"), "w"), indent=1)
    ctx.close()


@pytest.mark.gpu
def test_gpu_features_and_costs_match_reference_golden():
    """fp16 activations / fp32 accumulate on the matrix cores vs the reference network in float32 (the committed
    fixture): feature and edge-cost errors within 1.5x of the error of the reference's own torch.half evaluation."""
    from art_planner_amd.context import Context
    ctx = Context(0, "yaml")
    ctx.cost_load_weights(convert_weights.to_blob(mo.random_params(0)))
    crop = GOLD["crop"].astype(np.float32)
    res = float(GOLD["res"])
    L = crop.shape[0] * res
    ctx.cost_update_map(crop, res, L, L)
    f = ctx.cost_features()                      # [F][F][48]
    c = ctx.cost_query(GOLD["edges"])
    # vs the REFERENCE network's float32 outputs, within 1.5x of the reference's own half-vs-float32 error
    _assert_within_reference_half_error(f, GOLD["features"], c, GOLD["costs"], "golden_112")
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["sq", "rect"])
def test_gpu_gather_and_costs_equal_the_reference_cost_query(tag):
    """The HIP cost query against the reference's own CostQuery (tests/golden/cost_query_ref.npz): the gathered cell of
    every edge -- computed by the device function both cost kernels use -- is IDENTICAL to the reference's (clamps,
    cell borders +- 1 ulp, non-square feature map 36 x 32), the costs within the reference's own half-vs-float32 error."""
    from art_planner_amd.context import Context
    g = CQ_REF
    crop = g[f"{tag}_crop"].astype(np.float32)
    res = float(g["res"])
    Lx, Ly = crop.shape[0] * res, crop.shape[1] * res
    ctx = Context(0, "yaml")
    ctx.cost_load_weights(convert_weights.to_blob(mo.random_params(0)))
    ctx.cost_update_map(crop, res, Lx, Ly)
    f = ctx.cost_features()
    assert f.shape[:2] == ((36, 32) if tag == "rect" else (32, 32))
    e = g[f"{tag}_edges"]
    rows, cols = ctx.cost_query_cells(e)
    assert np.array_equal(rows, g[f"{tag}_rows"]) and np.array_equal(cols, g[f"{tag}_cols"])
    fc = ctx.cost_fc_path()                                 # the MFMA form passed its probe batch at weight load
    assert fc["mfma"] == 1 and fc["selfcheck"] == 1 and fc["max_abs_diff"] < 1e-4, fc
    c = ctx.cost_query(e)
    big = np.tile(e, (24, 1))                               # an edge's cost does not depend on the batch around it
    assert np.array_equal(ctx.cost_query(big)[:len(e)], c)
    ce = np.abs(c - g[f"{tag}_costs"])
    a = ANCHOR["golden_112"]
    assert ce.max() <= ANCHOR_FACTOR * a["cost_err_max"] and ce.mean() <= ANCHOR_FACTOR * a["cost_err_mean"], (ce.max(), ce.mean())
    # and the features the costs were computed on, against the float32 oracle (== the reference to 2e-5)
    _assert_features_close(f, mo.cnn_features(mo.random_params(0), crop), tag)
    ctx.close()


@pytest.mark.gpu
def test_gpu_cost_full_map_properties(big_map):
    """C3 size (400x400): feature map 176x176; queries are deterministic, depend only on the start cell
    and the delta pose, and clamp at the feature-map border like the reference."""
    from art_planner_amd.context import Context
    ctx = Context(0, "yaml")
    p = mo.random_params(0)
    ctx.cost_load_weights(convert_weights.to_blob(p))
    elv = np.ascontiguousarray(big_map["elevation"][::-1, ::-1]).astype(np.float32)
    ctx.cost_update_map(elv, big_map.res, big_map.len_x, big_map.len_y)
    f = ctx.cost_features()
    assert f.shape == (176, 176, 48) and np.isfinite(f).all()
    rng = np.random.default_rng(1)
    B = 50000
    s = rng.uniform(-9, 9, (B, 2))      # some starts outside the 16 m map -> clamped rows/cols
    d = rng.uniform(-0.5, 0.5, (B, 2))
    e = np.stack([s[:, 0] + d[:, 0], s[:, 1] + d[:, 1], rng.uniform(-3, 3, B), s[:, 0], s[:, 1],
                  rng.uniform(-3, 3, B)], 1).astype(np.float32)
    c1 = ctx.cost_query(e)
    assert np.array_equal(c1, ctx.cost_query(e))
    assert (c1[:, 0] >= 0).all() and (c1[:, 1] >= 0).all() and ((c1[:, 2] >= 0) & (c1[:, 2] <= 1)).all()
    # against the numpy oracle fed with the GPU's own feature map (isolates the per-edge MLP, fp32)
    co = mo.fc_costs(p, np.transpose(f, (2, 0, 1)), e[:4096], big_map.res, big_map.len_x, big_map.len_y)
    # the MLP runs as MFMA tiles with half-float hi / lo operand pairs (cost_kernels.h fc_cost_mfma_kernel): fp32 accuracy
    assert np.abs(c1[:4096] - co).max() < 5e-5
    # ... and equals the fp32 VALU kernels (a lane per edge / four lanes per edge; artp_cost_set_fc_path(ctx, 0) before the load)
    ctx_v = Context(0, "yaml")
    ctx_v.cost_set_fc_path(False)
    ctx_v.cost_load_weights(convert_weights.to_blob(p))
    ctx_v.cost_update_map(elv, big_map.res, big_map.len_x, big_map.len_y)
    assert ctx_v.cost_fc_path() == {"mfma": 0, "selfcheck": -1, "max_abs_diff": 0.0}
    c_v = ctx_v.cost_query(e)                               # 50 000 edges <= 2^16: fc_cost_split_kernel, four lanes per edge
    assert np.abs(c1 - c_v).max() < 2e-5 and not np.array_equal(c1, c_v)   # two different kernels, the same numbers
    # the two fp32 VALU kernels accumulate every unit in the same order: bit-identical (> 2^16 edges: fc_cost_kernel, a lane per edge)
    c_vb = ctx_v.cost_query(np.tile(e, (2, 1)))
    assert np.array_equal(c_vb[:B], c_v) and np.array_equal(c_vb[B:], c_v)
    ctx_v.close()
    # an edge's result does not depend on the batch it arrives in (tile position, ragged sizes, small and large batches)
    big = np.tile(e, (2, 1))[:70001]
    cb = ctx.cost_query(big)
    assert np.array_equal(cb[:B], c1) and np.array_equal(cb[B:], c1[:70001 - B])
    for n_small in (1, 63, 64, 65, 4097):
        assert np.array_equal(ctx.cost_query(e[:n_small]), c1[:n_small])
    ctx.close()


@pytest.mark.gpu
def test_cost_map_from_gridmap_layer_matches_server_layout(big_map):
    """N3: the cost node receives the planner's grid_map layer and re-indexes it with
    np.rot90(.., 2).transpose() (cost_query_server.py:66-74); artp_cost_update_map_layer must give the same
    features as handing over the re-indexed array, and refuse a layer with holes."""
    from art_planner_amd._capi import ArtpError
    from art_planner_amd.context import Context
    a, b = Context(0, "yaml"), Context(0, "yaml")
    blob = convert_weights.to_blob(mo.random_params(0))
    a.cost_load_weights(blob)
    b.cost_load_weights(blob)
    layer = big_map["elevation"]
    # what the server computes from the message: data = column-major layer, dims (cols, rows)
    sent = np.asarray(layer, np.float32).flatten(order="F")
    server = np.rot90(sent.reshape((layer.shape[1], layer.shape[0])), 2).transpose()
    a.cost_update_map(np.ascontiguousarray(server), big_map.res, big_map.len_x, big_map.len_y)
    b.cost_update_map_layer(layer, big_map.res, big_map.len_x, big_map.len_y)
    assert np.array_equal(a.cost_features(), b.cost_features())
    holes = layer.copy()
    holes[5, 7] = np.nan
    with pytest.raises(ArtpError):
        b.cost_update_map_layer(holes, big_map.res, big_map.len_x, big_map.len_y)
    a.close()
    b.close()
