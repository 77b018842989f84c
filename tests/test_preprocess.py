"""'Next' row N2: the per-map preprocessing chain on the device vs the numpy restatement of
processors::Basic / estimateNormals / the CDF (art_planner_amd/synthetic.py, which also generates the
benchmark maps).  OpenCV is not available: parity with the reference's cv::erode / cv::circle is unpinned."""
import numpy as np
import pytest

import common
import oracle_py as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,res,seed", [(200, 0.04, 5), (120, 0.05, 9)])
def test_device_preprocessing_matches_numpy_restatement(n, res, seed):
    from art_planner_amd.context import Context
    from synthetic import make_map
    gm = make_map(n, res, seed=seed)
    ctx = Context(0, "yaml")
    pp = ctx.preprocess_map(gm["elevation"], gm.len_x, gm.len_y, gm.pos_x, gm.pos_y,
                            traversability=gm["traversability"])
    # masks and everything built from min / max / select is exact
    for name in ("traversability_thresholded", "elevation_masked", "sample_probability"):
        assert np.array_equal(pp.layer(name), gm[name]), name
    assert np.array_equal(pp.layer("plane_fit_std_dev"), gm["plane_fit_std_dev"])
    # float sums: same order of accumulation, numpy's norm / division differ in the last bits
    for name in ("normal_x", "normal_y", "normal_z"):
        assert np.abs(pp.layer(name) - gm[name]).max() < 2e-6, name
    with np.errstate(invalid="ignore"):
        d = np.abs(pp.layer("cum_prob") - gm["cum_prob"])
        assert np.nanmax(d) < 1e-5 and np.array_equal(np.isnan(pp.layer("cum_prob")), np.isnan(gm["cum_prob"]))
    assert np.abs(pp.layer("cum_prob_rowwise") - gm["cum_prob_rowwise"]).max() < 1e-5
    pp.close()
    ctx.close()


def test_install_equals_host_upload():
    """artp_preprocessed_install == Planner::setMap with the host-preprocessed layers: same labels for the
    same states, the same sampler cells (the CDFs agree to 1e-5: a draw that close to a bin edge may land in the
    neighbouring cell)."""
    from art_planner_amd.context import Context
    from synthetic import make_map
    gm = make_map(200, 0.04, seed=5)
    a, b = Context(0, "yaml"), Context(0, "yaml")
    a.upload_map(gm)
    pp = b.preprocess_map(gm["elevation"], gm.len_x, gm.len_y, gm.pos_x, gm.pos_y, traversability=gm["traversability"])
    pp.install()
    se3 = a.sample_states(3, 0, 1 << 16)
    assert np.array_equal(a.validate_states(se3), b.validate_states(se3))
    assert np.array_equal(b.validate_states(se3[:20000]), O.OracleMap(gm).states_valid(O.robot("yaml"), se3[:20000]))
    # same cell; the position is the cell centre pushed along the (1e-6-different) normal
    sb = b.sample_states(3, 0, 1 << 16)
    same = np.abs(sb[:, :3] - se3[:, :3]).max(axis=1) < 1e-5
    assert same.mean() > 0.999
    # edges need the z bounds install() sets
    va, na = a.check_edges_interp(se3[:2000], se3[1:2001])
    vb, nb = b.check_edges_interp(se3[:2000], se3[1:2001])
    assert np.array_equal(va, vb) and np.array_equal(na, nb)
    assert np.array_equal(a.check_motions(se3[:500], se3[1:501]), b.check_motions(se3[:500], se3[1:501]))
    pp.close()
    a.close()
    b.close()


@pytest.mark.gpu
def test_reinstall_takes_the_rectangle_path_and_equals_a_fresh_install():
    """artp_preprocessed_install on a context that already holds a map of the same geometry compares the new height
    fields with the installed ones bit for bit and rewrites only the rectangle that differs (tables through the
    rectangle-update path).  After a small change, a no-op re-install and a large change the context must be
    indistinguishable from a fresh one that installed the final map: partner tables of both slots, labels, edges,
    the sampler's states; the map version moves with every install that changed something."""
    from art_planner_amd.context import Context
    from synthetic import make_map
    gm = make_map(200, 0.04, seed=5)
    elev = gm["elevation"].copy()
    a = Context(0, "yaml")

    def install(ctx, e):
        pp = ctx.preprocess_map(e, gm.len_x, gm.len_y, gm.pos_x, gm.pos_y, traversability=gm["traversability"])
        pp.install()
        return pp

    def same_as_fresh(e, tag):
        b = Context(0, "yaml")
        ppb = install(b, e)
        for slot in (0, 1):
            ta, ra = a.partner_table(slot, (gm.rows, gm.cols))
            tb, rb = b.partner_table(slot, (gm.rows, gm.cols))
            assert ra == rb > 0 and np.array_equal(ta, tb), (tag, slot, int((ta != tb).sum()))
        se3 = b.sample_states(3, 0, 1 << 16)
        assert np.array_equal(a.sample_states(3, 0, 1 << 16), se3, equal_nan=True), tag   # a NaN cell gives a NaN z
        assert np.array_equal(a.validate_states(se3), b.validate_states(se3)), tag
        fin = se3[np.isfinite(se3).all(axis=1)]                   # the edge entry points refuse non-finite states
        va, na = a.check_edges_interp(fin[:3000], fin[1:3001])
        vb, nb = b.check_edges_interp(fin[:3000], fin[1:3001])
        assert np.array_equal(va, vb) and np.array_equal(na, nb), tag
        assert np.array_equal(a.check_motions(fin[:500], fin[1:501]), b.check_motions(fin[:500], fin[1:501])), tag
        ppb.close()
        b.close()

    pps = [install(a, elev)]
    v0 = a.map_version()
    elev[60:85, 90:130] += np.float32(0.07)                       # a small patch: the rectangle path
    elev[70, 100] = np.nan                                        # ... that brings the layer's first NaN
    pps.append(install(a, elev))
    v1 = a.map_version()
    assert v1 > v0
    same_as_fresh(elev, "small change")
    pps.append(install(a, elev))                                  # nothing changed
    same_as_fresh(elev, "no change")
    elev[70, 100] = elev[70, 101]                                 # the NaN goes again: the flags must follow
    elev[10:190, 5:195] -= np.float32(0.02)                       # most of the map: the whole-layer path
    pps.append(install(a, elev))
    assert a.map_version() > v1
    same_as_fresh(elev, "large change")
    for pp in pps:
        pp.close()
    a.close()


def _gauss_taps(k, sigma):
    """cv::getGaussianKernel: float taps exp(-x^2 / (2 sigma^2)), normalised by their (double) sum."""
    x = np.arange(k) - (k - 1) * 0.5
    t = np.exp(-0.5 / (sigma * sigma) * x * x).astype(np.float32)
    return (t * (1.0 / t.astype(np.float64).sum())).astype(np.float32)


def _blur_reflect101(a, taps):
    r = len(taps) // 2
    out = a.astype(np.float32)
    for axis in (0, 1):
        p = np.pad(out, [(r, r) if ax == axis else (0, 0) for ax in (0, 1)], mode="reflect")  # gfedcb|abc...
        acc = np.zeros_like(out)
        for d in range(len(taps)):
            sl = [slice(None), slice(None)]
            sl[axis] = slice(d, d + out.shape[axis])
            acc = acc + taps[d] * p[tuple(sl)]
        out = acc.astype(np.float32)
    return out


def test_sampling_distribution_processors():
    """The rest of Planner::setUpMapProcessors' new-map chain (planner.cpp:43-56): inverse vertex density
    (sample_density.cpp:12-43), base distribution, capped unknown share (probability_distribution.cpp:50-90),
    CDF -- against a numpy restatement; and unknown_space_untraversable through the "observed" layer."""
    from art_planner_amd.context import Context
    from synthetic import make_map, cumulative_distribution
    gm = make_map(160, 0.05, seed=12)
    ctx = Context(0, "yaml")
    rng = np.random.default_rng(4)
    observed = np.ones((gm.rows, gm.cols), np.float32)
    observed[:, :40] = 0.0                      # a strip the sensors never saw
    verts = np.zeros((3000, 7))
    verts[:, 0] = gm.pos_x + rng.uniform(-0.6, 0.6, 3000) * gm.len_x   # some fall outside the map
    verts[:, 1] = gm.pos_y + rng.normal(0, 0.12, 3000) * gm.len_y
    verts[:, 6] = 1.0
    plain = ctx.preprocess_map(gm["elevation"], gm.len_x, gm.len_y, gm.pos_x, gm.pos_y,
                               traversability=gm["traversability"], use_inverse_vertex_density=0,
                               use_max_prob_unknown_samples=0)
    sf = plain.layer("traversability_sample_filter")
    assert np.array_equal(sf, plain.layer("sample_probability"))        # 1 * filter
    # with everything on; unknown space stays traversable here so that it keeps some probability mass
    prm = ctx.params
    ctx2 = Context(0, __import__("art_planner_amd.context", fromlist=["make_params"]).make_params(
        "yaml", unknown_space_untraversable=0))
    pp = ctx2.preprocess_map(gm["elevation"], gm.len_x, gm.len_y, gm.pos_x, gm.pos_y,
                             traversability=gm["traversability"], observed=observed, vertices=verts,
                             max_prob_unknown_samples=0.05)
    # --- restatement
    res = gm.res
    tx = -((verts[:, 0] - gm.pos_x) - 0.5 * gm.len_x)
    ty = -((verts[:, 1] - gm.pos_y) - 0.5 * gm.len_y)
    ins = (tx >= 0) & (ty >= 0) & (tx < gm.len_x) & (ty < gm.len_y)
    cnt = np.zeros((gm.rows, gm.cols), np.float32)
    np.add.at(cnt, ((tx[ins] / np.float32(res).astype(np.float64)).astype(int).clip(0, gm.rows - 1),
                    (ty[ins] / np.float32(res).astype(np.float64)).astype(int).clip(0, gm.cols - 1)), 1.0)
    radius = (prm.torso_length + prm.torso_width) * 0.25
    k = int(6 * radius / res)
    k += 1 if k % 2 == 0 else 0
    blurred = _blur_reflect101(cnt, _gauss_taps(k, radius / res))
    got_blur = pp.layer("n_samples")
    assert np.abs(got_blur - blurred).max() < 1e-5 * max(1.0, blurred.max())
    prob = (np.float32(got_blur.max()) - got_blur) * sf
    known, unknown = prob[observed > 0].astype(np.float64).sum(), prob[observed <= 0].astype(np.float64).sum()
    assert unknown / (known + unknown) > 0.05                            # the cap really acts
    mult = np.where(observed > 0, (1 - 0.05) / known, 0.05 / unknown).astype(np.float32)
    expect = prob * mult
    got = pp.layer("sample_probability")
    assert np.abs(got - expect).max() <= 2e-6 * expect.max()
    assert abs(got[observed <= 0].astype(np.float64).sum() / got.astype(np.float64).sum() - 0.05) < 1e-5
    cp, cr = cumulative_distribution(got)
    with np.errstate(invalid="ignore"):
        assert np.nanmax(np.abs(pp.layer("cum_prob") - cp)) < 1e-5
    assert np.abs(pp.layer("cum_prob_rowwise") - cr).max() < 1e-5
    # unknown_space_untraversable (the default): unobserved cells lose their traversability
    pu = ctx.preprocess_map(gm["elevation"], gm.len_x, gm.len_y, gm.pos_x, gm.pos_y,
                            traversability=gm["traversability"], observed=observed)
    assert (pu.layer("traversability_thresholded_no_safety")[:, :40] == 0).all()
    assert np.isneginf(pu.layer("elevation_masked")[:, :35]).all()
    for m in (plain, pp, pu):
        m.close()
    ctx.close()
    ctx2.close()


def test_change_detection_between_maps():
    """computeChange (change.cpp:9-51): the 'updated' layer between an old and a new map, also when the new
    map's origin moved by whole cells; the rectangle bounds the updated cells."""
    from art_planner_amd.context import Context
    from synthetic import make_map
    gm = make_map(160, 0.05, seed=12)
    ctx = Context(0, "yaml")
    old = ctx.preprocess_map(gm["elevation"], gm.len_x, gm.len_y, gm.pos_x, gm.pos_y, traversability=gm["traversability"])
    e2 = gm["elevation"].copy()
    e2[30:50, 70:95] += np.float32(0.2)           # a changed block
    t2 = gm["traversability"].copy()
    t2[100:110, 20:30] = 0.0                       # a patch that turned untraversable
    new = ctx.preprocess_map(e2, gm.len_x, gm.len_y, gm.pos_x, gm.pos_y, traversability=t2)
    upd, rect, cnt = new.change_from(old, 0.05)
    safety_old, safety_new = old.layer("traversability_thresholded"), new.layer("traversability_thresholded")
    expect = ((np.abs(e2 - gm["elevation"]) > np.float32(0.05)) | ((safety_old - safety_new) > 0.5)).astype(np.float32)
    assert np.array_equal(upd, expect) and cnt == int(expect.sum())
    ii, jj = np.nonzero(expect)
    assert rect == (ii.min(), jj.min(), ii.max() - ii.min() + 1, jj.max() - jj.min() + 1)
    assert expect[30:50, 70:95].all() and expect[100:110, 20:30].any()
    same_upd, same_rect, same_cnt = old.change_from(old, 0.05)
    assert same_cnt == 0 and same_rect[2] == 0 and not same_upd.any()
    # the map followed the robot: origin shifted by (+3, -2) cells; cells without an old counterpart are "updated"
    sh = ctx.preprocess_map(e2, gm.len_x, gm.len_y, gm.pos_x + 3 * gm.res, gm.pos_y - 2 * gm.res, traversability=t2)
    upd2, _, _ = sh.change_from(old, 0.05)
    exp2 = np.ones_like(expect)
    ss_old = safety_old
    ss_new = sh.layer("traversability_thresholded")
    for i in range(gm.rows):
        io = i - 3
        if not 0 <= io < gm.rows:
            continue
        for j in range(gm.cols):
            jo = j + 2
            if 0 <= jo < gm.cols:
                ch = abs(e2[i, j] - gm["elevation"][io, jo]) > np.float32(0.05) or (ss_old[io, jo] - ss_new[i, j]) > 0.5
                exp2[i, j] = 1.0 if ch else 0.0
    assert np.array_equal(upd2, exp2)
    for m in (old, new, sh):
        m.close()
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_hole_filling_quantisation_matches_the_reference_arithmetic(mode):
    """artp_inpaint_layer: inpaintMatrix (utils.cpp:13-64, mode 0) and the cost node's _elvMapProcess
    (cost_query_server.py:92-111, mode 1).  The reference quantises the WHOLE layer to 8 bit when it has holes;
    that arithmetic is restated exactly, so every cell that is not a hole must be bit-equal to the numpy restatement
    (mode 0 also carries the column-0 / row-0 copy).  The fill of the hole cells is not Telea's (unpinned): they
    must be finite and lie within the range of the valid cells around them."""
    from art_planner_amd.context import Context
    from synthetic import raw_map
    gm = raw_map(160, 0.05, seed=12)
    elev = gm["elevation"].copy()
    rng = np.random.default_rng(3)
    holes = np.zeros(elev.shape, bool)
    for _ in range(14):  # blobs of missing cells, one of them wider than the fill radius
        r, c, w = rng.integers(5, 150), rng.integers(5, 150), rng.integers(1, 9)
        holes[r:r + w, c:c + w] = True
    holes[100:120, 40:52] = True
    elev[holes] = np.nan
    ctx = Context(0, "yaml")
    out, n = ctx.inpaint_layer(elev, mode)
    assert n == holes.sum() and np.isfinite(out).all()
    telea = bool(mode & 2)     # ARTP_INPAINT_TELEA: the fill by fast marching (host); quantisation etc. as without it
    mode &= 1
    lo, hi = np.float32(np.nanmin(elev)), np.float32(np.nanmax(elev))
    with np.errstate(invalid="ignore"):
        if mode == 0:
            a, b = np.float32(255) / (hi - lo), -lo * np.float32(255) / (hi - lo)
            q = np.clip(np.rint(elev * a + b), 0, 255)                       # cvRound, saturate_cast<uchar>
            ref = q.astype(np.float32) * ((hi - lo) / np.float32(255)) + lo
        else:
            q = np.clip(np.trunc((elev - lo) * np.float32(255) / (hi - lo)), 0, 255)   # .astype(np.uint8)
            ref = q.astype(np.float32) * (hi - lo) / np.float32(255) + lo
    keep = ~holes
    if mode == 0:                       # mat_inpainted.col(0) = col(1); row(0) = row(1)
        ref[:, 0] = ref[:, 1]
        keep[:, 0] = keep[:, 1]
        ref[0, :] = ref[1, :]
        keep[0, :] = keep[1, :]
    assert np.array_equal(out[keep], ref[keep])
    # hole cells: between the extremes of the valid cells within the reach of the fill (the blob + 3 cells)
    from scipy import ndimage
    near = ndimage.binary_dilation(holes, iterations=12) & ~holes
    slack = (hi - lo) * 3 / 255 if telea else 1e-6   # Telea's first-order term may overshoot by a few levels
    assert out[holes].min() >= ref[near].min() - slack and out[holes].max() <= ref[near].max() + slack
    if telea:
        # the device-side mean fill and Telea's fill agree on everything but the hole cells
        plain, _ = ctx.inpaint_layer(elev, mode)
        same = ~holes
        if mode == 0:
            same[:, 0] = same[:, 1]
            same[0, :] = same[1, :]
        assert np.array_equal(out[same], plain[same]) and not np.array_equal(out[holes], plain[holes])
        assert np.abs(out[holes] - plain[holes]).max() <= (hi - lo) * 0.5   # an obstacle edge inside a wide hole: the two fills differ by up to its height
    # a layer without holes comes back untouched (the reference only inpaints when something is missing)
    whole, n0 = ctx.inpaint_layer(gm["elevation"], mode | (2 if telea else 0))
    assert n0 == 0 and np.array_equal(whole, gm["elevation"])
    ctx.close()


@pytest.mark.gpu
def test_cost_map_layer_with_holes_is_filled_when_asked(big_map):
    """N3: with artp_cost_set_hole_filling the cost map accepts a layer with holes like the cost node does --
    same features as handing over the layer filled by artp_inpaint_layer(mode 1) -- and still refuses it without."""
    import os
    import sys
    sys.path.insert(0, os.path.join(common.ROOT, "oracle"))
    sys.path.insert(0, os.path.join(common.ROOT, "tools"))
    import convert_weights
    from art_planner_amd._capi import ArtpError
    from art_planner_amd.context import Context
    ctx = Context(0, "yaml")
    ctx.cost_load_weights(convert_weights.to_blob(convert_weights.random_params(0)))
    layer = big_map["elevation"].copy()
    layer[50:58, 200:211] = np.nan
    layer[300, 17] = np.inf
    with pytest.raises(ArtpError):
        ctx.cost_update_map_layer(layer, big_map.res, big_map.len_x, big_map.len_y)
    ctx.cost_set_hole_filling(True)
    ctx.cost_update_map_layer(layer, big_map.res, big_map.len_x, big_map.len_y)
    f1 = ctx.cost_features()
    filled, n = ctx.inpaint_layer(layer, 1)
    assert n == 8 * 11 + 1
    ctx.cost_set_hole_filling(False)
    ctx.cost_update_map_layer(filled, big_map.res, big_map.len_x, big_map.len_y)
    assert np.array_equal(f1, ctx.cost_features()) and np.isfinite(f1).all()
    ctx.close()
