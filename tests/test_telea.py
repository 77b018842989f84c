"""N3: Telea's fast-marching hole fill (art_planner_amd/csrc/telea.h; cv::inpaint(..., 3, cv::INPAINT_TELEA) in
art_planner/src/utils.cpp:44-48 and cost_query_server.py:107) through the context-free C entry artp_telea_inpaint_u8 -- host
code, runs without a GPU.  OpenCV is not installed here: the algorithm is restated (unpinned), the tests are analytic."""
import numpy as np

import common  # noqa: F401  (sys.path)


def _telea(img, mask, r=3):
    import ctypes as C  # noqa: F401
    from art_planner_amd import _capi
    L = _capi.load()
    img = np.ascontiguousarray(img, np.uint8)
    mask = np.ascontiguousarray(mask, np.uint8)
    out = np.empty_like(img)
    assert L.artp_telea_inpaint_u8(img.ctypes.data, mask.ctypes.data, img.shape[0], img.shape[1], r, out.ctypes.data) == 0
    return out


def test_telea_fill_analytic_cases():
    """N3: the fast-marching fill behind ARTP_INPAINT_TELEA (csrc/telea.h, cv::inpaint's conventions; unpinned -- no OpenCV
    here) on cases with known answers.  Host code: no GPU needed."""
    H, W = 40, 50
    m = np.zeros((H, W), np.uint8)
    m[10:20, 15:30] = 1
    # known pixels are never touched; a layer without holes comes back as it is
    rng = np.random.default_rng(0)
    noise = rng.integers(0, 255, (H, W)).astype(np.uint8)
    o = _telea(noise, m)
    assert np.array_equal(o[m == 0], noise[m == 0])
    assert np.array_equal(_telea(noise, np.zeros((H, W), np.uint8)), noise)
    assert np.array_equal(o, _telea(noise, m))                       # deterministic
    # the masked pixels' own values do not matter
    junk = noise.copy()
    junk[m == 1] = 255
    assert np.array_equal(_telea(junk, m), o)
    # a constant image stays constant to within the method's rounding (value + 0.5, round to nearest, marching inwards)
    const = np.full((H, W), 77, np.uint8)
    oc = _telea(const, m)
    assert np.abs(oc[m == 1].astype(int) - 77).max() <= 2
    # a planar ramp across a small hole: the weighted average + normalised first-order term stay near the plane
    small = np.zeros((H, W), np.uint8)
    small[15:22, 20:27] = 1
    ramp_x = np.tile((np.arange(W) * 4).astype(np.uint8), (H, 1))
    ramp_y = np.tile((np.arange(H) * 5).astype(np.uint8)[:, None], (1, W))
    for ramp, step in ((ramp_x, 4), (ramp_y, 5)):
        orr = _telea(ramp, small)
        assert np.abs(orr[small == 1].astype(int) - ramp[small == 1].astype(int)).max() <= 3 * step
        assert orr[small == 1].min() >= int(ramp[14:23, 19:28].min()) - step and orr[small == 1].max() <= int(ramp[14:23, 19:28].max()) + step
    # a single missing pixel: close to the distance-weighted mean of its radius-3 disc
    one = np.zeros((H, W), np.uint8)
    one[20, 25] = 1
    oo = _telea(noise, one)
    yy, xx = np.mgrid[-3:4, -3:4]
    d2 = (yy * yy + xx * xx).astype(float)
    sel = (d2 > 0) & (d2 <= 9)
    patch = noise[17:24, 22:29].astype(float)
    lo, hi = patch[sel].min(), patch[sel].max()
    assert lo <= oo[20, 25] <= hi
    # holes at the border and in a corner; a hole much wider than the radius; an image with ONE known pixel
    edge = np.zeros((H, W), np.uint8)
    edge[0:5, 0:6] = 1
    edge[H - 3:, W - 4:] = 1
    oe = _telea(np.full((H, W), 120, np.uint8), edge)
    assert np.abs(oe[edge == 1].astype(int) - 120).max() <= 12
    big = np.zeros((H, W), np.uint8)
    big[5:35, 10:40] = 1
    ob = _telea(ramp_x, big)
    assert ob[big == 1].min() >= int(ramp_x[:, 9].min()) - 8 and ob[big == 1].max() <= int(ramp_x[:, 40].max()) + 8
    assert (np.diff(ob[20, 10:40].astype(int)) >= -4).all()      # still rising from the low rim to the high one
    lone = np.ones((H, W), np.uint8)
    lone[7, 9] = 0
    img = np.zeros((H, W), np.uint8)
    img[7, 9] = 200
    ol = _telea(img, lone)
    assert ol[7, 9] == 200 and np.abs(ol.astype(int) - 200).max() <= 10


def test_telea_refuses_degenerate_images():
    """cv::inpaint's border index rule reads row / column 1 of the image, so a one-row or one-column image
    would be read out of bounds: INVALID_ARG (status 1), nothing written."""
    from art_planner_amd import _capi
    L = _capi.load()
    for h, w in ((1, 8), (8, 1), (1, 1)):
        img = np.full((h, w), 9, np.uint8)
        mask = np.zeros((h, w), np.uint8)
        mask.flat[0] = 1
        out = np.full((h, w), 123, np.uint8)
        rc = L.artp_telea_inpaint_u8(img.ctypes.data, mask.ctypes.data, h, w, 3, out.ctypes.data)
        assert rc != 0 and (out == 123).all()
    two = np.array([[10, 10], [10, 10]], np.uint8)
    m2 = np.array([[0, 1], [0, 0]], np.uint8)
    assert abs(int(_telea(two, m2)[0, 1]) - 10) <= 1     # the smallest image the march accepts
