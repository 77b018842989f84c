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
    from art_planner_amd.synthetic import make_map
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
    from art_planner_amd.synthetic import make_map
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
