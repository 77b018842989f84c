"""CPU suite: the oracle restatement against the committed golden vectors (real patched ODE
outputs) and, where oracle/_ref was built, against the real ODE on fresh random poses."""
import numpy as np
import pytest

import common
import golden_io
import oracle_py as O


@pytest.mark.parametrize("name", golden_io.MAPS)
def test_oracle_boxes_match_reference_golden(name):
    gm, combos = golden_io.load_boxes(name)
    assert combos
    for cname, c in combos.items():
        of = O.OracleField(gm[c["layer"]], gm.len_x, gm.len_y, gm.pos_x, gm.pos_y)
        hit, ec, _ = of.check_boxes(c["side"], c["poses"], True)
        assert np.array_equal(hit, c["hit"]), f"{name}/{cname}: {(hit != c['hit']).sum()} label mismatches"
        assert np.array_equal(ec, c["exit"]), f"{name}/{cname}: exit codes changed"


@pytest.mark.parametrize("name", golden_io.MAPS)
def test_oracle_states_match_reference_golden(name):
    gm, _ = golden_io.load_boxes(name)
    om = O.OracleMap(gm)
    for rname, s in golden_io.load_states(name).items():
        valid = om.states_valid(O.robot(rname), s["se3"])
        assert np.array_equal(valid, s["valid"]), f"{name}/{rname}"


@pytest.mark.parametrize("name", golden_io.MAPS)
def test_oracle_edges_match_golden(name):
    gm, _ = golden_io.load_boxes(name)
    om = O.OracleMap(gm)
    for rname, e in golden_io.load_edges(name).items():
        rob = O.robot(rname)
        assert abs(om.z_extent(rob) - e["z_extent"]) < 1e-12
        cm, _ = om.check_motions(rob, e["s1"][:200], e["s2"][:200])
        assert np.array_equal(cm, e["check_motion"][:200])
        assert np.array_equal(om.segment_counts(rob, e["s1"], e["s2"]), e["nd"])
        ei, nint = om.edges_interp_valid(rob, e["s1"], e["s2"])
        assert np.array_equal(ei, e["interp_valid"])
        assert np.array_equal(nint, e["n_interp"])


@pytest.mark.parametrize("name", golden_io.MAPS)
def test_oracle_matches_bulk_reference_golden(name):
    """SURVEY.md 8c volumes: 20 000 dPoses per (map, box) (hit bits AND exit codes), 20 000 states and 2 000 edges per
    (map, robot), all labelled by the real patched ODE (tests/golden/make_golden_bulk.py); inputs regenerated from
    seeds and checksum-verified by golden_io.load_bulk."""
    gm, boxes, states, edges = golden_io.load_bulk(name)
    assert len(boxes) >= 4
    for cname, c in boxes.items():
        assert len(c["poses"]) == golden_io.BULK_POSES
        hit, ec, _ = O.OracleField(gm[c["layer"]], gm.len_x, gm.len_y, gm.pos_x, gm.pos_y).check_boxes(c["side"], c["poses"], True)
        assert np.array_equal(hit, c["hit"]), f"{name}/{cname}: {(hit != c['hit']).sum()} label mismatches"
        assert np.array_equal(ec, c["exit"]), f"{name}/{cname}: exit codes changed"
    om = O.OracleMap(gm)
    for rname in ("yaml", "defaults"):
        rob = O.robot(rname)
        assert len(states[rname]["se3"]) == golden_io.BULK_STATES
        assert np.array_equal(om.states_valid(rob, states[rname]["se3"]), states[rname]["valid"]), f"{name}/{rname}"
        e = edges[rname]
        assert len(e["s1"]) == golden_io.BULK_EDGES
        ei, nint = om.edges_interp_valid(rob, e["s1"], e["s2"])
        assert np.array_equal(ei, e["interp_valid"]) and np.array_equal(nint, e["n_interp"])
        assert np.array_equal(om.segment_counts(rob, e["s1"], e["s2"]), e["nd"])
        cm, _ = om.check_motions(rob, e["s1"][:400], e["s2"][:400])   # ~50 states per edge: a slice keeps the CPU suite short
        assert np.array_equal(cm, e["check_motion"][:400])


def test_flat_field_known_answers():
    """Known answers established on the real ODE during the survey: flat field,
    box half-height 0.1: hit for centre z in {0.09, 0, -0.09}; no hit for +-0.11, +-0.5, 0.1
    (-0.11 demonstrates art_planner's 'under = no collision' patch)."""
    lay = np.zeros((100, 100), np.float32, order="F")
    of = O.OracleField(lay, 10.0, 10.0)
    zs = [0.09, 0.0, -0.09, 0.11, -0.11, 0.5, -0.5, 0.1]
    P = np.zeros((len(zs), 16), np.float32)
    P[:, 2] = zs
    P[:, 4] = P[:, 9] = P[:, 14] = 1
    assert list(of.check_boxes([1.05, 0.55, 0.2], P)) == [1, 1, 1, 0, 0, 0, 0, 0]


def test_interpolate_endpoints_and_unit_norm():
    rng = np.random.default_rng(3)
    gm, _ = golden_io.load_boxes("flat100")
    s = common.random_states(gm, 20, rng)
    for a, b in zip(s[:10], s[10:]):
        assert np.allclose(O.interpolate(a, b, 0.0), a, atol=1e-12)
        e = O.interpolate(a, b, 1.0)
        assert np.allclose(e[:3], b[:3], atol=1e-12)
        assert min(np.abs(e[3:] - b[3:]).max(), np.abs(e[3:] + b[3:]).max()) < 1e-9
        m = O.interpolate(a, b, 0.37)
        assert abs(np.linalg.norm(m[3:]) - 1.0) < 1e-9


def test_check_motion_last_valid_overload_matches_first_overload():
    """The second checkMotion overload (lastValid) of the restated DiscreteMotionValidator: same verdicts as the
    first overload on the golden edges (which carry the first overload's verdicts); on a failing edge the state at
    lastValid.second is the returned state, and when lastValid.second > 0 it is a valid state whose successor on
    the discretised segment (or s2) is invalid."""
    for name in ("perlin64", "slab120"):
        gm, _ = golden_io.load_boxes(name)
        om = O.OracleMap(gm)
        for kind in ("yaml", "defaults"):
            e = golden_io.load_edges(name)[kind]
            rob = O.robot(kind)
            s1, s2 = e["s1"][:400], e["s2"][:400]
            ok, t, st = om.check_motions_last_valid(rob, s1, s2)
            assert np.array_equal(ok, e["check_motion"][:400])
            nd = e["nd"][:400].astype(np.int64)
            bad = np.flatnonzero(ok == 0)
            assert len(bad) > 10
            for i in bad[:60]:
                j = int(round(t[i] * nd[i]))            # lastValid.second = j / nd
                assert 0 <= j <= nd[i] - 1 and abs(t[i] - j / nd[i]) < 1e-15
                assert np.allclose(st[i], O.interpolate(s1[i], s2[i], t[i]), atol=0, rtol=0)
                nxt = s2[i] if j + 1 == nd[i] else O.interpolate(s1[i], s2[i], (j + 1) / nd[i])
                assert om.states_valid(rob, nxt[None])[0] == 0
                if j >= 1:
                    assert om.states_valid(rob, st[i][None])[0] == 1


def test_sampler_oracle_properties(big_map):
    """R6: samples are cell centres of cells with non-zero sample probability, z near the terrain,
    unit quaternions; pure function of (seed, index)."""
    smp = O.OracleSampler(big_map)
    rob = O.robot("yaml")
    se3, rc = smp.sample(rob, 42, 0, 5000)
    se3b, rcb = smp.sample(rob, 42, 1000, 100)
    assert np.array_equal(se3[1000:1100], se3b) and np.array_equal(rc[1000:1100], rcb)
    prob = big_map["sample_probability"]
    assert (prob[rc[:, 0], rc[:, 1]] > 0).all()
    assert np.allclose(np.linalg.norm(se3[:, 3:], axis=1), 1.0, atol=1e-12)
    elev = big_map["elevation"][rc[:, 0], rc[:, 1]]
    assert np.abs(se3[:, 2] - elev).max() <= 0.5 * rob.reach_z + 1e-9
    # row marginal follows the CDF (coarse chi-square-free check on quartiles of rows)
    rowp = prob.sum(1) / prob.sum()
    emp = np.bincount(rc[:, 0], minlength=big_map.rows) / len(rc)
    q = np.add.reduceat(rowp, [0, 100, 200, 300])
    qe = np.add.reduceat(emp, [0, 100, 200, 300])
    assert np.abs(q - qe).max() < 0.03


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref (real ODE) only exists in the build container")
def test_oracle_vs_real_ode_random(big_map):
    rng = np.random.default_rng(7)
    rob = O.robot("yaml")
    for side, layer, zoff, n in [(rob.torso, "elevation", (0.42, 0.12), 20000),
                                 (rob.foot, "elevation_masked", (0.0, 0.08), 60000)]:
        P = common.random_dposes(big_map, n, rng, zoff, tilt=0.3)
        of = O.OracleField(big_map[layer], big_map.len_x, big_map.len_y)
        ref = O.RefChecker(side, big_map[layer], big_map.len_x, big_map.len_y)
        assert np.array_equal(of.check_boxes(side, P), ref.check(P))
        ref.close()
