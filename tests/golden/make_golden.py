#!/usr/bin/env python3
"""Generate the committed golden fixtures from the REAL reference (patched ODE).

Run in the build container only (needs /root/reference and `make -C oracle ref`):

    python tests/golden/make_golden.py

Outputs (data only: inputs + expected outputs, no reference source text):
  tests/golden/boxes_<map>.npz   maps + dPoses (position + float32 quaternion) + hit bits produced by
                                 dCollide(box, heightfield, 1, ...) of the reference's ODE, driven like
                                 HeightMapBoxChecker (oracle/ref_driver.cpp)
  tests/golden/states_<map>.npz  SE3 states + validity labels + per-box exit codes; labels come from
                                 the reference ODE applied to the five dPoses of every state with the
                                 body/feet logic of validity_checker*.cpp
  tests/golden/edges_<map>.npz   state pairs + checkMotion / 0.5 m-interpolation labels (oracle;
                                 every interior state cross-checked against the reference ODE)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import common  # noqa: E402
import oracle_py as O  # noqa: E402
from synthetic import make_map  # noqa: E402


def quat_dposes(pos, quat):
    """dPose (n,16) float32 from float32 position + float32 quaternion (w,x,y,z): pure float32
    + - * arithmetic, bit-reproducible everywhere."""
    R = common.quat_to_R_f32(quat[:, 0], quat[:, 1], quat[:, 2], quat[:, 3])
    P = np.zeros((pos.shape[0], 16), np.float32)
    P[:, 0:3] = pos
    P[:, 4:7] = R[:, 0]
    P[:, 8:11] = R[:, 1]
    P[:, 12:15] = R[:, 2]
    return P


def random_pose_params(gm, n, rng, z_off, tilt=0.45, layer="elevation"):
    x = gm.pos_x + rng.uniform(-gm.len_x * 0.6, gm.len_x * 0.6, n)
    y = gm.pos_y + rng.uniform(-gm.len_y * 0.6, gm.len_y * 0.6, n)
    z = common.elevation_at(gm, x, y, layer) + z_off[0] + rng.normal(0, 1, n) * z_off[1]
    w, qx, qy, qz = common.rpy_to_quat(rng.uniform(-tilt, tilt, n), rng.uniform(-tilt, tilt, n),
                                       rng.uniform(-np.pi, np.pi, n))
    pos = np.stack([x, y, z], 1).astype(np.float32)
    quat = np.stack([w, qx, qy, qz], 1).astype(np.float32)
    return pos, quat


def engineered(gm, side, layer, rng):
    """Edge cases: exactly resting +-ulps, map border straddling, centre outside, identity rotation."""
    pos, quat = [], []
    half = np.float32(side[2]) * np.float32(0.5)
    for _ in range(300):
        x = gm.pos_x + rng.uniform(-gm.len_x * 0.45, gm.len_x * 0.45)
        y = gm.pos_y + rng.uniform(-gm.len_y * 0.45, gm.len_y * 0.45)
        h = np.float32(common.elevation_at(gm, np.array([x]), np.array([y]), layer)[0])
        z = np.float32(h + half)
        for k in (-3, -1, 0, 1, 3):  # resting within a few ulps
            zz = z
            for _ in range(abs(k)):
                zz = np.nextafter(zz, np.float32(np.inf if k > 0 else -np.inf), dtype=np.float32)
            pos.append([x, y, zz])
            yaw = rng.choice([0.0, np.pi / 2, rng.uniform(-np.pi, np.pi)])
            quat.append([np.cos(yaw / 2), 0.0, 0.0, np.sin(yaw / 2)])
    for _ in range(400):  # border straddling / outside
        edge = rng.integers(0, 4)
        t = rng.uniform(-0.5, 0.5)
        d = rng.uniform(-0.8, 0.8)
        if edge == 0:
            x, y = gm.pos_x + gm.len_x / 2 + d, gm.pos_y + t * gm.len_y
        elif edge == 1:
            x, y = gm.pos_x - gm.len_x / 2 + d, gm.pos_y + t * gm.len_y
        elif edge == 2:
            x, y = gm.pos_x + t * gm.len_x, gm.pos_y + gm.len_y / 2 + d
        else:
            x, y = gm.pos_x + t * gm.len_x, gm.pos_y - gm.len_y / 2 + d
        z = common.elevation_at(gm, np.array([x]), np.array([y]), layer)[0] + rng.uniform(-0.1, 0.3)
        pos.append([x, y, z])
        w, qx, qy, qz = common.rpy_to_quat(rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3),
                                           rng.uniform(-np.pi, np.pi))
        quat.append([w, qx, qy, qz])
    return np.array(pos, np.float32), np.array(quat, np.float32)


def build_maps():
    big = make_map(400, 0.04, seed=1234)
    perlin64 = common.crop_map(big, 150, 60, 64)
    # make sure the crop holds -inf patches and add a NaN layer
    m = perlin64["elevation_masked"].copy()
    m[10:18, 40:52] = -np.inf
    perlin64.layers["elevation_masked"] = np.asfortranarray(m)
    mn = m.copy()
    mn[30:32, 20:22] = np.nan
    mn[63, 63] = np.nan  # a NaN at the end of some windows' scan order
    perlin64.layers["elevation_nan"] = np.asfortranarray(mn)
    flat = make_map(100, 0.1, flat=True)
    slab = common.slab_slit_map()
    return {"flat100": flat, "perlin64": perlin64, "slab120": slab}


def main():
    assert O.have_ref(), "build the reference first: make -C oracle ref"
    rng = np.random.default_rng(20260926)
    robots = {"yaml": O.robot("yaml"), "defaults": O.robot("defaults")}
    for name, gm in build_maps().items():
        out = {"rows": gm.rows, "cols": gm.cols, "res": gm.res, "pos_x": gm.pos_x, "pos_y": gm.pos_y}
        layer_names = [k for k in ("elevation", "elevation_masked", "elevation_nan") if k in gm.layers]
        for k in layer_names:
            out["layer_" + k] = np.asarray(gm[k], np.float32)
        combos = []
        for rname, rob in robots.items():
            combos.append((f"{rname}_torso", rob.torso, "elevation", (0.35, 0.2)))
            combos.append((f"{rname}_foot", rob.foot, "elevation_masked", (0.02, 0.08)))
            if "elevation_nan" in gm.layers and rname == "yaml":
                combos.append((f"{rname}_foot_nan", rob.foot, "elevation_nan", (0.02, 0.08)))
                combos.append((f"{rname}_torso_nan", rob.torso, "elevation_nan", (0.2, 0.2)))
        for cname, side, layer, zoff in combos:
            p1, q1 = random_pose_params(gm, 3000, rng, zoff, layer=layer)
            p2, q2 = engineered(gm, side, layer, rng)
            pos, quat = np.concatenate([p1, p2]), np.concatenate([q1, q2])
            P = quat_dposes(pos, quat)
            ref = O.RefChecker(side, gm[layer], gm.len_x, gm.len_y, gm.pos_x, gm.pos_y)
            hit_ref = ref.check(P)
            ref.close()
            of = O.OracleField(gm[layer], gm.len_x, gm.len_y, gm.pos_x, gm.pos_y)
            hit_o, ec, nv = of.check_boxes(side, P, True)
            assert np.array_equal(hit_ref, hit_o), (name, cname, int((hit_ref != hit_o).sum()))
            out[f"{cname}__side"] = np.asarray(side, np.float32)
            out[f"{cname}__layer"] = np.array(layer)
            out[f"{cname}__pos"] = pos
            out[f"{cname}__quat"] = quat
            out[f"{cname}__hit"] = np.packbits(hit_ref)
            out[f"{cname}__exit"] = ec
            print(f"{name:9s} {cname:18s} n={len(pos)} hit={hit_ref.mean():.3f} exits={np.bincount(ec, minlength=9)}")
        np.savez_compressed(os.path.join(HERE, f"boxes_{name}.npz"), **out)

        # ---- full states: labels from the reference ODE at the dPose boundary -------------------
        sout = {}
        eout = {}
        om = O.OracleMap(gm)
        for rname, rob in robots.items():
            se3 = common.random_states(gm, 3000, rng, spread=0.56)
            poses, inside = om.state_poses(rob, se3)
            refb = O.RefChecker(rob.torso, gm["elevation"], gm.len_x, gm.len_y, gm.pos_x, gm.pos_y)
            reff = O.RefChecker(rob.foot, gm["elevation_masked"], gm.len_x, gm.len_y, gm.pos_x, gm.pos_y)
            hb = refb.check(poses[:, 0])
            hf = np.stack([reff.check(poses[:, 1 + k]) for k in range(4)], 1)
            body_ok = np.where(inside[:, 0] != 0, hb == 0, True)
            feet_ok = np.where(inside[:, 1:] != 0, hf != 0, not rob.unknown_space_untraversable)
            label_ref = (body_ok & feet_ok.all(1)).astype(np.uint8)
            label_o = om.states_valid(rob, se3)
            assert np.array_equal(label_ref, label_o), (name, rname)
            det = np.stack([om.state_detail(rob, s)[1] for s in se3]).astype(np.int8)
            sout[f"{rname}__se3"] = se3
            sout[f"{rname}__valid"] = np.packbits(label_ref)
            sout[f"{rname}__detail"] = det
            print(f"{name:9s} states {rname:9s} valid={label_ref.mean():.3f}")

            # ---- edges ----------------------------------------------------------------------------
            idx = rng.permutation(len(se3))
            a = se3[idx[:600]]
            b = a.copy()
            b[:, 0] += rng.uniform(-1.2, 1.2, 600)
            b[:, 1] += rng.uniform(-1.2, 1.2, 600)
            b[:, 2] = common.elevation_at(gm, b[:, 0], b[:, 1]) + rng.normal(0, 0.03, 600)
            w, qx, qy, qz = common.rpy_to_quat(rng.uniform(-0.15, 0.15, 600), rng.uniform(-0.15, 0.15, 600),
                                               rng.uniform(-np.pi, np.pi, 600))
            b[:, 3], b[:, 4], b[:, 5], b[:, 6] = qx, qy, qz, w
            cm, nchk = om.check_motions(rob, a, b)
            nd = om.segment_counts(rob, a, b)
            ei, nint = om.edges_interp_valid(rob, a, b)
            # every state either rule evaluates goes through the REFERENCE ODE (its five dPoses, the body / feet
            # logic of validity_checker*.cpp): the edge verdicts folded from those labels must equal the oracle's
            def ref_labels(states):
                poses, inside = om.state_poses(rob, states)
                hb_ = refb.check(poses[:, 0])
                hf_ = np.stack([reff.check(poses[:, 1 + k]) for k in range(4)], 1)
                b_ok = np.where(inside[:, 0] != 0, hb_ == 0, True)
                f_ok = np.where(inside[:, 1:] != 0, hf_ != 0, not rob.unknown_space_untraversable)
                return b_ok & f_ok.all(1)
            cm_states, cm_edge, ei_states, ei_edge = [b], [np.arange(len(a))], [], []
            for e_i in range(len(a)):
                if nd[e_i] >= 2:   # DiscreteMotionValidator: s2, then t = j / nd, j = 1 .. nd-1
                    ts = np.arange(1, nd[e_i]) / float(nd[e_i])
                    cm_states.append(np.stack([O.interpolate(a[e_i], b[e_i], t) for t in ts]))
                    cm_edge.append(np.full(len(ts), e_i))
                if nint[e_i] >= 1:  # addValidMilestone: t = step * (1 / (n_interp + 1)), step = 1 .. n_interp
                    div = 1.0 / (nint[e_i] + 1)
                    ei_states.append(np.stack([O.interpolate(a[e_i], b[e_i], st * div)
                                               for st in range(1, nint[e_i] + 1)]))
                    ei_edge.append(np.full(nint[e_i], e_i))
            lab = ref_labels(np.concatenate(cm_states))
            cm_ref = np.ones(len(a), bool)
            np.logical_and.at(cm_ref, np.concatenate(cm_edge), lab)
            assert np.array_equal(cm_ref.astype(np.uint8), cm), (name, rname, "checkMotion vs reference ODE")
            ei_ref = np.ones(len(a), bool)
            if ei_states:
                np.logical_and.at(ei_ref, np.concatenate(ei_edge), ref_labels(np.concatenate(ei_states)))
            assert np.array_equal(ei_ref.astype(np.uint8), ei), (name, rname, "interpolation rule vs reference ODE")
            n_checked_ode = len(lab) + sum(len(x) for x in ei_states)
            eout[f"{rname}__s1"] = a
            eout[f"{rname}__s2"] = b
            eout[f"{rname}__check_motion"] = np.packbits(cm)
            eout[f"{rname}__nd"] = nd
            eout[f"{rname}__interp_valid"] = np.packbits(ei)
            eout[f"{rname}__n_interp"] = nint
            eout[f"{rname}__z_extent"] = np.float64(om.z_extent(rob))
            refb.close()
            reff.close()
            print(f"{name:9s} edges  {rname:9s} checkMotion={cm.mean():.3f} interp={ei.mean():.3f} "
                  f"nd mean={nd.mean():.1f}; {n_checked_ode} edge states labelled by the reference ODE")
        np.savez_compressed(os.path.join(HERE, f"states_{name}.npz"), **sout)
        np.savez_compressed(os.path.join(HERE, f"edges_{name}.npz"), **eout)


if __name__ == "__main__":
    main()
