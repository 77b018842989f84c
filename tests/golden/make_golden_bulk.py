#!/usr/bin/env python3
"""Bulk golden vectors at the volumes SURVEY.md 8c asks for: 20 000 dPoses per (map, box), 20 000 states and 2 000
edges per (map, robot), labelled by the REAL patched ODE (oracle/_ref, compiled from /root/reference/ode where it lies).

Build container only.  The inputs are a pure function of (seed, committed map layers) -- tests/golden_io.py regenerates
them and verifies the sha1 stored here -- so bulk_<map>.npz holds the expected outputs only (hit bits, exit codes of the
oracle, state labels, edge verdicts, segment counts): ~90 kB per map instead of ~5 MB.  Every edge state the two edge
rules evaluate goes through the reference ODE, like in make_golden.py; the oracle must agree everywhere."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_io as G  # noqa: E402
import oracle_py as O  # noqa: E402


def main():
    assert O.have_ref(), "build the reference first: make -C oracle ref"
    robots = {"yaml": O.robot("yaml"), "defaults": O.robot("defaults")}
    for mi, name in enumerate(G.MAPS):
        gm, _ = G.load_boxes(name)   # the committed map layers
        out = {}
        combos = []
        for rname, rob in robots.items():
            combos.append((f"{rname}_torso", rob.torso, "elevation", (0.35, 0.2)))
            combos.append((f"{rname}_foot", rob.foot, "elevation_masked", (0.02, 0.08)))
            if "elevation_nan" in gm.layers and rname == "yaml":
                combos.append((f"{rname}_foot_nan", rob.foot, "elevation_nan", (0.02, 0.08)))
                combos.append((f"{rname}_torso_nan", rob.torso, "elevation_nan", (0.2, 0.2)))
        for ci, (cname, side, layer, zoff) in enumerate(combos):
            seed = 910000 + 100 * mi + ci
            P = G.bulk_pose_params(gm, G.BULK_POSES, seed, zoff, layer=layer)
            ref = O.RefChecker(side, gm[layer], gm.len_x, gm.len_y, gm.pos_x, gm.pos_y)
            hit_ref = ref.check(P)
            ref.close()
            hit_o, ec, _ = O.OracleField(gm[layer], gm.len_x, gm.len_y, gm.pos_x, gm.pos_y).check_boxes(side, P, True)
            assert np.array_equal(hit_ref, hit_o), (name, cname, int((hit_ref != hit_o).sum()))
            out[f"{cname}__side"] = np.asarray(side, np.float32)
            out[f"{cname}__layer"] = np.array(layer)
            out[f"{cname}__zoff"] = np.asarray(zoff, np.float64)
            out[f"{cname}__n"], out[f"{cname}__seed"] = np.int64(G.BULK_POSES), np.int64(seed)
            out[f"{cname}__sha1"] = np.array(G.sha1_of(P))
            out[f"{cname}__hit"] = np.packbits(hit_ref)
            out[f"{cname}__exit"] = ec
            print(f"{name:9s} {cname:18s} n={len(P)} hit={hit_ref.mean():.3f} exits={np.bincount(ec, minlength=9)}")
        om = O.OracleMap(gm)
        for ri, (rname, rob) in enumerate(robots.items()):
            refb = O.RefChecker(rob.torso, gm["elevation"], gm.len_x, gm.len_y, gm.pos_x, gm.pos_y)
            reff = O.RefChecker(rob.foot, gm["elevation_masked"], gm.len_x, gm.len_y, gm.pos_x, gm.pos_y)

            def ref_labels(states):
                poses, inside = om.state_poses(rob, states)
                hb = refb.check(poses[:, 0])
                hf = np.stack([reff.check(poses[:, 1 + k]) for k in range(4)], 1)
                b_ok = np.where(inside[:, 0] != 0, hb == 0, True)
                f_ok = np.where(inside[:, 1:] != 0, hf != 0, not rob.unknown_space_untraversable)
                return b_ok & f_ok.all(1)

            sseed = 920000 + 100 * mi + ri
            se3 = G.bulk_states(gm, G.BULK_STATES, sseed)
            lab = ref_labels(se3).astype(np.uint8)
            assert np.array_equal(lab, om.states_valid(rob, se3)), (name, rname)
            out[f"{rname}__states_n"], out[f"{rname}__states_seed"] = np.int64(G.BULK_STATES), np.int64(sseed)
            out[f"{rname}__states_sha1"] = np.array(G.sha1_of(se3))
            out[f"{rname}__valid"] = np.packbits(lab)
            eseed = 930000 + 100 * mi + ri
            a, b = G.bulk_edges(gm, se3, G.BULK_EDGES, eseed)
            cm, _ = om.check_motions(rob, a, b)
            nd = om.segment_counts(rob, a, b)
            ei, nint = om.edges_interp_valid(rob, a, b)
            cm_states, cm_edge, ei_states, ei_edge = [b], [np.arange(len(a))], [], []
            for e in range(len(a)):
                if nd[e] >= 2:   # DiscreteMotionValidator: s2, then t = j / nd, j = 1 .. nd-1
                    ts = np.arange(1, nd[e]) / float(nd[e])
                    cm_states.append(np.stack([O.interpolate(a[e], b[e], t) for t in ts]))
                    cm_edge.append(np.full(len(ts), e))
                if nint[e] >= 1:  # addValidMilestone: t = step * (1 / (n_interp + 1))
                    div = 1.0 / (nint[e] + 1)
                    ei_states.append(np.stack([O.interpolate(a[e], b[e], st * div) for st in range(1, nint[e] + 1)]))
                    ei_edge.append(np.full(nint[e], e))
            cm_ref = np.ones(len(a), bool)
            np.logical_and.at(cm_ref, np.concatenate(cm_edge), ref_labels(np.concatenate(cm_states)))
            assert np.array_equal(cm_ref.astype(np.uint8), cm), (name, rname, "checkMotion vs reference ODE")
            ei_ref = np.ones(len(a), bool)
            if ei_states:
                np.logical_and.at(ei_ref, np.concatenate(ei_edge), ref_labels(np.concatenate(ei_states)))
            assert np.array_equal(ei_ref.astype(np.uint8), ei), (name, rname, "interpolation rule vs reference ODE")
            out[f"{rname}__edges_n"], out[f"{rname}__edges_seed"] = np.int64(G.BULK_EDGES), np.int64(eseed)
            out[f"{rname}__edges_sha1"] = np.array(G.sha1_of(a, b))
            out[f"{rname}__check_motion"] = np.packbits(cm)
            out[f"{rname}__nd"] = nd.astype(np.uint32)
            out[f"{rname}__interp_valid"] = np.packbits(ei)
            out[f"{rname}__n_interp"] = nint.astype(np.uint32)
            refb.close()
            reff.close()
            n_ode = sum(len(x) for x in cm_states) + sum(len(x) for x in ei_states)
            print(f"{name:9s} {rname:9s} states valid={lab.mean():.3f} checkMotion={cm.mean():.3f} interp={ei.mean():.3f}; "
                  f"{n_ode} edge states labelled by the reference ODE")
        np.savez_compressed(os.path.join(HERE, f"bulk_{name}.npz"), **out)


if __name__ == "__main__":
    main()
