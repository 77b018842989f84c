"""Python view of one artp_ctx (one GPU).  Thin plumbing over the C ABI: numpy for host buffers,
torch only for device memory / streams in the `_dev` calls.  All compute is in libartp.so (HIP)."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _capi


def make_params(kind="yaml", **overrides) -> _capi.Params:
    L = _capi.load()
    p = _capi.Params()
    (L.artp_params_yaml if kind == "yaml" else L.artp_params_defaults)(C.byref(p))
    for k, v in overrides.items():
        setattr(p, k, v)
    return p


def _f32F(a) -> np.ndarray:
    return np.asfortranarray(a, dtype=np.float32)


class Context:
    """Mirrors what art_planner::Planner owns for the hot path: the validity checker's two height
    fields, the sampler's layers and the state-space bounds (art_planner/src/planner.cpp:75-163)."""

    def __init__(self, device: int = 0, params="yaml", lib=None):
        """lib: path of another build of the library (e.g. _capi.VARIANTS_LIB_PATH); default libartp.so."""
        self.L = _capi.load(lib)
        self.params = params if isinstance(params, _capi.Params) else make_params(params)
        h = C.c_void_p()
        rc = self.L.artp_create(device, C.byref(self.params), C.byref(h))
        _capi.check(None, rc, "artp_create", self.L)
        self.h = h
        self.device = device

    @classmethod
    def borrowed(cls, handle, device: int, params) -> "Context":
        """View of an artp_ctx somebody else owns (a device group's member): close() does not destroy it."""
        self = cls.__new__(cls)
        self.L = _capi.load()
        self.params = params
        self.h = C.c_void_p(handle)
        self.device = device
        self._borrowed = True
        return self

    def close(self):
        if getattr(self, "h", None):
            if not getattr(self, "_borrowed", False):
                self.L.artp_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def arch(self) -> str:
        return self.L.artp_device_arch(self.h).decode()

    def _chk(self, rc, what):
        _capi.check(self.h, rc, what, self.L)

    # ---- map ---------------------------------------------------------------------------------
    def upload_layer(self, slot, layer, len_x, len_y, pos_x=0.0, pos_y=0.0):
        layer = _f32F(layer)
        self._chk(self.L.artp_upload_layer(self.h, slot, layer.ctypes.data, layer.shape[0], layer.shape[1],
                                           len_x, len_y, pos_x, pos_y), "artp_upload_layer")

    def update_layer_rect(self, slot, patch, row0, col0):
        patch = _f32F(patch)
        self._chk(self.L.artp_update_layer_rect(self.h, slot, patch.ctypes.data, row0, col0,
                                                patch.shape[0], patch.shape[1]), "artp_update_layer_rect")

    def update_layer_rects(self, slot, patches, origins):
        """artp_update_layer_rects: patches = list of 2-D arrays, origins = list of (row0, col0)."""
        ps = [_f32F(p) for p in patches]
        ptrs = (C.c_void_p * len(ps))(*[p.ctypes.data for p in ps])
        rects = np.array([[r0, c0, p.shape[0], p.shape[1]] for p, (r0, c0) in zip(ps, origins)], np.int32)
        self._chk(self.L.artp_update_layer_rects(self.h, slot, len(ps), C.addressof(ptrs), rects.ctypes.data),
                  "artp_update_layer_rects")

    def upload_map(self, gm, body_layer="elevation", feet_layer="elevation_masked", sampler=True):
        """Planner::setMap (planner.cpp:135-163): both height fields, the sampler layers and the
        z bounds (min/max finite elevation -/+ reach.z/2)."""
        self.upload_layer(0, gm[body_layer], gm.len_x, gm.len_y, gm.pos_x, gm.pos_y)
        self.upload_layer(1, gm[feet_layer], gm.len_x, gm.len_y, gm.pos_x, gm.pos_y)
        elev = gm[body_layer]
        fin = elev[np.isfinite(elev)]
        lo = float(fin.min()) if fin.size else 0.0
        hi = float(fin.max()) if fin.size else 0.0
        self.set_z_bounds(lo - self.params.reach_z / 2, hi + self.params.reach_z / 2)
        if sampler and "cum_prob" in gm.layers:
            ls = [_f32F(gm["cum_prob"]), np.ascontiguousarray(gm["cum_prob_rowwise"], np.float32),
                  _f32F(gm[body_layer]), _f32F(gm["normal_x"]), _f32F(gm["normal_y"]),
                  _f32F(gm["normal_z"]), _f32F(gm["plane_fit_std_dev"])]
            self._chk(self.L.artp_upload_sampler_layers(self.h, *[a.ctypes.data for a in ls], gm.rows,
                                                        gm.cols, gm.len_x, gm.len_y, gm.pos_x, gm.pos_y),
                      "artp_upload_sampler_layers")

    # ---- host-buffer entry points ------------------------------------------------------------
    def check_boxes(self, slot, box, poses, want_exit_codes=False):
        box = np.ascontiguousarray(box, np.float32)
        poses = np.ascontiguousarray(poses, np.float32).reshape(-1, 16)
        n = poses.shape[0]
        hit = np.empty(n, np.uint8)
        ec = np.empty(n, np.uint8) if want_exit_codes else None
        self._chk(self.L.artp_check_boxes(self.h, slot, box.ctypes.data, poses.ctypes.data, n,
                                          hit.ctypes.data, ec.ctypes.data if want_exit_codes else None),
                  "artp_check_boxes")
        return (hit, ec) if want_exit_codes else hit

    def validate_states(self, se3, want_detail=False):
        se3 = np.ascontiguousarray(se3, np.float64).reshape(-1, 7)
        n = se3.shape[0]
        valid = np.empty(n, np.uint8)
        detail = np.empty((n, 6), np.int8) if want_detail else None
        self._chk(self.L.artp_validate_states(self.h, se3.ctypes.data, n, valid.ctypes.data,
                                              detail.ctypes.data if want_detail else None),
                  "artp_validate_states")
        return (valid, detail) if want_detail else valid

    def sample_states(self, seed, first_index, n):
        out = np.empty((n, 7), np.float64)
        self._chk(self.L.artp_sample_states(self.h, seed, first_index, n, out.ctypes.data),
                  "artp_sample_states")
        return out

    def set_persistent_latency(self, enabled=True):
        """<= 16 states per host call through ONE resident workgroup polling a mailbox (no launch per call); off by default."""
        self._chk(self.L.artp_set_persistent_latency(self.h, 1 if enabled else 0), "artp_set_persistent_latency")

    def persistent_latency_stats(self):
        out = (C.c_uint64 * 2)()
        self._chk(self.L.artp_persistent_latency_stats(self.h, C.byref(out)), "artp_persistent_latency_stats")
        return {"launches": int(out[0]), "requests": int(out[1])}

    def set_few_edges(self, enabled=True):
        """<= 64 edges per host call: the one-launch latency kernel (default) or the batch pipeline (enabled=False)."""
        self._chk(self.L.artp_set_few_edges(self.h, 1 if enabled else 0), "artp_set_few_edges")

    def set_edge_passes(self, two_pass=True, coarse_stride=0):
        self._chk(self.L.artp_set_edge_passes(self.h, 1 if two_pass else 0, coarse_stride), "artp_set_edge_passes")

    def check_motions(self, s1, s2):
        s1 = np.ascontiguousarray(s1, np.float64).reshape(-1, 7)
        s2 = np.ascontiguousarray(s2, np.float64).reshape(-1, 7)
        valid = np.empty(s1.shape[0], np.uint8)
        self._chk(self.L.artp_check_motions(self.h, s1.ctypes.data, s2.ctypes.data, s1.shape[0],
                                            valid.ctypes.data), "artp_check_motions")
        return valid

    def set_z_bounds(self, z_low, z_high):
        """artp_set_z_bounds; the pair is kept as `z_bounds`."""
        self._chk(self.L.artp_set_z_bounds(self.h, float(z_low), float(z_high)), "artp_set_z_bounds")
        self.z_bounds = (float(z_low), float(z_high))

    def set_r3_extent(self, max_extent):
        """checkMotion's R^3 maxExtent fixed to max_extent (0 = follow the installed map and z bounds): artp_set_r3_extent."""
        self._chk(self.L.artp_set_r3_extent(self.h, float(max_extent)), "artp_set_r3_extent")

    def check_motions_last_valid(self, s1, s2):
        """(valid, lastValid.second, *lastValid.first) of checkMotion's second overload; t = 1 / s2 where valid."""
        s1 = np.ascontiguousarray(s1, np.float64).reshape(-1, 7)
        s2 = np.ascontiguousarray(s2, np.float64).reshape(-1, 7)
        n = s1.shape[0]
        valid = np.empty(n, np.uint8)
        t = np.empty(n, np.float64)
        st = np.empty((n, 7), np.float64)
        self._chk(self.L.artp_check_motions_last_valid(self.h, s1.ctypes.data, s2.ctypes.data, n, valid.ctypes.data,
                                                       t.ctypes.data, st.ctypes.data), "artp_check_motions_last_valid")
        return valid, t, st

    def sample_and_validate(self, seed, first_index, n, with_version=False):
        se3 = np.empty((n, 7), np.float64)
        valid = np.empty(n, np.uint8)
        ver = C.c_uint64(0)
        self._chk(self.L.artp_sample_and_validate(self.h, seed, first_index, n, se3.ctypes.data, valid.ctypes.data,
                                                  C.addressof(ver)), "artp_sample_and_validate")
        return (se3, valid, int(ver.value)) if with_version else (se3, valid)

    def map_version(self):
        """artp_map_version: bumped by every call that changes a layer, its tables or the sampler tables."""
        return int(self.L.artp_map_version(self.h))

    def check_edges_interp(self, s1, s2):
        s1 = np.ascontiguousarray(s1, np.float64).reshape(-1, 7)
        s2 = np.ascontiguousarray(s2, np.float64).reshape(-1, 7)
        n = s1.shape[0]
        valid = np.empty(n, np.uint8)
        nint = np.empty(n, np.uint32)
        self._chk(self.L.artp_check_edges_interp(self.h, s1.ctypes.data, s2.ctypes.data, n,
                                                 valid.ctypes.data, nint.ctypes.data),
                  "artp_check_edges_interp")
        return valid, nint

    # ---- device-buffer entry points (torch tensors on this context's GPU) ----------------------
    def use_torch_stream(self):
        import torch
        self._chk(self.L.artp_set_stream(self.h, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)),
                  "artp_set_stream")

    def synchronize(self):
        self._chk(self.L.artp_synchronize(self.h), "artp_synchronize")

    def set_lane(self, lane: int):
        """Switch the current lane (include/artp_c.h artp_set_lane): own stream + scratch, shared map.  Calls issued
        on different lanes overlap on the GPU."""
        self._chk(self.L.artp_set_lane(self.h, int(lane)), "artp_set_lane")

    @property
    def lane(self) -> int:
        return int(self.L.artp_get_lane(self.h))

    def validate_states_dev(self, se3_t, valid_t, detail_t=None):
        n = se3_t.shape[0]
        self._chk(self.L.artp_validate_states_dev(self.h, se3_t.data_ptr(), n, valid_t.data_ptr(),
                                                  detail_t.data_ptr() if detail_t is not None else None),
                  "artp_validate_states_dev")

    def sample_states_dev(self, seed, first_index, n, out_t):
        self._chk(self.L.artp_sample_states_dev(self.h, seed, first_index, n, out_t.data_ptr()),
                  "artp_sample_states_dev")

    def sample_and_validate_dev(self, seed, first_index, n, se3_t, valid_t, count=False) -> Optional[int]:
        cnt = C.c_size_t(0)
        self._chk(self.L.artp_sample_and_validate_dev(self.h, seed, first_index, n, se3_t.data_ptr(),
                                                      valid_t.data_ptr(), C.byref(cnt) if count else None),
                  "artp_sample_and_validate_dev")
        return cnt.value if count else None

    def check_motions_dev(self, s1_t, s2_t, valid_t):
        self._chk(self.L.artp_check_motions_dev(self.h, s1_t.data_ptr(), s2_t.data_ptr(), s1_t.shape[0],
                                                valid_t.data_ptr()), "artp_check_motions_dev")

    def check_motions_last_valid_dev(self, s1_t, s2_t, valid_t, last_t_t, last_state_t=None):
        self._chk(self.L.artp_check_motions_last_valid_dev(self.h, s1_t.data_ptr(), s2_t.data_ptr(), s1_t.shape[0],
                                                           valid_t.data_ptr(), last_t_t.data_ptr(),
                                                           last_state_t.data_ptr() if last_state_t is not None else None),
                  "artp_check_motions_last_valid_dev")

    def check_edges_interp_dev(self, s1_t, s2_t, valid_t, nint_t=None):
        self._chk(self.L.artp_check_edges_interp_dev(self.h, s1_t.data_ptr(), s2_t.data_ptr(), s1_t.shape[0],
                                                     valid_t.data_ptr(),
                                                     nint_t.data_ptr() if nint_t is not None else None),
                  "artp_check_edges_interp_dev")

    def check_boxes_dev(self, slot, box, poses_t, hit_t, ec_t=None):
        box = np.ascontiguousarray(box, np.float32)
        self._chk(self.L.artp_check_boxes_dev(self.h, slot, box.ctypes.data, poses_t.data_ptr(),
                                              poses_t.shape[0], hit_t.data_ptr(),
                                              ec_t.data_ptr() if ec_t is not None else None),
                  "artp_check_boxes_dev")

    def compact_valid_dev(self, se3_t, valid_t, out_t, count_t):
        """count_t: 1-element int64/uint64 device tensor."""
        self._chk(self.L.artp_compact_valid_dev(self.h, se3_t.data_ptr(), valid_t.data_ptr(), se3_t.shape[0],
                                                out_t.data_ptr(), count_t.data_ptr()), "artp_compact_valid_dev")

    def algorithmic_vertices_dev(self, se3_t) -> int:
        v = C.c_uint64(0)
        self._chk(self.L.artp_algorithmic_vertices_dev(self.h, se3_t.data_ptr(), se3_t.shape[0], C.byref(v)),
                  "artp_algorithmic_vertices_dev")
        return v.value

    def pipeline_counters(self):
        out = (C.c_uint64 * 8)()
        self._chk(self.L.artp_debug_pipeline_counters(self.h, C.byref(out)), "artp_debug_pipeline_counters")
        return {"torso_queued": out[0], "feet_queued": out[4], "exact_grouping": out[1], "feet_plane_stage": out[5],
                "feet_partner_pass": out[6],
                "torso_staged_pass": out[2]}

    def partner_table(self, slot, shape):
        """(flags[nD, nW] uint8 in ODE sample layout, radius) of a layer's partner table; (None, 0) if
        the layer has none."""
        r = C.c_int(0)
        self._chk(self.L.artp_debug_partner_table(self.h, slot, None, 0, C.byref(r)), "artp_debug_partner_table")
        if r.value == 0:
            return None, 0
        out = np.empty(shape[0] * shape[1], np.uint8)
        self._chk(self.L.artp_debug_partner_table(self.h, slot, out.ctypes.data, out.size, C.byref(r)),
                  "artp_debug_partner_table")
        return out.reshape(shape[1], shape[0]), r.value

    # ---- device-side map preprocessing (N2) -----------------------------------------------------
    def preprocess_map(self, elevation, len_x, len_y, pos_x=0.0, pos_y=0.0, traversability=None, kind="yaml",
                       observed=None, vertices=None, **overrides):
        """The new-map processor chain (Basic, [inverse vertex density], base distribution, [unknown cap], CDF)
        on the device; returns a PreprocessedMap."""
        p = _capi.PreprocessParams()
        (self.L.artp_preprocess_params_yaml if kind == "yaml" else self.L.artp_preprocess_params_defaults)(C.byref(p))
        for k, v in overrides.items():
            setattr(p, k, v)
        e = _f32F(elevation)
        t = _f32F(traversability) if traversability is not None else None
        o = _f32F(observed) if observed is not None else None
        vs = np.ascontiguousarray(vertices, np.float64).reshape(-1, 7) if vertices is not None else None
        inp = _capi.PreprocessInputs(e.ctypes.data, t.ctypes.data if t is not None else None,
                                     o.ctypes.data if o is not None else None,
                                     vs.ctypes.data if vs is not None else None, 0 if vs is None else vs.shape[0],
                                     e.shape[0], e.shape[1], len_x, len_y, pos_x, pos_y)
        h = C.c_void_p()
        self._chk(self.L.artp_preprocess_map_ex(self.h, C.byref(inp), C.byref(p), C.byref(h)), "artp_preprocess_map_ex")
        pm = PreprocessedMap(self, h, e.shape)
        pm.geom = (len_x, len_y, pos_x, pos_y)
        pm.params = p
        return pm

    def inpaint_layer(self, layer, mode=0):
        """(filled layer, number of holes): artp_inpaint_layer, mode 0 = inpaintMatrix, 1 = the cost node's."""
        a = _f32F(layer)
        out = np.empty(a.shape, np.float32, order="F")
        n = C.c_uint64(0)
        self._chk(self.L.artp_inpaint_layer(self.h, a.ctypes.data, a.shape[0], a.shape[1], mode, out.ctypes.data,
                                            C.byref(n)), "artp_inpaint_layer")
        return out, n.value

    def cost_set_external_query(self, fn=None):
        """The MotionCostFunc seam (prm_motion_cost.cpp:27-73): fn(edges [B, 6] float32) -> costs [B, 3] (or None = the
        call failed) prices every learned-cost batch of this context's roadmaps; fn=None: device pricing."""
        if fn is None:
            self._ext_cost = None
            self._chk(self.L.artp_cost_set_external_query(self.h, _capi.COST_QUERY_FN(0), None), "artp_cost_set_external_query")
            return

        def thunk(_user, edges, b, cost):
            try:
                out = fn(np.ctypeslib.as_array(edges, shape=(b, 6)).copy())
                if out is None:
                    return 1
                np.ctypeslib.as_array(cost, shape=(b, 3))[:] = np.asarray(out, np.float32).reshape(b, 3)
                return 0
            except Exception:  # noqa: BLE001 -- an exception must not cross the C boundary
                return 1

        self._ext_cost = _capi.COST_QUERY_FN(thunk)   # kept alive as long as it is installed
        self._chk(self.L.artp_cost_set_external_query(self.h, self._ext_cost, None), "artp_cost_set_external_query")

    def cost_set_fc_path(self, mfma=True):
        self._chk(self.L.artp_cost_set_fc_path(self.h, 1 if mfma else 0), "artp_cost_set_fc_path")

    def cost_set_hole_filling(self, enabled=True):
        self._chk(self.L.artp_cost_set_hole_filling(self.h, 1 if enabled else 0), "artp_cost_set_hole_filling")

    # ---- learned motion cost (R8 / R9) ---------------------------------------------------------
    def cost_load_weights(self, blob: bytes):
        buf = (C.c_char * len(blob)).from_buffer_copy(blob)
        self._chk(self.L.artp_cost_load_weights(self.h, buf, len(blob)), "artp_cost_load_weights")

    def cost_update_map(self, elev_xy, res, len_x, len_y, cx=0.0, cy=0.0):
        a = np.ascontiguousarray(elev_xy, np.float32)
        self._chk(self.L.artp_cost_update_map(self.h, a.ctypes.data, a.shape[0], a.shape[1], res, len_x, len_y,
                                              cx, cy), "artp_cost_update_map")

    def cost_update_map_dev(self, elev_xy_t, res, len_x, len_y, cx=0.0, cy=0.0):
        """elev_xy_t: float32 device tensor [rows, cols] (row-major); asynchronous."""
        self._chk(self.L.artp_cost_update_map_dev(self.h, elev_xy_t.data_ptr(), elev_xy_t.shape[0], elev_xy_t.shape[1],
                                                  res, len_x, len_y, cx, cy), "artp_cost_update_map_dev")

    def cost_update_map_layer(self, layer, res, len_x, len_y, pos_x=0.0, pos_y=0.0):
        """From the planner's grid_map layer (the cost server's re-indexing is applied inside)."""
        a = _f32F(layer)
        self._chk(self.L.artp_cost_update_map_layer(self.h, a.ctypes.data, a.shape[0], a.shape[1], res, len_x, len_y,
                                                    pos_x, pos_y), "artp_cost_update_map_layer")

    def cost_query(self, edges):
        e = np.ascontiguousarray(edges, np.float32).reshape(-1, 6)
        out = np.empty((e.shape[0], 3), np.float32)
        self._chk(self.L.artp_cost_query(self.h, e.ctypes.data, e.shape[0], out.ctypes.data), "artp_cost_query")
        return out

    def cost_query_cells(self, edges):
        """(rows, cols) of the feature-map cells the cost query gathers (artp_cost_debug_query_cells)."""
        e = np.ascontiguousarray(edges, np.float32).reshape(-1, 6)
        rows, cols = np.empty(e.shape[0], np.int32), np.empty(e.shape[0], np.int32)
        self._chk(self.L.artp_cost_debug_query_cells(self.h, e.ctypes.data, e.shape[0], rows.ctypes.data,
                                                     cols.ctypes.data), "artp_cost_debug_query_cells")
        return rows, cols

    def cost_fc_path(self):
        """{'mfma': 0/1, 'selfcheck': -1/0/1, 'max_abs_diff': float} -- which kernel answers cost queries (artp_cost_fc_path)."""
        m, sc, d = C.c_int32(0), C.c_int32(0), C.c_float(0)
        self._chk(self.L.artp_cost_fc_path(self.h, C.byref(m), C.byref(sc), C.byref(d)), "artp_cost_fc_path")
        return {"mfma": int(m.value), "selfcheck": int(sc.value), "max_abs_diff": float(d.value)}

    def cost_query_dev(self, edges_t, cost_t):
        self._chk(self.L.artp_cost_query_dev(self.h, edges_t.data_ptr(), edges_t.shape[0], cost_t.data_ptr()),
                  "artp_cost_query_dev")

    def cost_features(self):
        fh, fw = C.c_int(0), C.c_int(0)
        self._chk(self.L.artp_cost_get_features(self.h, None, C.byref(fh), C.byref(fw)), "artp_cost_get_features")
        out = np.empty((fh.value, fw.value, 48), np.float32)
        self._chk(self.L.artp_cost_get_features(self.h, out.ctypes.data, C.byref(fh), C.byref(fw)),
                  "artp_cost_get_features")
        return out

    def compact_valid_indices_dev(self, valid_t, idx_t, count_t):
        """idx_t: uint32/int32 device tensor [>= n]; count_t: 1-element int64 device tensor."""
        self._chk(self.L.artp_compact_valid_indices_dev(self.h, valid_t.data_ptr(), valid_t.shape[0],
                                                        idx_t.data_ptr(), count_t.data_ptr()),
                  "artp_compact_valid_indices_dev")

    def pack_edge_results_dev(self, valid_t, ei_t, ej_t, cost_t, records_t, count_t):
        """records_t: int32/uint32 device tensor [>= n, 5]; count_t: 1-element int64 device tensor."""
        self._chk(self.L.artp_pack_edge_results_dev(self.h, valid_t.data_ptr(), ei_t.data_ptr(), ej_t.data_ptr(),
                                                    cost_t.data_ptr(), valid_t.shape[0], records_t.data_ptr(),
                                                    count_t.data_ptr()), "artp_pack_edge_results_dev")

    def pack_valid_bits_dev(self, valid_t, bits_t):
        """bits_t: int64 device tensor [ceil(n / 64)]."""
        self._chk(self.L.artp_pack_valid_bits_dev(self.h, valid_t.data_ptr(), valid_t.shape[0], bits_t.data_ptr()),
                  "artp_pack_valid_bits_dev")

    def indices_from_bits_dev(self, bits_t, n, idx_t, count_t):
        """The ascending indices of the set bits among the first n bits; idx_t int32 [>= set bits], count_t int64[1]."""
        self._chk(self.L.artp_indices_from_bits_dev(self.h, bits_t.data_ptr(), n, idx_t.data_ptr(), count_t.data_ptr()),
                  "artp_indices_from_bits_dev")

    def materialise_from_bits_dev(self, seed, gathered_bits_t, prefix_bits, base_indices, cap, out_t, counts_t):
        """artp_materialise_from_bits_dev: gathered_bits_t int64 [n_ranks, words]; base_indices = first global sample
        index of every rank's batch; out_t float64 [n_ranks, cap, 7]; counts_t int64 [n_ranks]."""
        n_ranks, words = gathered_bits_t.shape
        base = np.ascontiguousarray(base_indices, np.uint64)
        self._chk(self.L.artp_materialise_from_bits_dev(self.h, seed, gathered_bits_t.data_ptr(), n_ranks, words, prefix_bits,
                                                        base.ctypes.data, cap, out_t.data_ptr(), counts_t.data_ptr()),
                  "artp_materialise_from_bits_dev")

    def sample_states_at_dev(self, seed, base_index, idx_t, count_t, cap, out_t):
        self._chk(self.L.artp_sample_states_at_dev(self.h, seed, base_index, idx_t.data_ptr(), count_t.data_ptr(),
                                                   cap, out_t.data_ptr()), "artp_sample_states_at_dev")


class PreprocessedMap:
    """Device-resident result of Context.preprocess_map (artp_preprocessed)."""

    def __init__(self, ctx, handle, shape):
        self.ctx, self.h, self.shape = ctx, handle, shape

    def layer(self, name):
        n = self.shape[0] if name == "cum_prob_rowwise" else self.shape[0] * self.shape[1]
        out = np.empty(n, np.float32)
        self.ctx._chk(self.ctx.L.artp_preprocessed_get_layer(self.ctx.h, self.h, name.encode(), out.ctypes.data),
                      "artp_preprocessed_get_layer")
        return out if name == "cum_prob_rowwise" else out.reshape(self.shape[1], self.shape[0]).T

    def change_from(self, old, height_change_for_update=0.05):
        """computeChange against an older PreprocessedMap: (updated layer, (row0, col0, nrows, ncols), count)."""
        upd = np.empty(self.shape[0] * self.shape[1], np.float32)
        rect = (C.c_int * 4)()
        cnt = C.c_uint64(0)
        self.ctx._chk(self.ctx.L.artp_preprocessed_change(self.ctx.h, self.h, old.h, height_change_for_update,
                                                          upd.ctypes.data, C.byref(rect), C.byref(cnt)),
                      "artp_preprocessed_change")
        return upd.reshape(self.shape[1], self.shape[0]).T, tuple(rect), cnt.value

    def install(self):
        self.ctx._chk(self.ctx.L.artp_preprocessed_install(self.ctx.h, self.h), "artp_preprocessed_install")

    def reweight_dev(self, vertices_t=None, install_sampler=True):
        """Map::reApplyPreprocessing for the sampling distribution: inverse density of the given roadmap vertices
        (torch float64 device tensor [n, 7], or None for no density term), new CDF, optionally installed."""
        n = 0 if vertices_t is None else vertices_t.shape[0]
        self.ctx._chk(self.ctx.L.artp_preprocessed_reweight_dev(self.ctx.h, self.h, C.byref(self.params),
                                                                vertices_t.data_ptr() if n else None, n,
                                                                1 if install_sampler else 0),
                      "artp_preprocessed_reweight_dev")

    def close(self):
        if self.h:
            self.ctx.L.artp_preprocessed_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
