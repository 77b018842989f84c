"""ctypes binding of libartp.so (include/artp_c.h).  Fails loudly when the HIP library is missing or
no GPU is present -- there is no CPU fallback anywhere in this package."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libartp.so")
if os.environ.get("ARTP_LIB"):  # tuning builds (e.g. libartp_timing.so); never a CPU path
    LIB_PATH = os.path.abspath(os.environ["ARTP_LIB"])

# every symbol include/artp_c.h declares (tests check the library exports all of them)
SYMBOLS = [
    "artp_params_defaults", "artp_params_yaml", "artp_status_string", "artp_last_error",
    "artp_device_arch", "artp_create", "artp_destroy", "artp_set_stream", "artp_use_own_stream",
    "artp_synchronize", "artp_set_lane", "artp_get_lane", "artp_map_version",
    "artp_upload_layer", "artp_update_layer_rect", "artp_update_layer_rects", "artp_check_boxes", "artp_check_boxes_dev",
    "artp_validate_states", "artp_validate_states_dev", "artp_set_persistent_latency", "artp_persistent_latency_stats", "artp_upload_sampler_layers",
    "artp_sample_states", "artp_sample_states_dev", "artp_sample_and_validate_dev",
    "artp_sample_and_validate", "artp_check_motions_last_valid", "artp_check_motions_last_valid_dev",
    "artp_set_z_bounds", "artp_set_few_edges", "artp_set_edge_passes", "artp_cost_set_fc_path", "artp_check_motions", "artp_check_motions_dev", "artp_check_edges_interp",
    "artp_check_edges_interp_dev", "artp_compact_valid_dev", "artp_compact_valid_indices_dev", "artp_sample_states_at_dev",
    "artp_pack_edge_results_dev", "artp_cost_update_map_dev", "artp_pack_valid_bits_dev", "artp_indices_from_bits_dev",
    "artp_materialise_from_bits_dev",
    "artp_shard_first_index", "artp_group_create", "artp_group_unique_id", "artp_group_create_rank",
    "artp_group_destroy", "artp_group_last_error", "artp_group_world_size", "artp_group_local_count",
    "artp_group_rank", "artp_group_ctx", "artp_group_ranks_seen", "artp_group_configure",
    "artp_group_sample_and_validate_step", "artp_group_step_buffers", "artp_group_exchange_edges",
    "artp_group_edge_buffers", "artp_group_synchronize", "artp_group_abort",
    "artp_algorithmic_vertices_dev",
    "artp_debug_pipeline_counters", "artp_debug_partner_table", "artp_roadmap_params_defaults",
    "artp_roadmap_build", "artp_roadmap_stats", "artp_roadmap_export", "artp_roadmap_solve", "artp_roadmap_destroy",
    "artp_roadmap_revalidate", "artp_roadmap_set_query", "artp_roadmap_simplify_path", "artp_roadmap_grow",
    "artp_roadmap_solve_until", "artp_roadmap_set_density_map", "artp_preprocessed_reweight_dev",
    "artp_preprocess_params_defaults", "artp_preprocess_params_yaml", "artp_preprocess_map",
    "artp_preprocess_map_ex", "artp_preprocessed_change",
    "artp_preprocessed_get_layer", "artp_preprocessed_install", "artp_preprocessed_destroy",
    "artp_inpaint_layer", "artp_cost_set_hole_filling", "artp_cost_set_external_query",
    "artp_cost_blob_bytes", "artp_cost_load_weights", "artp_cost_update_map_layer",
    "artp_cost_update_map", "artp_cost_query", "artp_cost_query_dev", "artp_cost_get_features", "artp_cost_debug_query_cells", "artp_cost_fc_path", "artp_set_r3_extent", "artp_telea_inpaint_u8",
]


# artp_cost_query_fn: int (*)(void* user, const float* edges, size_t b, float* cost)
COST_QUERY_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_float), C.c_size_t, C.POINTER(C.c_float))


class ArtpError(RuntimeError):
    status = None   # the artp_status code (include/artp_c.h) when the error came out of check()


class Params(C.Structure):
    _fields_ = [(n, C.c_double) for n in
                ("torso_length", "torso_width", "torso_height", "torso_off_x", "torso_off_y",
                 "torso_off_z", "feet_off_x", "feet_off_y", "feet_off_z", "reach_x", "reach_y",
                 "reach_z")] + [("unknown_space_untraversable", C.c_int),
                                ("max_pitch_pert", C.c_double), ("max_roll_pert", C.c_double),
                                ("sample_from_distribution", C.c_int)]


_lib = None


class PreprocessParams(C.Structure):  # artp_preprocess_params (include/artp_c.h)
    _fields_ = [("traversability_thres", C.c_float), ("foothold_margin", C.c_double),
                ("foothold_margin_max_hole_size", C.c_double), ("foothold_margin_max_drop", C.c_double),
                ("foothold_margin_max_drop_search_radius", C.c_double), ("foothold_margin_min_step", C.c_double),
                ("foothold_size", C.c_double), ("use_inverse_vertex_density", C.c_int),
                ("use_max_prob_unknown_samples", C.c_int), ("max_prob_unknown_samples", C.c_double)]


class PreprocessInputs(C.Structure):  # artp_preprocess_inputs
    _fields_ = [("elevation", C.c_void_p), ("traversability", C.c_void_p), ("observed", C.c_void_p),
                ("vertex_se3", C.c_void_p), ("n_vertices", C.c_size_t), ("rows", C.c_int), ("cols", C.c_int),
                ("len_x", C.c_double), ("len_y", C.c_double), ("pos_x", C.c_double), ("pos_y", C.c_double)]


class GroupEdges(C.Structure):  # artp_group_edges
    _fields_ = [("valid", C.c_void_p), ("edge_i", C.c_void_p), ("edge_j", C.c_void_p), ("cost", C.c_void_p),
                ("n", C.c_size_t)]


class RoadmapParams(C.Structure):  # artp_roadmap_params (include/artp_c.h)
    _fields_ = [("seed", C.c_uint64), ("first_index", C.c_uint64), ("n_milestones", C.c_uint32),
                ("k_neighbors", C.c_uint32), ("objective", C.c_int32), ("max_replans", C.c_uint32),
                ("max_lon_vel", C.c_double), ("max_lat_vel", C.c_double), ("max_ang_vel", C.c_double),
                ("w_energy", C.c_float), ("w_time", C.c_float), ("w_risk", C.c_float),
                ("risk_threshold", C.c_float),
                ("max_n_edges", C.c_uint32), ("recompute_density_after_n_samples", C.c_uint32),
                ("max_sample_time", C.c_double), ("density_map", C.c_void_p), ("density_params", C.c_void_p),
                ("construction", C.c_int32), ("max_query_edge_length", C.c_double)]


VARIANTS_LIB_PATH = os.path.join(_HERE, "csrc", "libartp_variants.so")   # make -C art_planner_amd/csrc variants
_libs = {}


def load(path=None):
    """Load libartp.so (or another build of it: `path`, e.g. VARIANTS_LIB_PATH -- the forms that were measured and did not
    become the default, for the tests that pin their parity); raise ArtpError if it was not built (run `python -c 'import
    __graft_entry__ as g; g.build()'` or `make -C art_planner_amd/csrc`).  A context belongs to the library that made it."""
    global _lib
    if path is None and _lib is not None:
        return _lib
    if path is not None and os.path.abspath(path) != LIB_PATH:
        path = os.path.abspath(path)
        if path not in _libs:
            _libs[path] = _load_path(path)
        return _libs[path]
    _lib = _load_path(LIB_PATH)
    return _lib


def _load_path(LIB_PATH):
    if not os.path.exists(LIB_PATH):
        raise ArtpError(f"{LIB_PATH} is missing: build the HIP extension first "
                        "(make -C art_planner_amd/csrc). There is no CPU fallback.")
    try:
        # torch bundles its own HIP runtime: it must be the first one loaded in the process, otherwise
        # two copies of libamdhip64 end up side by side and neither sees the device properly.
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, sz, u64, dbl, i32 = C.c_void_p, C.c_size_t, C.c_uint64, C.c_double, C.c_int
    L.artp_params_defaults.argtypes = [C.POINTER(Params)]
    L.artp_params_yaml.argtypes = [C.POINTER(Params)]
    L.artp_status_string.argtypes = [i32]
    L.artp_status_string.restype = C.c_char_p
    L.artp_last_error.argtypes = [vp]
    L.artp_last_error.restype = C.c_char_p
    L.artp_device_arch.argtypes = [vp]
    L.artp_device_arch.restype = C.c_char_p
    L.artp_create.argtypes = [i32, C.POINTER(Params), C.POINTER(vp)]
    L.artp_destroy.argtypes = [vp]
    L.artp_destroy.restype = None
    L.artp_set_stream.argtypes = [vp, vp]
    L.artp_use_own_stream.argtypes = [vp]
    L.artp_synchronize.argtypes = [vp]
    L.artp_set_lane.argtypes = [vp, i32]
    L.artp_get_lane.argtypes = [vp]
    L.artp_upload_layer.argtypes = [vp, i32, vp, i32, i32, dbl, dbl, dbl, dbl]
    L.artp_update_layer_rect.argtypes = [vp, i32, vp, i32, i32, i32, i32]
    L.artp_update_layer_rects.argtypes = [vp, i32, i32, vp, vp]
    for name in ("artp_check_boxes", "artp_check_boxes_dev"):
        getattr(L, name).argtypes = [vp, i32, vp, vp, sz, vp, vp]
    for name in ("artp_validate_states", "artp_validate_states_dev"):
        getattr(L, name).argtypes = [vp, vp, sz, vp, vp]
    L.artp_upload_sampler_layers.argtypes = [vp] + [vp] * 7 + [i32, i32, dbl, dbl, dbl, dbl]
    for name in ("artp_sample_states", "artp_sample_states_dev"):
        getattr(L, name).argtypes = [vp, u64, u64, sz, vp]
    L.artp_sample_and_validate_dev.argtypes = [vp, u64, u64, sz, vp, vp, C.POINTER(sz)]
    L.artp_sample_and_validate.argtypes = [vp, u64, u64, sz, vp, vp, vp]
    L.artp_map_version.argtypes = [vp]
    L.artp_map_version.restype = C.c_uint64
    L.artp_set_z_bounds.argtypes = [vp, dbl, dbl]
    L.artp_set_few_edges.argtypes = [vp, i32]
    L.artp_set_persistent_latency.argtypes = [vp, i32]
    L.artp_persistent_latency_stats.argtypes = [vp, C.POINTER(u64 * 2)]
    L.artp_set_edge_passes.argtypes = [vp, i32, i32]
    L.artp_cost_set_fc_path.argtypes = [vp, i32]
    for name in ("artp_check_motions_last_valid", "artp_check_motions_last_valid_dev"):
        getattr(L, name).argtypes = [vp, vp, vp, sz, vp, vp, vp]
    for name in ("artp_check_motions", "artp_check_motions_dev"):
        getattr(L, name).argtypes = [vp, vp, vp, sz, vp]
    for name in ("artp_check_edges_interp", "artp_check_edges_interp_dev"):
        getattr(L, name).argtypes = [vp, vp, vp, sz, vp, vp]
    L.artp_compact_valid_dev.argtypes = [vp, vp, vp, sz, vp, vp]
    L.artp_compact_valid_indices_dev.argtypes = [vp, vp, sz, vp, vp]
    L.artp_sample_states_at_dev.argtypes = [vp, u64, u64, vp, vp, sz, vp]
    L.artp_materialise_from_bits_dev.argtypes = [vp, u64, vp, i32, sz, sz, vp, sz, vp, vp]
    L.artp_pack_edge_results_dev.argtypes = [vp, vp, vp, vp, vp, sz, vp, vp]
    L.artp_pack_valid_bits_dev.argtypes = [vp, vp, sz, vp]
    L.artp_indices_from_bits_dev.argtypes = [vp, vp, sz, vp, vp]
    L.artp_shard_first_index.argtypes = [u64, i32, i32, u64]
    L.artp_shard_first_index.restype = u64
    L.artp_group_create.argtypes = [C.POINTER(i32), i32, C.POINTER(Params), i32, C.POINTER(vp)]
    L.artp_group_unique_id.argtypes = [vp]
    L.artp_group_create_rank.argtypes = [i32, i32, i32, vp, C.POINTER(Params), C.POINTER(vp)]
    L.artp_group_destroy.argtypes = [vp]
    L.artp_group_destroy.restype = None
    L.artp_group_last_error.argtypes = [vp]
    L.artp_group_last_error.restype = C.c_char_p
    L.artp_group_world_size.argtypes = [vp]
    L.artp_group_local_count.argtypes = [vp]
    L.artp_group_rank.argtypes = [vp, i32]
    L.artp_group_ctx.argtypes = [vp, i32]
    L.artp_group_ctx.restype = vp
    L.artp_group_ranks_seen.argtypes = [vp, C.POINTER(i32)]
    L.artp_group_configure.argtypes = [vp, u64, sz, sz, sz]
    L.artp_group_sample_and_validate_step.argtypes = [vp, u64]
    L.artp_group_step_buffers.argtypes = [vp, i32, u64] + [C.POINTER(vp)] * 5
    L.artp_group_exchange_edges.argtypes = [vp, C.POINTER(GroupEdges), sz]
    L.artp_group_edge_buffers.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(vp)]
    L.artp_group_synchronize.argtypes = [vp, i32]
    L.artp_group_abort.argtypes = [vp]
    L.artp_algorithmic_vertices_dev.argtypes = [vp, vp, sz, C.POINTER(u64)]
    L.artp_debug_pipeline_counters.argtypes = [vp, C.POINTER(u64 * 8)]
    L.artp_debug_partner_table.argtypes = [vp, C.c_int, vp, C.c_size_t, C.POINTER(C.c_int)]
    L.artp_roadmap_params_defaults.argtypes = [C.POINTER(RoadmapParams)]
    L.artp_roadmap_params_defaults.restype = None
    L.artp_roadmap_build.argtypes = [vp, C.POINTER(RoadmapParams), vp, vp, C.POINTER(vp)]
    L.artp_roadmap_stats.argtypes = [vp, C.POINTER(u64 * 8)]
    L.artp_roadmap_export.argtypes = [vp] + [vp] * 8
    L.artp_roadmap_solve.argtypes = [vp, vp, sz, C.POINTER(sz), C.POINTER(dbl), C.POINTER(i32)]
    L.artp_roadmap_revalidate.argtypes = [vp, C.POINTER(u64 * 4)]
    L.artp_roadmap_set_query.argtypes = [vp, vp, vp]
    L.artp_roadmap_simplify_path.argtypes = [vp, vp, sz, vp, C.POINTER(sz), C.POINTER(dbl)]
    L.artp_roadmap_grow.argtypes = [vp, C.c_uint64, vp]
    L.artp_roadmap_solve_until.argtypes = [vp, dbl, C.c_uint32, vp, sz, C.POINTER(sz), C.POINTER(dbl), vp]
    L.artp_roadmap_set_density_map.argtypes = [vp, vp, C.POINTER(PreprocessParams)]
    L.artp_preprocessed_reweight_dev.argtypes = [vp, vp, C.POINTER(PreprocessParams), vp, sz, i32]
    L.artp_roadmap_destroy.argtypes = [vp]
    L.artp_roadmap_destroy.restype = None
    for name in ("artp_preprocess_params_defaults", "artp_preprocess_params_yaml"):
        getattr(L, name).argtypes = [C.POINTER(PreprocessParams)]
        getattr(L, name).restype = None
    L.artp_preprocess_map.argtypes = [vp, vp, vp, i32, i32, dbl, dbl, dbl, dbl, C.POINTER(PreprocessParams),
                                      C.POINTER(vp)]
    L.artp_preprocess_map_ex.argtypes = [vp, C.POINTER(PreprocessInputs), C.POINTER(PreprocessParams), C.POINTER(vp)]
    L.artp_preprocessed_change.argtypes = [vp, vp, vp, C.c_float, vp, C.POINTER(i32 * 4), C.POINTER(u64)]
    L.artp_preprocessed_get_layer.argtypes = [vp, vp, C.c_char_p, vp]
    L.artp_preprocessed_install.argtypes = [vp, vp]
    L.artp_preprocessed_destroy.argtypes = [vp]
    L.artp_preprocessed_destroy.restype = None
    L.artp_inpaint_layer.argtypes = [vp, vp, i32, i32, i32, vp, C.POINTER(u64)]
    L.artp_cost_set_hole_filling.argtypes = [vp, i32]
    L.artp_cost_set_external_query.argtypes = [vp, COST_QUERY_FN, vp]
    L.artp_cost_blob_bytes.argtypes = []
    L.artp_cost_blob_bytes.restype = sz
    L.artp_cost_load_weights.argtypes = [vp, vp, sz]
    L.artp_cost_update_map.argtypes = [vp, vp, i32, i32, dbl, dbl, dbl, dbl, dbl]
    L.artp_cost_update_map_dev.argtypes = [vp, vp, i32, i32, dbl, dbl, dbl, dbl, dbl]
    L.artp_cost_update_map_layer.argtypes = [vp, vp, i32, i32, dbl, dbl, dbl, dbl, dbl]
    L.artp_cost_query.argtypes = [vp, vp, sz, vp]
    L.artp_cost_query_dev.argtypes = [vp, vp, sz, vp]
    L.artp_cost_debug_query_cells.argtypes = [vp, vp, sz, vp, vp]
    L.artp_cost_get_features.argtypes = [vp, vp, C.POINTER(i32), C.POINTER(i32)]
    L.artp_set_r3_extent.argtypes = [vp, dbl]
    L.artp_telea_inpaint_u8.argtypes = [vp, vp, i32, i32, i32, vp]
    L.artp_cost_fc_path.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(C.c_float)]
    for name in SYMBOLS:
        fn = getattr(L, name)
        if fn.restype is C.c_int and name not in ("artp_destroy", "artp_cost_blob_bytes", "artp_roadmap_destroy",
                                                  "artp_roadmap_params_defaults", "artp_preprocess_params_defaults",
                                                  "artp_preprocess_params_yaml", "artp_preprocessed_destroy",
                                                  "artp_group_destroy"):
            fn.restype = C.c_int
    return L


def check(ctx, rc: int, what: str, L=None) -> None:
    if rc != 0:
        L = L or load()
        msg = L.artp_status_string(rc).decode()
        detail = L.artp_last_error(ctx).decode() if ctx else ""
        err = ArtpError(f"{what} failed: {msg} ({rc}) {detail}")
        err.status = rc
        raise err
