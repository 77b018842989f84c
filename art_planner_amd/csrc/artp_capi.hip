// artp_capi.hip -- C ABI (include/artp_c.h) over the gfx950 kernels.  No CPU compute path exists
// here: without a HIP device artp_create fails with ARTP_ERR_NO_DEVICE.
#include "../../include/artp_c.h"

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cmath>
#include <cstdio>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <array>
#include <vector>

#include "pipeline.h"
#include "cost_kernels.h"

using namespace artp;

#define ARTP_FEW_STATES 16
#define ARTP_SVC_MAX_STATES 2   // artp_set_persistent_latency serves calls of up to this many states
// mapped block of the edge latency path: offsets of its parts (see artp_ctx::pin_edges)
#define FEW_EDGE_S1 0
#define FEW_EDGE_S2 (ARTP_FEW_EDGES * 7 * sizeof(double))
#define FEW_EDGE_LAST_T (2 * ARTP_FEW_EDGES * 7 * sizeof(double))
#define FEW_EDGE_LAST_STATE (FEW_EDGE_LAST_T + ARTP_FEW_EDGES * sizeof(double))
#define FEW_EDGE_AUX (FEW_EDGE_LAST_STATE + ARTP_FEW_EDGES * 7 * sizeof(double))
#define FEW_EDGE_STATUS (FEW_EDGE_AUX + ARTP_FEW_EDGES * sizeof(uint32_t))
#define FEW_EDGE_BLOCK_BYTES (FEW_EDGE_STATUS + ARTP_FEW_EDGES)
#define ARTP_MAX_LANES 4

struct artp_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  artp_params params{};
  RobotDev robot{};
  MapGeom geom{};
  bool have_geom = false;
  // artp_map_version(): bumped (under `mu`) by everything that changes a layer, its tables or the sampler tables
  std::atomic<uint64_t> map_version{0};
  FieldDev field[2]{};
  float* field_data[2] = {nullptr, nullptr};
  size_t field_elems[2] = {0, 0};
  bool have_field[2] = {false, false};
  void* rect_stage_host = nullptr;   // artp_update_layer_rects: pinned staging (rectangle records + patches) ...
  void* rect_stage_dev = nullptr;    // ... and its device twin
  size_t rect_stage_cap = 0;
  hipEvent_t rect_stage_done = nullptr;  // the last update's host-to-device copy
  // Ordering of map writes against the lanes: a map write is issued on the current lane's stream.  In
  // front of it that stream waits for the work every other lane has been given so far (they may still read the old
  // samples / tables: lane_mark); behind it `map_written` is recorded, and a lane whose map_seen is behind
  // map_write_seq waits for that event before its next launch (artp_set_lane / the write itself for the current lane).
  hipEvent_t map_written = nullptr;
  hipEvent_t lane_mark[4] = {nullptr, nullptr, nullptr, nullptr};
  uint64_t map_write_seq = 0;
  uint64_t map_seen[4] = {0, 0, 0, 0};
  SamplerDev sampler{};
  float* sampler_buf = nullptr;
  float* sampler_pack = nullptr;  // derived tables: packed cells, row-major CDF, pivots
  int sampler_rows = 0, sampler_cols = 0;  // grid size sampler_buf / sampler_pack were allocated for
  bool have_sampler = false;
  double z_low = 0.0, z_high = 0.0;
  bool have_z = false;
  int fc_mfma_wanted = 1;     // artp_cost_set_fc_path: 1 = FCpart on the matrix cores (self-checked at load), 0 = the fp32 VALU kernels
  artp_cost_query_fn ext_cost_fn = nullptr;   // artp_cost_set_external_query: the roadmap's learned-cost batches go here
  void* ext_cost_user = nullptr;
  bool few_edges = true;      // <= ARTP_FEW_EDGES edges per HOST call: the one-launch latency kernel (artp_set_few_edges)
  bool edge_two_pass = true;  // artp_check_motions: coarse pass first (artp_set_edge_passes)
  int edge_coarse_stride = ARTP_COARSE_STRIDE;  // $ARTP_COARSE_STRIDE (tuning)
  // map tables (pipeline.h): per layer 6 levels of {max, min} + 6 levels of non-finite / NaN flag bytes, the
  // partner table and the raw cross products it is built from
  float* table_buf[2] = {nullptr, nullptr};
  unsigned char* flag_buf[2] = {nullptr, nullptr};  // per-level non-finite / NaN block flags
  unsigned* stride_buf[2] = {nullptr, nullptr};      // stride tables (TablesDev::st)
  unsigned char* partner_buf[2] = {nullptr, nullptr};
  unsigned* partner_cnt[2] = {nullptr, nullptr};  // partner counts per cell (pipeline.h partner_count_kernel)
  int* d_diff = nullptr;                          // result of layer_diff_kernel (six ints)
  int partner_R_built[2] = {-1, -1};
  int layer_has_nonfinite[2] = {1, 1};
  float4* tri_raw_buf[2] = {nullptr, nullptr};
  size_t table_elems[2] = {0, 0};
  TablesDev tables[2]{};
  ScratchCaps caps_full{0, 0, 0, 0};  // window tile + triangle list + hash table (1 wave / block)
  ScratchCaps caps_scan{0, 0, 0, 0};  // torso resolve stage: window tile + short triangle list, per wave
  ScratchCaps caps_feet{0, 0, 0, 0};  // foot resolve stage: per 16-lane group
  ScratchCaps caps_feet_wave{0, 0, 0, 0};  // foot list pass: a whole wavefront per box
  ScratchCaps caps_foot_full{0, 0, 0, 0};  // validate_few_kernel: a foot wavefront's full zone-test scratch
  int n_cus = 256;
  // latency path (<= ARTP_FEW_STATES states per call): mapped pinned host memory, read / written by the kernel
  double* pin_states = nullptr;           // host view, ARTP_FEW_STATES x 7
  volatile uint8_t* pin_labels = nullptr; // host view
  double* pin_states_dev = nullptr;       // device view of the same memory
  uint8_t* pin_labels_dev = nullptr;
  bool poll_labels = true;                // spin on the mapped labels instead of hipStreamSynchronize
  // persistent latency service (artp_set_persistent_latency): its own stream, a mailbox in mapped host memory
  bool svc_enabled = false, svc_launched = false;
  hipStream_t svc_stream = nullptr;
  hipEvent_t svc_after_map = nullptr;     // orders a (re)launch behind the map writes already on the context's stream
  SvcMailbox* svc = nullptr;              // host view
  SvcMailbox* svc_dev = nullptr;          // device view
  uint32_t svc_seq = 0;
  uint64_t svc_map_version = 0;
  uint64_t svc_launches = 0, svc_requests = 0;
  // the same switch keeps a POOL of workgroups resident for edge calls of one or two edges (check_motions_pool_kernel)
  bool pool_launched = false;
  hipStream_t pool_stream = nullptr;
  hipEvent_t pool_after_map = nullptr;
  EdgeMailbox* pool_mb = nullptr;         // the request block: DEVICE memory that the host writes through the PCIe BAR
  int pool_bar = -1;                      // does the device expose its memory to the host (large BAR)?  -1 = not asked yet
  PoolResponse* pool_resp = nullptr;      // host view (per-workgroup slots)
  PoolResponse* pool_resp_dev = nullptr;
  PoolCtl* pool_ctl = nullptr;            // device memory
  uint64_t pool_shadow[5][7] = {};        // the payload words of the request block as last written
  unsigned pool_wgs = ARTP_POOL_WGS;
  uint32_t pool_seq = 0;
  uint64_t pool_map_version = 0;
  uint64_t pool_launches = 0, pool_requests = 0;
  // latency path of the edge checks (<= ARTP_FEW_EDGES edges per call): one mapped block
  //   s1 | s2 (64 x 7 f64 each) | last_t (64 f64) | last_state (64 x 7 f64) | aux (64 u32) | status (64 u8)
  char* pin_edges_in = nullptr;           // s1 | s2 of a call, host view
  char* pin_edges_in_dev = nullptr;
  char* pin_edges = nullptr;              // host view
  char* pin_edges_dev = nullptr;          // device view
  FewEdgeSync* d_few_sync = nullptr;      // device, armed {~0, 0, 0} per edge (the kernel re-arms what it used)
  // device scratch
  int* d_error = nullptr;
  unsigned long long* d_count = nullptr;
  void* tmp[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  size_t tmp_cap[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  void* cub_tmp = nullptr;
  size_t cub_cap = 0;
  // Lanes (artp_set_lane): everything above that a call in flight owns -- stream, scratch buffers, counters -- exists
  // once per lane, so calls issued on different lanes (from one host thread, on different streams) overlap on the
  // GPU.  The members above are the CURRENT lane's; the others are parked here.  The map, its tables and the
  // sampler are shared and read-only while states are validated.
  struct lane_state {
    bool init = false;
    hipStream_t own_stream = nullptr, stream = nullptr;
    unsigned long long* d_count = nullptr;
    int* d_error = nullptr;  // per lane: a capacity overflow is reported to the caller of the batch that raised it
    void* tmp[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t tmp_cap[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    void* cub_tmp = nullptr;
    size_t cub_cap = 0;
  };
  lane_state lanes[ARTP_MAX_LANES];
  int cur_lane = 0;
  // motion cost (R8/R9)
  bool have_weights = false;
  half8* d_convw[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // [4]: the 15 x 15 layer's B fragments, per-row packing
  float* d_convb[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  float* d_c12 = nullptr;              // conv1 o conv2 composed: [24][25] + [24] (conv12_pool_kernel)
  unsigned char* d_c12m = nullptr;     // the same as MFMA fragments, hi / lo half floats + bias[32] (conv12_mfma_kernel)
  int conv12_mfma = 1;                 // $ARTP_CONV12_MFMA=0: the VALU form (rounds 3-4)
  half8* d_convw_chunk[3] = {nullptr, nullptr, nullptr};  // conv3..5 B fragments in chunk order (conv345_kernel)
  half8* d_convw_p32 = nullptr;   // variants build: the 15 x 15 layer's weights for conv15_pair32_kernel
  float* d_fc = nullptr;               // FcWeights::TOTAL floats
  char* d_fc_mfma = nullptr;           // FcMfma::TOTAL bytes: the same MLP in MFMA fragment order (fc_mfma_pack)
  int feet_dense = 0;                  // $ARTP_FEET_DENSE=1: feet_stream2_kernel (corner arithmetic on dense lanes; measured: no faster)
  double r3_extent_override = 0.0;     // artp_set_r3_extent: > 0 = checkMotion's R^3 maxExtent, whatever the installed map's bounds
  int fc_mfma = 1;                     // the kernel artp_cost_query uses now (artp_cost_fc_path)
  int fc_selfcheck = -1;               // artp_cost_load_weights' probe batch: 1 = the MFMA kernel agreed with the fp32 one,
                                       // 0 = it did not (fc_mfma forced to 0), -1 = not run
  float fc_selfcheck_err = 0.f;        // largest |MFMA - fp32| of the probe batch
  half_t* d_act[2] = {nullptr, nullptr};
  size_t act_cap = 0;
  half_t* d_feat = nullptr;            // NHWC [Fh][Fw][48]
  size_t feat_cap = 0;
  float* d_map_f32 = nullptr;
  size_t map_cap = 0;
  CostMapGeom cost_geom{};
  int feat_h = 0, feat_w = 0;
  bool have_features = false;
  bool cost_fill_holes = false;        // artp_cost_set_hole_filling
  bool cost_fill_telea = false;   // artp_cost_set_hole_filling(ctx, 2): Telea's fast-marching fill (telea.h)
  std::string last_error;
  std::string arch;
  std::recursive_mutex mu;  // recursive: host entry points hold it across the _dev calls they are built from
};

namespace {

#define HIP_TRY(ctx, expr)                                                        \
  do {                                                                            \
    hipError_t _e = (expr);                                                       \
    if (_e != hipSuccess) {                                                       \
      (ctx)->last_error = std::string(#expr) + ": " + hipGetErrorString(_e);      \
      return ARTP_ERR_HIP;                                                        \
    }                                                                             \
  } while (0)

int ensure_tmp(artp_ctx* c, int slot, size_t bytes) {
  if (c->tmp_cap[slot] >= bytes) return ARTP_OK;
  if (c->tmp[slot]) HIP_TRY(c, hipFree(c->tmp[slot]));
  c->tmp[slot] = nullptr;
  c->tmp_cap[slot] = 0;
  const size_t want = bytes + bytes / 4 + 256;
  HIP_TRY(c, hipMalloc(&c->tmp[slot], want));
  c->tmp_cap[slot] = want;
  return ARTP_OK;
}

void park_lane(artp_ctx* c) {  // current members -> lanes[cur_lane]
  artp_ctx::lane_state& l = c->lanes[c->cur_lane];
  l.init = true;
  l.own_stream = c->own_stream;
  l.stream = c->stream;
  l.d_count = c->d_count;
  l.d_error = c->d_error;
  l.cub_tmp = c->cub_tmp;
  l.cub_cap = c->cub_cap;
  for (int k = 0; k < 8; ++k) {
    l.tmp[k] = c->tmp[k];
    l.tmp_cap[k] = c->tmp_cap[k];
  }
}

void unpark_lane(artp_ctx* c, int lane) {
  const artp_ctx::lane_state& l = c->lanes[lane];
  c->own_stream = l.own_stream;
  c->stream = l.stream;
  c->d_count = l.d_count;
  c->d_error = l.d_error;
  c->cub_tmp = l.cub_tmp;
  c->cub_cap = l.cub_cap;
  for (int k = 0; k < 8; ++k) {
    c->tmp[k] = l.tmp[k];
    c->tmp_cap[k] = l.tmp_cap[k];
  }
  c->cur_lane = lane;
}

// In front of a map write: the current stream waits for everything the other lanes were given so far.
int order_after_other_lanes(artp_ctx* c) {
  for (int l = 0; l < ARTP_MAX_LANES; ++l) {
    if (l == c->cur_lane || !c->lanes[l].init) continue;
    if (!c->lane_mark[l]) HIP_TRY(c, hipEventCreateWithFlags(&c->lane_mark[l], hipEventDisableTiming));
    HIP_TRY(c, hipEventRecord(c->lane_mark[l], c->lanes[l].stream));
    HIP_TRY(c, hipStreamWaitEvent(c->stream, c->lane_mark[l], 0));
  }
  return ARTP_OK;
}

// Behind a map write that returns before the device is done (rectangle updates, same-geometry re-install): the other
// lanes' next launches wait for it.  The host-synchronous writers (full upload, sampler layers) call it too: free.
int publish_map_write(artp_ctx* c) {
  if (!c->map_written) HIP_TRY(c, hipEventCreateWithFlags(&c->map_written, hipEventDisableTiming));
  HIP_TRY(c, hipEventRecord(c->map_written, c->stream));
  c->map_seen[c->cur_lane] = ++c->map_write_seq;
  return ARTP_OK;
}

int lane_sees_map(artp_ctx* c) {  // after unpark_lane: the now-current lane catches up with the last map write
  if (c->map_seen[c->cur_lane] != c->map_write_seq && c->map_written) {
    HIP_TRY(c, hipStreamWaitEvent(c->stream, c->map_written, 0));
    c->map_seen[c->cur_lane] = c->map_write_seq;
  }
  return ARTP_OK;
}

void fill_robot(artp_ctx* c) {
  const artp_params& p = c->params;
  RobotDev& r = c->robot;
  // HeightMapBoxChecker(float, float, float) ctor args (validity_checker_body.cpp:10-14,
  // validity_checker_feet.cpp:13-18): doubles narrowed to float at the call.
  r.torso[0] = (float)p.torso_length;
  r.torso[1] = (float)p.torso_width;
  r.torso[2] = (float)p.torso_height;
  r.foot[0] = (float)p.reach_x;
  r.foot[1] = (float)p.reach_y;
  r.foot[2] = (float)p.reach_z;
  // Pose3FromXYZ(Scalar, Scalar, Scalar) args (validity_checker.cpp:41-43)
  r.torso_off[0] = (float)p.torso_off_x;
  r.torso_off[1] = (float)p.torso_off_y;
  r.torso_off[2] = (float)(p.torso_off_z - p.feet_off_z);
  r.feet_off_x = (float)p.feet_off_x;
  r.feet_off_y = (float)p.feet_off_y;
  r.unknown_space_untraversable = p.unknown_space_untraversable;
  r.reach_z = p.reach_z;
  r.max_pitch_pert = p.max_pitch_pert;
  r.max_roll_pert = p.max_roll_pert;
}

// dRFrom2Axes (ode/ode/src/rotation.cpp:94-130) for the field rotation.
void r_from_2_axes(float* R, float ax, float ay, float az, float bx, float by, float bz) {
  float l = sqrtf(ax * ax + ay * ay + az * az);
  if (l <= 0.0f) return;
  l = 1.0f / l;
  ax *= l; ay *= l; az *= l;
  const float k = ax * bx + ay * by + az * bz;
  bx -= k * ax; by -= k * ay; bz -= k * az;
  l = sqrtf(bx * bx + by * by + bz * bz);
  if (l <= 0.0f) return;
  l = 1.0f / l;
  bx *= l; by *= l; bz *= l;
  R[0] = ax; R[4] = ay; R[8] = az;
  R[1] = bx; R[5] = by; R[9] = bz;
  R[2] = -by * az + ay * bz;
  R[6] = -bz * ax + az * bx;
  R[10] = -bx * ay + ax * by;
  R[3] = R[7] = R[11] = 0.0f;
}

// LDS window tile sized from the largest box diagonal and the sample spacing (any orientation).
int size_scratch(artp_ctx* c) {
  const artp_params& p = c->params;
  const double d_torso = std::sqrt(p.torso_length * p.torso_length + p.torso_width * p.torso_width +
                                   p.torso_height * p.torso_height);
  const double d_foot = std::sqrt(p.reach_x * p.reach_x + p.reach_y * p.reach_y + p.reach_z * p.reach_z);
  double diag = d_torso > d_foot ? d_torso : d_foot;
  double spacing = 1e30;
  int max_n = 2;
  for (int s = 0; s < 2; ++s) {
    if (!c->have_field[s]) continue;
    spacing = std::fmin(spacing, (double)std::fmin(c->field[s].sample_w, c->field[s].sample_d));
    max_n = std::max(max_n, std::max(c->field[s].nW, c->field[s].nD));
  }
  if (spacing > 1e29) return ARTP_OK;
  int maxdim = (int)std::ceil(diag / spacing) + 4;
  if (maxdim > max_n) maxdim = max_n;
  if (maxdim < 4) maxdim = 4;
  long verts = (long)maxdim * maxdim;
  long tris = 2L * (maxdim - 1) * (maxdim - 1);
  if (tris > 4096) tris = 4096;  // 64 lanes x 64-bit assignment mask
  verts = (verts + 3) & ~3L;
  tris = (tris + 7) & ~7L;
  long tab = 64;
  while (tab < 2 * tris && tab < 4096) tab <<= 1;  // partner-detection fast path up to tab/2 triangles
  if (maxdim > 64) {
    c->last_error = "box window wider than 64 samples: not supported by the packed triangle ids";
    return ARTP_ERR_CAPACITY;
  }
  c->caps_full = ScratchCaps{64 * 36, (int)verts, (int)tris, (int)tab};
  // resolve stage: window tile + a short triangle list (longer lists take the exact-grouping stage)
  c->caps_scan = ScratchCaps{64 * 36, (int)verts, (int)(tris < 1024 ? tris : 1024), 0};
  {
    int fdim = (int)std::ceil(d_foot / spacing) + 4;
    if (fdim > max_n) fdim = max_n;
    if (fdim < 4) fdim = 4;
    long fv = ((long)fdim * fdim + 3) & ~3L;
    long ft = (2L * (fdim - 1) * (fdim - 1) + 7) & ~7L;
    if (ft > 1024) ft = 1024;
    c->caps_feet = ScratchCaps{32 * 36, (int)fv, (int)ft, 0};
    c->caps_feet_wave = ScratchCaps{64 * 36, (int)fv, (int)ft, 0};
    // a whole wavefront per foot box (validate_few_kernel): full list + hash table like caps_full
    long ftf = (2L * (fdim - 1) * (fdim - 1) + 7) & ~7L;
    if (ftf > 4096) ftf = 4096;
    long ftab = 64;
    while (ftab < 2 * ftf && ftab < 4096) ftab <<= 1;
    c->caps_foot_full = ScratchCaps{64 * 36, (int)fv, (int)ftf, (int)ftab};
  }
  if (scratch_bytes_per_wave(c->caps_full) > 160 * 1024 ||
      scratch_bytes_per_wave(c->caps_scan) * ARTP_WAVES_PER_BLOCK > 160 * 1024 ||
      scratch_bytes_per_wave(c->caps_full) + 4 * scratch_bytes_per_wave(c->caps_foot_full) > 160 * 1024) {
    c->last_error = "box too large for the LDS window tile";
    return ARTP_ERR_CAPACITY;
  }
  return ARTP_OK;
}

size_t lds_full(const artp_ctx* c) { return scratch_bytes_per_wave(c->caps_full); }
size_t lds_scan(const artp_ctx* c) { return scratch_bytes_per_wave(c->caps_scan) * ARTP_WAVES_PER_BLOCK; }
size_t lds_feet(const artp_ctx* c) { return scratch_bytes_per_wave(c->caps_feet) * 4 * ARTP_WAVES_PER_BLOCK; }
size_t lds_feet_wave(const artp_ctx* c) { return scratch_bytes_per_wave(c->caps_feet_wave) * ARTP_WAVES_PER_BLOCK; }
size_t lds_few(const artp_ctx* c) {
  return scratch_bytes_per_wave(c->caps_full) + 4 * scratch_bytes_per_wave(c->caps_foot_full);
}

int set_kernel_lds(artp_ctx* c) {
  // > 64 KiB of dynamic LDS needs the opt-in attribute
  HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(check_boxes_kernel<1>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_full(c)));
  HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(validate_states_kernel<1>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_full(c)));
  HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(plane_stage_kernel<1>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_full(c)));
  HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(validate_few_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_few(c)));
  HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(check_motions_few_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_few(c)));
  HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(resolve_boxes_kernel<ARTP_WAVES_PER_BLOCK, 64, 3>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_scan(c)));
  HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(resolve_boxes_kernel<ARTP_WAVES_PER_BLOCK, 16, 1>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_feet(c)));
  HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(resolve_boxes_kernel<ARTP_WAVES_PER_BLOCK, 64, 2>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_feet_wave(c)));
  return ARTP_OK;
}

// persistent grids: as many blocks as fit the LDS on every CU (the rest is grid-stride / work queue)
int grid_full(const artp_ctx* c, size_t tasks) {
  size_t per_cu = (160 * 1024) / (lds_full(c) ? lds_full(c) : 1);
  if (per_cu > 16) per_cu = 16;
  if (per_cu < 1) per_cu = 1;
  size_t g = (size_t)c->n_cus * per_cu;
  if (tasks && tasks < g) g = tasks;
  return (int)(g ? g : 1);
}
int grid_scan(const artp_ctx* c, size_t lds) {
  size_t per_cu = (160 * 1024) / (lds ? lds : 1);
  if (per_cu > 8) per_cu = 8;
  if (per_cu < 1) per_cu = 1;
  return (int)((size_t)c->n_cus * per_cu);
}

// persistent grid of the sub-queue consumers: per_cu workgroups per CU, rounded up to a multiple of ARTP_NSUB
unsigned grid_sub(const artp_ctx* c, int per_cu) {
  const unsigned g = (unsigned)c->n_cus * (unsigned)per_cu;
  return (g + ARTP_NSUB - 1) / ARTP_NSUB * ARTP_NSUB;
}

// Partner table of a layer (FieldDev::partner_flags).  Radius = the largest index window a box of that layer can have:
// box diagonal / sample spacing, + the window's rounding and dCollideHeightfield's +1 border; -1 = no table (windows
// larger than the stages hold anyway).
int partner_radius(const artp_ctx* c, int slot) {
  const FieldDev& f = c->field[slot];
  const float* side = slot == 1 ? c->robot.foot : c->robot.torso;
  const double diag = std::sqrt((double)side[0] * side[0] + (double)side[1] * side[1] + (double)side[2] * side[2]);
  const int R = (int)std::ceil(diag / std::fmin((double)f.sample_w, (double)f.sample_d)) + 3;
  return R > 63 ? -1 : R;
}

// launch geometry over a rectangle set: the largest rectangle (widened by `margin`) decides grid.x, the tallest grid.y
static void partner_rect_extent(const FieldDev& f, const PartnerRects& pr, int margin, int* max_cells, int* max_rows) {
  *max_cells = *max_rows = 1;
  for (int k = 0; k < pr.n; ++k) {
    const int nx = std::min(pr.x1[k] + margin, f.nW - 1) - std::max(pr.x0[k] - margin, 0) + 1;
    const int nz = std::min(pr.z1[k] + margin, f.nD - 1) - std::max(pr.z0[k] - margin, 0) + 1;
    *max_cells = std::max(*max_cells, nx * nz);
    *max_rows = std::max(*max_rows, pr.z1[k] - pr.z0[k] + 1);
  }
}

// Whole layer: raw cross products, counts over every full neighbourhood, flags.
int build_partner_table_full(artp_ctx* c, int slot) {
  FieldDev& f = c->field[slot];
  f.partner_flags = nullptr;
  f.partner_R = 0;
  c->partner_R_built[slot] = -1;
  const int R = partner_radius(c, slot);
  if (R < 0) return ARTP_OK;
  PartnerRects all{};
  all.n = 1;
  all.x0[0] = all.z0[0] = 0;
  all.x1[0] = f.nW - 1;   // the last column / row hold no cell: tri_raw writes +inf there, the counts stay 0
  all.z1[0] = f.nD - 1;
  const int ncell = f.nW * f.nD;
  hipLaunchKernelGGL(tri_raw_rects_kernel, dim3((ncell + 255) / 256, 1), dim3(256), 0, c->stream, f, all,
                     c->tri_raw_buf[slot], c->partner_cnt[slot]);
  hipLaunchKernelGGL(partner_count_kernel<false>, dim3((ncell + 255) / 256, 2 * R + 1, 1), dim3(256), 0, c->stream, f, R,
                     (const float4*)c->tri_raw_buf[slot], c->partner_cnt[slot], all, 1);
  hipLaunchKernelGGL(partner_flags_from_counts_kernel, dim3((ncell + 255) / 256, 1), dim3(256), 0, c->stream, f.nW, f.nD, all,
                     0, (const unsigned*)c->partner_cnt[slot], c->partner_buf[slot]);
  HIP_TRY(c, hipGetLastError());
  c->partner_R_built[slot] = R;
  f.partner_flags = c->partner_buf[slot];
  f.partner_R = R;
  return ARTP_OK;
}

// Rectangle updates, step 1 of 2 -- BEFORE the samples are overwritten.  dirty: n x {x0, z0, x1, z1} inclusive sample
// ranges.  The cells whose triangles change are the rectangles widened by one cell towards the origin; rectangles that
// touch are merged into their bounding box (a cell treated as changed although it is not costs work, not
// correctness: its old and new contributions cancel).  Returns the rectangle set in *pr (n = 0: the caller rebuilds
// the whole table in step 2) after taking the OLD triangles' contributions off the counts of the cells around them.
int partner_update_begin(artp_ctx* c, int slot, const int* dirty, int n_dirty, PartnerRects* pr) {
  const FieldDev& f = c->field[slot];
  *pr = PartnerRects{};
  const int R = partner_radius(c, slot);
  if (R < 0 || c->partner_R_built[slot] != R || n_dirty <= 0) return ARTP_OK;
  std::vector<std::array<int, 4>> rects;
  for (int k = 0; k < n_dirty; ++k)
    rects.push_back({std::max(dirty[4 * k] - 1, 0), std::max(dirty[4 * k + 1] - 1, 0), std::min(dirty[4 * k + 2], f.nW - 2),
                     std::min(dirty[4 * k + 3], f.nD - 2)});
  for (bool merged = true; merged;) {
    merged = false;
    for (size_t a = 0; a < rects.size() && !merged; ++a)
      for (size_t b = a + 1; b < rects.size() && !merged; ++b)
        if (rects[a][0] <= rects[b][2] && rects[b][0] <= rects[a][2] && rects[a][1] <= rects[b][3] && rects[b][1] <= rects[a][3]) {
          rects[a] = {std::min(rects[a][0], rects[b][0]), std::min(rects[a][1], rects[b][1]),
                      std::max(rects[a][2], rects[b][2]), std::max(rects[a][3], rects[b][3])};
          rects.erase(rects.begin() + b);
          merged = true;
        }
  }
  if (rects.size() > 8) return ARTP_OK;
  // a changed region that is most of the layer: the full build is cheaper than two restricted passes
  size_t changed = 0;
  for (const auto& r : rects) changed += (size_t)(r[2] - r[0] + 1) * (r[3] - r[1] + 1);
  if (changed * 4 > (size_t)f.nW * f.nD) return ARTP_OK;
  for (size_t k = 0; k < rects.size(); ++k) {
    if (rects[k][2] < rects[k][0] || rects[k][3] < rects[k][1]) continue;
    const int j = pr->n++;
    pr->x0[j] = rects[k][0]; pr->z0[j] = rects[k][1]; pr->x1[j] = rects[k][2]; pr->z1[j] = rects[k][3];
  }
  if (pr->n == 0) return ARTP_OK;
  int cells, rows;
  partner_rect_extent(f, *pr, R, &cells, &rows);
  hipLaunchKernelGGL(partner_count_kernel<true>, dim3((cells + 255) / 256, rows, pr->n), dim3(256), 0, c->stream, f, R,
                     (const float4*)c->tri_raw_buf[slot], c->partner_cnt[slot], *pr, -1);
  HIP_TRY(c, hipGetLastError());
  // From here to the end of step 2 the counts are half-updated: until partner_update_end has queued its passes the
  // table counts as NOT built, so a failure in between (the copy, the scatter, the range tables) makes the next update
  // rebuild it from scratch instead of working incrementally on broken counts.
  c->partner_R_built[slot] = -1;
  c->tables[slot].valid = 0;
  c->field[slot].partner_flags = nullptr;  // a validation that ran on a half-updated map would take the list path
  c->field[slot].partner_R = 0;
  return ARTP_OK;
}

// Step 2 of 2 -- after the samples changed: the new raw cross products of the changed cells, the NEW triangles'
// contributions onto the cells around them, a full recount of the changed cells themselves, flags of everything touched.
int partner_update_end(artp_ctx* c, int slot, const PartnerRects& pr) {
  if (pr.n == 0) return build_partner_table_full(c, slot);
  FieldDev& f = c->field[slot];
  const int R = partner_radius(c, slot);  // what step 1 checked the table against (it then marked it "not built")
  int cells0, cellsR, rows;
  partner_rect_extent(f, pr, 0, &cells0, &rows);
  partner_rect_extent(f, pr, R, &cellsR, &rows);
  hipLaunchKernelGGL(tri_raw_rects_kernel, dim3((cells0 + 255) / 256, pr.n), dim3(256), 0, c->stream, f, pr,
                     c->tri_raw_buf[slot], c->partner_cnt[slot]);
  hipLaunchKernelGGL(partner_count_kernel<true>, dim3((cellsR + 255) / 256, rows, pr.n), dim3(256), 0, c->stream, f, R,
                     (const float4*)c->tri_raw_buf[slot], c->partner_cnt[slot], pr, 1);
  hipLaunchKernelGGL(partner_count_kernel<false>, dim3((cells0 + 255) / 256, 2 * R + 1, pr.n), dim3(256), 0, c->stream, f, R,
                     (const float4*)c->tri_raw_buf[slot], c->partner_cnt[slot], pr, 1);
  hipLaunchKernelGGL(partner_flags_from_counts_kernel, dim3((cellsR + 255) / 256, pr.n), dim3(256), 0, c->stream, f.nW, f.nD,
                     pr, R, (const unsigned*)c->partner_cnt[slot], c->partner_buf[slot]);
  HIP_TRY(c, hipGetLastError());
  c->partner_R_built[slot] = R;
  f.partner_flags = c->partner_buf[slot];
  f.partner_R = R;
  return ARTP_OK;
}

// Range tables of one layer (pipeline.h).  Levels 0..5 (block 1..32); the kernels use levels 2..5.  pr = the changed
// rectangles of a rectangle update whose step 1 (partner_update_begin) has run, nullptr = new layer.
int build_tables(artp_ctx* c, int slot, const PartnerRects* pr = nullptr) {
  const FieldDev& f = c->field[slot];
  const size_t elems = (size_t)f.nW * f.nD;
  if (c->table_elems[slot] < elems) {
    if (c->table_buf[slot]) HIP_TRY(c, hipFree(c->table_buf[slot]));
    if (c->flag_buf[slot]) HIP_TRY(c, hipFree(c->flag_buf[slot]));
    if (c->stride_buf[slot]) HIP_TRY(c, hipFree(c->stride_buf[slot]));
    c->stride_buf[slot] = nullptr;
    if (c->partner_buf[slot]) HIP_TRY(c, hipFree(c->partner_buf[slot]));
    if (c->partner_cnt[slot]) HIP_TRY(c, hipFree(c->partner_cnt[slot]));
    c->partner_cnt[slot] = nullptr;
    HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->partner_cnt[slot]), elems * sizeof(unsigned)));
    if (c->tri_raw_buf[slot]) HIP_TRY(c, hipFree(c->tri_raw_buf[slot]));
    c->tri_raw_buf[slot] = nullptr;
    HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->tri_raw_buf[slot]), elems * sizeof(float4)));
    c->table_buf[slot] = nullptr;
    c->flag_buf[slot] = nullptr;
    c->partner_buf[slot] = nullptr;
    HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->partner_buf[slot]), (elems + 3) / 4 * 4));
    c->partner_R_built[slot] = -1;
    HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->table_buf[slot]), 12 * elems * sizeof(float)));
    HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->flag_buf[slot]), 6 * elems));
    // stride tables: (nW / s) x (nD / s) entries for s = 2, 4, 8 -- less than elems / 2 entries in all
    HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->stride_buf[slot]), (elems / 2 + 3 * (f.nW + f.nD) + 16) * sizeof(unsigned)));
    c->table_elems[slot] = elems;
  }
  float2* mm[6];
  unsigned char* fl[6];
  for (int l = 0; l < 6; ++l) {
    mm[l] = reinterpret_cast<float2*>(c->table_buf[slot]) + (size_t)l * elems;
    fl[l] = reinterpret_cast<unsigned char*>(c->flag_buf[slot]) + (size_t)l * elems;
  }
  const int n = (int)elems;
  hipLaunchKernelGGL(table_level0_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, f.data, n, mm[0], fl[0]);
  for (int l = 1; l < 6; ++l)
    hipLaunchKernelGGL(table_level_up_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream,
                       (const float2*)mm[l - 1], (const unsigned char*)fl[l - 1], f.nW, f.nD, 1 << (l - 1), mm[l],
                       fl[l]);
  HIP_TRY(c, hipGetLastError());
  TablesDev& t = c->tables[slot];
  t.mm = mm[2];  // the kernels use levels 2..5 (blocks of 4..32 samples), stored back to back
  t.fl = fl[2];
  t.stride = (unsigned)elems;
  t.has_nan = f.has_nan;
  t.has_nonfinite = c->layer_has_nonfinite[slot];
  {
    unsigned off[ARTP_STRIDE_LEVELS + 1] = {0};
    for (int l = 0; l < ARTP_STRIDE_LEVELS; ++l) {
      const int s = 2 << l;
      off[l + 1] = off[l] + (unsigned)(((f.nW + s - 1) / s) * ((f.nD + s - 1) / s));
    }
    t.st = c->stride_buf[slot];
    t.st_off1 = off[1];
    t.st_off2 = off[2];
    hipLaunchKernelGGL(stride_tables_kernel, dim3((off[3] + 255) / 256), dim3(256), 0, c->stream, t.mm, t.fl, t.stride,
                       f.nW, f.nD, off[1], off[2], off[3], c->stride_buf[slot]);
    HIP_TRY(c, hipGetLastError());
  }
  // a rectangle update that met a table not built for this R (or too many / too large rectangles) has pr->n = 0 and
  // rebuilds the whole table
  const int rc_partner = pr ? partner_update_end(c, slot, *pr) : build_partner_table_full(c, slot);
  if (rc_partner != ARTP_OK) return rc_partner;
  t.valid = 1;
  return ARTP_OK;
}

// Room for the PoseRecs of a batch (tmp[7]).
int ensure_recs(artp_ctx* c, size_t n) { return ensure_tmp(c, 7, n * sizeof(PoseRec)); }

// classify -> resolve -> plane stage on the context's stream (all asynchronous).  recs_ready: the PoseRecs of
// the batch are already in tmp[7] (the fused sampler wrote them); otherwise pose_rec_kernel makes them from se3.
int launch_validate_pipeline(artp_ctx* c, const double* se3, size_t n, uint8_t* valid, bool recs_ready = false) {
  if (n >= (1ull << 32)) {
    c->last_error = "batch too large (state index is 32 bit)";
    return ARTP_ERR_INVALID_ARG;
  }
  int rc = ensure_recs(c, n);
  if (rc) return rc;
  PoseRec* recs = static_cast<PoseRec*>(c->tmp[7]);
  if (!recs_ready)
    hipLaunchKernelGGL(pose_rec_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->field[0], se3, n,
                       recs);
  const size_t per_block_states = 64 * ARTP_CLASSIFY_SUB;
  const size_t n_blocks = (n + per_block_states - 1) / per_block_states;
  // sub-queue capacity: every workgroup of a sub-queue may queue all of its boxes
  const size_t seg_t = ((n_blocks + ARTP_NSUB - 1) / ARTP_NSUB) * per_block_states;
  rc = ensure_tmp(c, 4, 5 * ARTP_NSUB * seg_t * sizeof(PendingBox));
  if (rc) return rc;
  // tmp[5]: 8 counters | pad to 128 B | 2 * ARTP_NSUB sub-queue counters, one per 128-byte line | index queues
  const size_t ctr_bytes = 128 + (size_t)2 * ARTP_NSUB * 128;
  // index queues: q2 (5 n), q3 (4 n), q5 (4 n) | q4 and q6 in ARTP_NSUB segments like the big queues (4 seg_t / seg_t each)
  rc = ensure_tmp(c, 5, ctr_bytes + ((5 + 4 + 4) * n + 5 * (size_t)ARTP_NSUB * seg_t) * sizeof(unsigned) + 64);
  if (rc) return rc;
  PipelineQueues q;
  q.q1 = static_cast<PendingBox*>(c->tmp[4]);
  q.counters = static_cast<unsigned long long*>(c->tmp[5]);
  q.sub = q.counters + 16;
  q.seg_t = seg_t;
  q.q2 = reinterpret_cast<unsigned*>(static_cast<char*>(c->tmp[5]) + ctr_bytes);
  q.q3 = q.q2 + 5 * n;
  q.q5 = q.q3 + 4 * n;
  q.q4 = q.q5 + 4 * n;
  q.q6 = q.q4 + 4 * (size_t)ARTP_NSUB * seg_t;
  q.feet_base = (unsigned long long)ARTP_NSUB * seg_t;
  HIP_TRY(c, hipMemsetAsync(q.counters, 0, ctr_bytes, c->stream));
  const size_t per_block = 64 * ARTP_CLASSIFY_SUB;
  hipLaunchKernelGGL(classify_states_kernel, dim3((unsigned)((n + per_block - 1) / per_block)),
                     dim3(ARTP_CLASSIFY_THREADS), 0, c->stream,
                     c->field[0], c->field[1], c->tables[0], c->tables[1], c->geom, c->robot, (const PoseRec*)recs, n,
                     valid, q);
#ifdef ARTP_VARIANTS
  if (c->feet_dense)   // round 5 experiment: the corner stage's plane / contact arithmetic on dense lanes (pipeline_variants.h)
    hipLaunchKernelGGL(feet_stream2_kernel<ARTP_STREAM_WAVES>, dim3(grid_sub(c, ARTP_FEET_WAVES_PER_SIMD)), dim3(64 * ARTP_STREAM_WAVES), 0,
                       c->stream, c->field[1], c->robot, q, valid);
  else
#endif
    hipLaunchKernelGGL(feet_stream_kernel<ARTP_STREAM_WAVES>, dim3(grid_sub(c, ARTP_FEET_WAVES_PER_SIMD)), dim3(64 * ARTP_STREAM_WAVES), 0,
                       c->stream, c->field[1], c->robot, q, valid);
  hipLaunchKernelGGL(feet_lane_kernel, dim3((unsigned)c->n_cus * 8), dim3(ARTP_LANE_THREADS), 0, c->stream,
                     c->field[1], c->robot, q, valid);
  {
    // streaming pass: no LDS at all; the staged pass behind it takes the window tile and the list
    const ScratchCaps caps_stream{0, 0, 0, 0};
    const size_t lds_stream = 0;
    hipLaunchKernelGGL((resolve_boxes_kernel<ARTP_WAVES_PER_BLOCK, 64, 0>), dim3(grid_sub(c, ARTP_TORSO_WGS_PER_CU)),
                       dim3(64 * ARTP_WAVES_PER_BLOCK), lds_stream, c->stream, c->field[0], c->robot, q, valid,
                       caps_stream, c->d_error, c->tables[0].valid ? c->tables[0].mm : (const float2*)nullptr);
  }
  hipLaunchKernelGGL((resolve_boxes_kernel<ARTP_WAVES_PER_BLOCK, 64, 3>), dim3(grid_scan(c, lds_scan(c))),
                     dim3(64 * ARTP_WAVES_PER_BLOCK), lds_scan(c), c->stream, c->field[0], c->robot, q, valid,
                     c->caps_scan, c->d_error);
  hipLaunchKernelGGL((resolve_boxes_kernel<ARTP_WAVES_PER_BLOCK, 16, 1>), dim3(grid_scan(c, lds_feet(c))),
                     dim3(64 * ARTP_WAVES_PER_BLOCK), lds_feet(c), c->stream, c->field[1], c->robot, q, valid,
                     c->caps_feet, c->d_error);
  // list pass: few boxes (dozens on natural terrain), each a long chain -> a whole wavefront per box
  hipLaunchKernelGGL((resolve_boxes_kernel<ARTP_WAVES_PER_BLOCK, 64, 2>), dim3(grid_scan(c, lds_feet_wave(c))),
                     dim3(64 * ARTP_WAVES_PER_BLOCK), lds_feet_wave(c), c->stream, c->field[1], c->robot, q, valid,
                     c->caps_feet_wave, c->d_error);
  hipLaunchKernelGGL(plane_stage_kernel<1>, dim3(grid_full(c, 0)), dim3(64), lds_full(c), c->stream,
                     c->field[0], c->field[1], c->robot, q, valid, c->caps_full, c->d_error);
  HIP_TRY(c, hipGetLastError());
  return ARTP_OK;
}

int check_error_flag(artp_ctx* c) {
  int flag = 0;
  HIP_TRY(c, hipMemcpyAsync(&flag, c->d_error, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (flag) {
    HIP_TRY(c, hipMemsetAsync(c->d_error, 0, sizeof(int), c->stream));
    c->last_error = "a box window exceeded the LDS tile capacity";
    return ARTP_ERR_CAPACITY;
  }
  return ARTP_OK;
}

}  // namespace

extern "C" {

void artp_params_defaults(artp_params* p) {  // art_planner/include/art_planner/params.h:80-119
  p->torso_length = 1.05; p->torso_width = 0.55; p->torso_height = 0.2;
  p->torso_off_x = 0.0; p->torso_off_y = 0.0; p->torso_off_z = 0.0;
  p->feet_off_x = 0.362; p->feet_off_y = 0.225; p->feet_off_z = -0.525;
  p->reach_x = 0.25; p->reach_y = 0.1; p->reach_z = 0.15;
  p->unknown_space_untraversable = 1;
  p->max_pitch_pert = 10.0 / 180 * M_PI;
  p->max_roll_pert = 3.33 / 180 * M_PI;
  p->sample_from_distribution = 1;
}

void artp_params_yaml(artp_params* p) {  // art_planner_ros/config/params.yaml:44-45,55-71
  p->torso_length = 1.31; p->torso_width = 0.65; p->torso_height = 0.3;
  p->torso_off_x = 0.0; p->torso_off_y = 0.0; p->torso_off_z = 0.04;
  p->feet_off_x = 0.51; p->feet_off_y = 0.2; p->feet_off_z = -0.475;
  p->reach_x = 0.2; p->reach_y = 0.2; p->reach_z = 0.2;
  p->unknown_space_untraversable = 1;
  p->max_pitch_pert = 10 * M_PI / 180;
  p->max_roll_pert = 3.33 * M_PI / 180;
  p->sample_from_distribution = 1;
}

const char* artp_status_string(int s) {
  switch (s) {
    case ARTP_OK: return "ok";
    case ARTP_ERR_INVALID_ARG: return "invalid argument";
    case ARTP_ERR_NO_DEVICE: return "no usable HIP device (this library has no CPU fallback)";
    case ARTP_ERR_HIP: return "HIP runtime error";
    case ARTP_ERR_NO_MAP: return "required layer not uploaded";
    case ARTP_ERR_CAPACITY: return "box window exceeds the LDS tile capacity";
    case ARTP_ERR_NO_WEIGHTS: return "motion-cost weights not loaded";
    case ARTP_ERR_TIMEOUT: return "timed out waiting for a device group";
    case ARTP_ERR_COMM: return "RCCL unavailable or a communicator call failed";
    case ARTP_ERR_COST_FUNC: return "Motion cost call failed";
    default: return "unknown status";
  }
}

const char* artp_last_error(const artp_ctx* ctx) { return ctx ? ctx->last_error.c_str() : ""; }
const char* artp_device_arch(const artp_ctx* ctx) { return ctx ? ctx->arch.c_str() : ""; }

// ---- persistent latency service (kernels.h validate_service_kernel) ------------------------------------------------
// Tell the resident workgroup to leave and wait until it has (it polls `quit` every few microseconds; bounded by its own
// idle / lifetime limits whatever happens).
static void pool_stop(artp_ctx* c);
static void svc_stop(artp_ctx* c) {
  if (!c->svc || !c->svc_launched) return;
  c->svc->n_quit = 0x100u;
  std::atomic_thread_fence(std::memory_order_seq_cst);
  const auto t0 = std::chrono::steady_clock::now();
  while (c->svc->running && std::chrono::steady_clock::now() - t0 < std::chrono::milliseconds(200))
    ;
  (void)hipStreamSynchronize(c->svc_stream);   // the kernel has returned (at the latest after its lifetime limit)
  c->svc_launched = false;
}

static int svc_start(artp_ctx* c) {
  if (!c->svc) {
    void *p = nullptr, *pd = nullptr;
    HIP_TRY(c, hipHostMalloc(&p, sizeof(SvcMailbox), hipHostMallocMapped));
    HIP_TRY(c, hipHostGetDevicePointer(&pd, p, 0));
    std::memset(p, 0, sizeof(SvcMailbox));
    c->svc = static_cast<SvcMailbox*>(p);
    c->svc_dev = static_cast<SvcMailbox*>(pd);
    HIP_TRY(c, hipStreamCreateWithFlags(&c->svc_stream, hipStreamNonBlocking));
    HIP_TRY(c, hipEventCreateWithFlags(&c->svc_after_map, hipEventDisableTiming));
    HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(validate_service_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_few(c)));
  }
  // behind every map write already enqueued on the context's stream: the kernel captures the field arguments of NOW
  HIP_TRY(c, hipEventRecord(c->svc_after_map, c->stream));
  HIP_TRY(c, hipStreamWaitEvent(c->svc_stream, c->svc_after_map, 0));
  c->svc->n_quit = 0u;
  c->svc->running = 1u;
  c->svc->resp = c->svc_seq << 8;
  c->svc->req_seq = c->svc_seq;
  std::atomic_thread_fence(std::memory_order_seq_cst);
  hipLaunchKernelGGL(validate_service_kernel, dim3(1), dim3(320), lds_few(c), c->svc_stream, c->field[0], c->field[1], c->geom,
                     c->robot, c->svc_dev, c->svc_seq, c->caps_full, c->caps_foot_full);
  HIP_TRY(c, hipGetLastError());
  c->svc_launched = true;
  c->svc_map_version = c->map_version.load(std::memory_order_acquire);
  ++c->svc_launches;
  return ARTP_OK;
}

// n <= 16 states through the resident workgroup.  ARTP_ERR_TIMEOUT only if the device does not answer at all.
static int svc_validate(artp_ctx* c, const double* se3, size_t n, uint8_t* valid) {
  for (int attempt = 0; attempt < 3; ++attempt) {
    const bool stale = c->svc_launched && c->svc_map_version != c->map_version.load(std::memory_order_acquire);
    if (stale) svc_stop(c);                              // a map write since the launch: its field arguments are history
    if (c->svc_launched && !c->svc->running) {            // it left on its own (idle / lifetime limit)
      (void)hipStreamSynchronize(c->svc_stream);
      c->svc_launched = false;
    }
    if (!c->svc_launched) {
      const int rc = svc_start(c);
      if (rc) return rc;
    }
    if (n > 1) std::memcpy(c->svc->state1, se3 + 7, 7 * sizeof(double));
    std::memcpy(c->svc->state0, se3, 7 * sizeof(double));
    c->svc->n_quit = (uint32_t)n;
    // the state is globally visible before the number, and the number goes out NOW: a full fence on both sides
    std::atomic_thread_fence(std::memory_order_seq_cst);
    const uint32_t seq = ++c->svc_seq;
    c->svc->req_seq = seq;
    std::atomic_thread_fence(std::memory_order_seq_cst);
    const auto t0 = std::chrono::steady_clock::now();
    bool answered = false, gone = false;
    for (unsigned spin = 0;; ++spin) {
      if ((c->svc->resp >> 8) == (seq & 0xffffffu)) { answered = true; break; }
      if ((spin & 255u) == 255u) {
        if (!c->svc->running) { gone = true; break; }     // it left between our check and our request: start it again
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(50)) break;
      }
    }
    if (!answered && gone && (c->svc->resp >> 8) == (seq & 0xffffffu)) answered = true;   // it answered in its last breath
    if (answered) {
      const uint32_t r = c->svc->resp;   // one word: the labels came with the number
      bool overflow = false;
      for (size_t i = 0; i < n; ++i) {
        const unsigned b = (r >> (2 * i)) & 3u;
        valid[i] = b & 1u;
        overflow = overflow || (b & 2u);
      }
      ++c->svc_requests;
      if (overflow) {
        c->last_error = "a box window exceeded the LDS tile capacity";
        return ARTP_ERR_CAPACITY;
      }
      return ARTP_OK;
    }
    if (gone) {
      (void)hipStreamSynchronize(c->svc_stream);
      c->svc_launched = false;
      --c->svc_seq;   // the request was never seen: post it again under the same number
      continue;
    }
    c->last_error = "the persistent latency service did not answer within 50 ms";
    svc_stop(c);
    return ARTP_ERR_TIMEOUT;
  }
  c->last_error = "the persistent latency service could not be (re)started";
  return ARTP_ERR_TIMEOUT;
}

int artp_set_persistent_latency(artp_ctx* c, int enabled) {
  if (!c) return ARTP_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  HIP_TRY(c, hipSetDevice(c->device));
  c->svc_enabled = enabled != 0;
  if (!c->svc_enabled) {
    svc_stop(c);
    pool_stop(c);
  }
  return ARTP_OK;
}

int artp_persistent_latency_stats(artp_ctx* c, uint64_t out[2]) {
  if (!c || !out) return ARTP_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  out[0] = c->svc_launches + c->pool_launches;
  out[1] = c->svc_requests + c->pool_requests;
  return ARTP_OK;
}

int artp_create(int device, const artp_params* params, artp_ctx** out) {
  if (!params || !out) return ARTP_ERR_INVALID_ARG;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count)
    return ARTP_ERR_NO_DEVICE;
  artp_ctx* c = new artp_ctx();
  c->device = device;
  c->params = *params;
  if (hipSetDevice(device) != hipSuccess) {
    delete c;
    return ARTP_ERR_NO_DEVICE;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) {
    delete c;
    return ARTP_ERR_NO_DEVICE;
  }
  c->arch = prop.gcnArchName;
  c->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess ||
      hipMalloc(&c->d_error, sizeof(int)) != hipSuccess ||
      hipMalloc(&c->d_count, sizeof(unsigned long long)) != hipSuccess ||
      hipMalloc(reinterpret_cast<void**>(&c->d_diff), 8 * sizeof(int)) != hipSuccess ||
      hipMemset(c->d_error, 0, sizeof(int)) != hipSuccess) {
    artp_destroy(c);
    return ARTP_ERR_HIP;
  }
  c->stream = c->own_stream;
  {
    void *ps = nullptr, *pl = nullptr, *psd = nullptr, *pld = nullptr;
    if (hipHostMalloc(&ps, ARTP_FEW_STATES * 7 * sizeof(double), hipHostMallocMapped) != hipSuccess ||
        hipHostMalloc(&pl, 64, hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer(&psd, ps, 0) != hipSuccess || hipHostGetDevicePointer(&pld, pl, 0) != hipSuccess) {
      if (ps) (void)hipHostFree(ps);
      if (pl) (void)hipHostFree(pl);
      artp_destroy(c);
      return ARTP_ERR_HIP;
    }
    {
      void *pe = nullptr, *ped = nullptr;
      std::vector<FewEdgeSync> armed(ARTP_FEW_EDGES, FewEdgeSync{0xffffffffu, 0u, 0u, {0u}});
      void *pi = nullptr, *pid = nullptr;
      // the edges (host -> device) in a block of their own, apart from the polled results.  Plain coherent mapped memory:
      // a non-coherent mapping (L2-cacheable for the length of a kernel, so that only the first workgroup's read of an
      // edge crosses PCIe) was measured -- no difference at any call size -- and is not worth a second coherence rule.
      if (hipHostMalloc(&pi, 2 * ARTP_FEW_EDGES * 7 * sizeof(double), hipHostMallocMapped) == hipSuccess &&
          hipHostGetDevicePointer(&pid, pi, 0) == hipSuccess) {
        c->pin_edges_in = static_cast<char*>(pi);
        c->pin_edges_in_dev = static_cast<char*>(pid);
      } else if (pi) {
        (void)hipHostFree(pi);
      }
      if (!c->pin_edges_in ||
          hipHostMalloc(&pe, FEW_EDGE_BLOCK_BYTES, hipHostMallocMapped) != hipSuccess ||
          hipHostGetDevicePointer(&ped, pe, 0) != hipSuccess ||
          hipMalloc(reinterpret_cast<void**>(&c->d_few_sync), ARTP_FEW_EDGES * sizeof(FewEdgeSync)) != hipSuccess ||
          hipMemcpy(c->d_few_sync, armed.data(), ARTP_FEW_EDGES * sizeof(FewEdgeSync), hipMemcpyHostToDevice) != hipSuccess) {
        if (pe) (void)hipHostFree(pe);
        (void)hipHostFree(ps);
        (void)hipHostFree(pl);
        artp_destroy(c);
        return ARTP_ERR_HIP;
      }
      c->pin_edges = static_cast<char*>(pe);
      c->pin_edges_dev = static_cast<char*>(ped);
    }
    c->pin_states = static_cast<double*>(ps);
    c->pin_labels = static_cast<volatile uint8_t*>(pl);
    c->pin_states_dev = static_cast<double*>(psd);
    c->pin_labels_dev = static_cast<uint8_t*>(pld);
    const char* e = std::getenv("ARTP_NO_POLL");   // one of the library's three environment variables (include/artp_c.h)
    c->poll_labels = !(e && e[0] == '1');
#ifdef ARTP_VARIANTS
    if (const char* fd = std::getenv("ARTP_FEET_DENSE")) c->feet_dense = fd[0] != '0';
    if (const char* pw = std::getenv("ARTP_POOL_WGS")) {
      const long v = std::strtol(pw, nullptr, 10);
      if (v >= 1 && v <= ARTP_POOL_MAX_WGS) c->pool_wgs = (unsigned)v;
    }
#endif
  }
  fill_robot(c);
  *out = c;
  return ARTP_OK;
}

void artp_destroy(artp_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  svc_stop(c);   // the resident latency workgroups (if any) leave before anything they read is freed
  pool_stop(c);
  park_lane(c);
  for (auto& l : c->lanes)
    if (l.init && l.stream) (void)hipStreamSynchronize(l.stream);
  for (int s = 0; s < 2; ++s)
    if (c->field_data[s]) (void)hipFree(c->field_data[s]);
  if (c->sampler_buf) (void)hipFree(c->sampler_buf);
  if (c->sampler_pack) (void)hipFree(c->sampler_pack);
  if (c->rect_stage_host) (void)hipHostFree(c->rect_stage_host);
  if (c->rect_stage_dev) (void)hipFree(c->rect_stage_dev);
  if (c->rect_stage_done) (void)hipEventDestroy(c->rect_stage_done);
  if (c->map_written) (void)hipEventDestroy(c->map_written);
  for (auto& e : c->lane_mark)
    if (e) (void)hipEventDestroy(e);
  for (auto& l : c->lanes) {
    if (!l.init) continue;
    for (int s = 0; s < 8; ++s)
      if (l.tmp[s]) (void)hipFree(l.tmp[s]);
    if (l.cub_tmp) (void)hipFree(l.cub_tmp);
    if (l.d_count) (void)hipFree(l.d_count);
    if (l.d_error) (void)hipFree(l.d_error);
    if (l.own_stream) (void)hipStreamDestroy(l.own_stream);
  }
  for (int s = 0; s < 2; ++s) {
    if (c->table_buf[s]) (void)hipFree(c->table_buf[s]);
    if (c->flag_buf[s]) (void)hipFree(c->flag_buf[s]);
    if (c->stride_buf[s]) (void)hipFree(c->stride_buf[s]);
    if (c->partner_buf[s]) (void)hipFree(c->partner_buf[s]);
    if (c->partner_cnt[s]) (void)hipFree(c->partner_cnt[s]);
    if (c->tri_raw_buf[s]) (void)hipFree(c->tri_raw_buf[s]);
  }
  for (int l = 0; l < 5; ++l) {
    if (c->d_convw[l]) (void)hipFree(c->d_convw[l]);
    if (c->d_convb[l]) (void)hipFree(c->d_convb[l]);
  }
  if (c->d_fc) (void)hipFree(c->d_fc);
  if (c->d_fc_mfma) (void)hipFree(c->d_fc_mfma);
  if (c->d_c12) (void)hipFree(c->d_c12);
  if (c->d_c12m) (void)hipFree(c->d_c12m);
  for (int l = 0; l < 3; ++l)
    if (c->d_convw_chunk[l]) (void)hipFree(c->d_convw_chunk[l]);
  if (c->d_convw_p32) (void)hipFree(c->d_convw_p32);
  for (int l = 0; l < 2; ++l)
    if (c->d_act[l]) (void)hipFree(c->d_act[l]);
  if (c->d_feat) (void)hipFree(c->d_feat);
  if (c->d_map_f32) (void)hipFree(c->d_map_f32);
  if (c->d_diff) (void)hipFree(c->d_diff);
  if (c->svc) (void)hipHostFree(c->svc);
  if (c->svc_stream) (void)hipStreamDestroy(c->svc_stream);
  if (c->svc_after_map) (void)hipEventDestroy(c->svc_after_map);
  if (c->pool_stream) (void)hipStreamDestroy(c->pool_stream);
  if (c->pool_after_map) (void)hipEventDestroy(c->pool_after_map);
  if (c->pool_mb) (void)hipFree(c->pool_mb);
  if (c->pool_resp) (void)hipHostFree(c->pool_resp);
  if (c->pool_ctl) (void)hipFree(c->pool_ctl);
  if (c->pin_edges) (void)hipHostFree(c->pin_edges);
  if (c->pin_edges_in) (void)hipHostFree(c->pin_edges_in);
  if (c->d_few_sync) (void)hipFree(c->d_few_sync);
  if (c->pin_states) (void)hipHostFree(c->pin_states);
  if (c->pin_labels) (void)hipHostFree(const_cast<uint8_t*>(c->pin_labels));
  delete c;  // d_error / d_count of every lane went with the lanes above
}

int artp_set_stream(artp_ctx* c, void* hip_stream) {
  if (!c) return ARTP_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  c->stream = static_cast<hipStream_t>(hip_stream);  // NULL = HIP's legacy default stream
  if (c->map_written) {  // a stream this context has not used before: order it behind the last map write
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamWaitEvent(c->stream, c->map_written, 0));
  }
  return ARTP_OK;
}

int artp_use_own_stream(artp_ctx* c) {
  if (!c) return ARTP_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  c->stream = c->own_stream;
  if (c->map_written) {
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamWaitEvent(c->stream, c->map_written, 0));
  }
  return ARTP_OK;
}

int artp_synchronize(artp_ctx* c) {
  if (!c) return ARTP_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  for (int l = 0; l < ARTP_MAX_LANES; ++l)
    if (l != c->cur_lane && c->lanes[l].init) HIP_TRY(c, hipStreamSynchronize(c->lanes[l].stream));
  return ARTP_OK;
}

int artp_set_lane(artp_ctx* c, int lane) {
  if (!c || lane < 0 || lane >= ARTP_MAX_LANES) return ARTP_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  if (lane == c->cur_lane) return ARTP_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  if (!c->lanes[lane].init) {
    artp_ctx::lane_state l;
    HIP_TRY(c, hipStreamCreateWithFlags(&l.own_stream, hipStreamNonBlocking));
    if (hipMalloc(&l.d_count, sizeof(unsigned long long)) != hipSuccess ||
        hipMalloc(&l.d_error, sizeof(int)) != hipSuccess || hipMemset(l.d_error, 0, sizeof(int)) != hipSuccess) {
      if (l.d_count) (void)hipFree(l.d_count);
      if (l.d_error) (void)hipFree(l.d_error);
      (void)hipStreamDestroy(l.own_stream);
      c->last_error = "hipMalloc failed (lane counters)";
      return ARTP_ERR_HIP;
    }
    l.stream = l.own_stream;
    l.init = true;
    c->lanes[lane] = l;
  }
  park_lane(c);
  unpark_lane(c, lane);
  return lane_sees_map(c);
}

int artp_get_lane(artp_ctx* c) { return c ? c->cur_lane : -1; }

uint64_t artp_map_version(const artp_ctx* c) { return c ? c->map_version.load(std::memory_order_acquire) : 0; }

namespace {

// layer(i, j) column-major rows x cols  ->  ODE sample layout data[x + z * rows] = layer(x, cols-1-z)
// (field_.mat = layer.rowwise().reverse(), height_map_box_checker.cpp:44), with the layer's non-finite /
// NaN presence OR-ed into flags[0] / flags[1]
__global__ void __launch_bounds__(256)
flip_to_ode_layout_kernel(const float* __restrict__ layer, int rows, int cols, float* __restrict__ data,
                          int* __restrict__ flags) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= rows * cols) return;
  const int x = t % rows, z = t / rows;
  const float v = layer[(size_t)x + (size_t)(cols - 1 - z) * rows];
  data[t] = v;
  if (!artp::is_finite(v)) atomicOr(&flags[0], 1);
  if (v != v) atomicOr(&flags[1], 1);
}

// min / max over the finite samples (order-preserving integer keys), for the z bounds of planner.cpp:146-156
__device__ __forceinline__ int float_order_key(float v) {
  const int i = __float_as_int(v);
  return i >= 0 ? i : i ^ 0x7fffffff;
}
__global__ void __launch_bounds__(256)
finite_min_max_kernel(const float* __restrict__ layer, int n, int* __restrict__ keys) {
  // grid-stride, one pair of atomics per WORKGROUP (a pair per wavefront of a wavefront-per-256-samples grid kept the
  // two words' atomic unit busy for the kernel's whole 0.08 ms)
  __shared__ int wlo[4], whi[4];
  int lo = 0x7fffffff, hi = (int)0x80000000;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
    const float v = layer[t];
    if (artp::is_finite(v)) {
      const int k = float_order_key(v);
      lo = min(lo, k);
      hi = max(hi, k);
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    lo = min(lo, __shfl_xor(lo, off, 64));
    hi = max(hi, __shfl_xor(hi, off, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    wlo[threadIdx.x >> 6] = lo;
    whi[threadIdx.x >> 6] = hi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicMin(&keys[0], min(min(wlo[0], wlo[1]), min(wlo[2], wlo[3])));
    atomicMax(&keys[1], max(max(whi[0], whi[1]), max(whi[2], whi[3])));
  }
}

// Everything of artp_upload_layer that follows the sample data being in c->field_data[slot] (ODE layout) and
// dxHeightfieldData::SetData, the checker's frame, scratch sizing, tables.
int finish_layer(artp_ctx* c, int slot, int rows, int cols, double len_x, double len_y, double pos_x, double pos_y,
                 int has_nan, int has_nonfinite) {
  c->layer_has_nonfinite[slot] = has_nonfinite;
  FieldDev& f = c->field[slot];
  f.data = c->field_data[slot];
  // dxHeightfieldData::SetData (ode/ode/src/heightfield.cpp:130-169), single precision
  f.nW = rows;
  f.nD = cols;
  f.width = (float)len_x;
  f.depth = (float)len_y;
  f.half_w = f.width / 2.0f;
  f.half_d = f.depth / 2.0f;
  f.sample_w = f.width / ((float)f.nW - 1.0f);
  f.sample_d = f.depth / ((float)f.nD - 1.0f);
  f.zx_aspect = f.sample_d / f.sample_w;
  f.inv_w = 1.0f / f.sample_w;
  f.inv_d = 1.0f / f.sample_d;
  f.pos[0] = (float)pos_x;
  f.pos[1] = (float)pos_y;
  f.pos[2] = 0.0f;
  std::memset(f.R, 0, sizeof(f.R));
  r_from_2_axes(f.R, -1, 0, 0, 0, 0, 1);  // height_map_box_checker.cpp:22
  orthogonalize_R(f.R);                  // dBodySetRotation, :25
  f.has_nan = has_nan;
  {
    // relative spread of the constant cross component (sample_w * sample_d) between cells: the sample
    // coordinates are products float(index) * spacing, so a coordinate difference carries an absolute
    // error of ~2 ulp(map length); see DESIGN.md 4.1 for the bound partner_tol = 4 * (delta + 1e-5)
    const double map_len = std::fmax((double)f.width, (double)f.depth);
    const double spacing = std::fmin((double)f.sample_w, (double)f.sample_d);
    const double delta = 2.0 * (2.0 * map_len * 1.1920929e-07) / spacing;
    const double tol = 4.0 * (delta + 1e-5);
    f.partner_tol = tol < 0.05 ? (float)std::fmax(tol, 2e-3) : INFINITY;
  }
  c->have_field[slot] = true;
  c->geom.len_x = len_x;
  c->geom.len_y = len_y;
  c->geom.pos_x = pos_x;
  c->geom.pos_y = pos_y;
  c->geom.rows = rows;
  c->geom.cols = cols;
  c->geom.res = len_x / rows;
  c->have_geom = true;
  int rc = size_scratch(c);
  if (rc != ARTP_OK) return rc;
  rc = set_kernel_lds(c);
  if (rc != ARTP_OK) return rc;
  c->map_version.fetch_add(1, std::memory_order_release);  // even when the tables fail: the samples did change
  rc = build_tables(c, slot);
  if (rc != ARTP_OK) return rc;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return ARTP_OK;
}

int ensure_field_storage(artp_ctx* c, int slot, size_t elems) {
  if (c->field_elems[slot] < elems) {
    if (c->field_data[slot]) HIP_TRY(c, hipFree(c->field_data[slot]));
    c->field_data[slot] = nullptr;
    HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->field_data[slot]), elems * sizeof(float)));
    c->field_elems[slot] = elems;
  }
  return ARTP_OK;
}

}  // namespace

int artp_upload_layer(artp_ctx* c, int slot, const float* layer, int rows, int cols, double len_x,
                      double len_y, double pos_x, double pos_y) {
  if (!c || !layer || slot < 0 || slot > 1 || rows < 2 || cols < 2) return ARTP_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  HIP_TRY(c, hipSetDevice(c->device));
  const size_t elems = (size_t)rows * cols;
  // field_.mat = layer.rowwise().reverse() (height_map_box_checker.cpp:44): ODE sample (x, z) =
  // layer(x, cols-1-z), stored x-fastest.
  std::vector<float> host(elems);
  int has_nan = 0, has_nonfinite = 0;
  for (int j = 0; j < cols; ++j)
    for (int i = 0; i < rows; ++i) {
      const float v = layer[(size_t)i + (size_t)(cols - 1 - j) * rows];
      host[(size_t)i + (size_t)j * rows] = v;
      has_nan |= (v != v);
      has_nonfinite |= !std::isfinite(v);
    }
  int rc = order_after_other_lanes(c);  // lanes that still validate on the old samples
  if (rc != ARTP_OK) return rc;
  rc = ensure_field_storage(c, slot, elems);
  if (rc != ARTP_OK) return rc;
  HIP_TRY(c, hipMemcpyAsync(c->field_data[slot], host.data(), elems * sizeof(float),
                            hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  rc = finish_layer(c, slot, rows, cols, len_x, len_y, pos_x, pos_y, has_nan, has_nonfinite);
  if (rc != ARTP_OK) return rc;
  return publish_map_write(c);
}

// Difference between a device layer (column-major rows x cols) and the installed samples of a slot (ODE layout), bit for
// bit: out[0..3] = bounding rectangle {x0, z0, x1, z1} of the samples that differ (x0 > x1: none), out[4] / out[5] = the
// NEW layer holds a non-finite sample / a NaN.
__global__ void __launch_bounds__(256)
layer_diff_kernel(const float* __restrict__ layer, int rows, int cols, const float* __restrict__ data, int* __restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const bool in = t < rows * cols;
  const int x = in ? t % rows : 0, z = in ? t / rows : 0;
  const float v = in ? layer[(size_t)x + (size_t)(cols - 1 - z) * rows] : 0.0f;
  const bool differs = in && __float_as_uint(v) != __float_as_uint(data[t]);
  if (__any(in && !artp::is_finite(v)) && (threadIdx.x & 63) == 0) atomicOr(&out[4], 1);
  if (__any(in && v != v) && (threadIdx.x & 63) == 0) atomicOr(&out[5], 1);
  if (!__any(differs)) return;
  int x0 = differs ? x : 0x7fffffff, z0 = differs ? z : 0x7fffffff, x1 = differs ? x : -1, z1 = differs ? z : -1;
  for (int off = 32; off > 0; off >>= 1) {
    x0 = min(x0, __shfl_xor(x0, off, 64));
    z0 = min(z0, __shfl_xor(z0, off, 64));
    x1 = max(x1, __shfl_xor(x1, off, 64));
    z1 = max(z1, __shfl_xor(z1, off, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMin(&out[0], x0);
    atomicMin(&out[1], z0);
    atomicMax(&out[2], x1);
    atomicMax(&out[3], z1);
  }
}

// the samples [x0, x1] x [z0, z1] (ODE coordinates) of a device layer into the installed samples
__global__ void __launch_bounds__(256)
copy_rect_to_ode_layout_kernel(const float* __restrict__ layer, int rows, int cols, int x0, int z0, int nx, int nz,
                               float* __restrict__ data) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nx * nz) return;
  const int x = x0 + t % nx, z = z0 + t / nx;
  data[(size_t)x + (size_t)z * rows] = layer[(size_t)x + (size_t)(cols - 1 - z) * rows];
}

// Same from a layer that already lives in HBM (column-major, this context's device): the layout flip runs on the
// device.  When the slot already holds a layer of the same geometry, the new one is compared with it bit for bit
// first: no difference -> nothing to do; a difference confined to a quarter of the samples or less -> only that
// rectangle is rewritten and the tables take the rectangle-update path (what a 10 Hz map stream looks like: the
// partner table of the torso layer alone is 0.76 ms when rebuilt, a tenth of that for a small rectangle); the result is
// the same tables either way.
static int upload_layer_from_device(artp_ctx* c, int slot, const float* d_layer, int rows, int cols, double len_x,
                             double len_y, double pos_x, double pos_y) {
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  HIP_TRY(c, hipSetDevice(c->device));
  const size_t elems = (size_t)rows * cols;
  const FieldDev& f0 = c->field[slot];
  const bool same_geometry = c->have_field[slot] && c->have_geom && f0.nW == rows && f0.nD == cols &&
                             c->geom.len_x == len_x && c->geom.len_y == len_y && c->geom.pos_x == pos_x &&
                             c->geom.pos_y == pos_y && c->tables[slot].valid && c->field_elems[slot] >= elems;
  {
    const int rco = order_after_other_lanes(c);
    if (rco != ARTP_OK) return rco;
  }
  if (same_geometry) {
    int init[6] = {0x7fffffff, 0x7fffffff, -1, -1, 0, 0}, got[6];
    int* d_out = reinterpret_cast<int*>(c->d_diff);
    HIP_TRY(c, hipMemcpyAsync(d_out, init, sizeof(init), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(layer_diff_kernel, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, c->stream, d_layer, rows, cols,
                       (const float*)c->field_data[slot], d_out);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(got, d_out, sizeof(got), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (got[0] > got[2]) return ARTP_OK;  // bit-identical: the installed layer and its tables stand
    const int nx = got[2] - got[0] + 1, nz = got[3] - got[1] + 1;
    if ((size_t)nx * nz * 4 <= elems) {
      c->map_version.fetch_add(1, std::memory_order_release);
      const int dirty[4] = {got[0], got[1], got[2], got[3]};
      PartnerRects pr;
      int rc = partner_update_begin(c, slot, dirty, 1, &pr);  // the old triangles' contributions: before the copy
      if (rc != ARTP_OK) return rc;
      hipLaunchKernelGGL(copy_rect_to_ode_layout_kernel, dim3((unsigned)((nx * nz + 255) / 256)), dim3(256), 0, c->stream,
                         d_layer, rows, cols, got[0], got[1], nx, nz, c->field_data[slot]);
      HIP_TRY(c, hipGetLastError());
      c->field[slot].has_nan = got[5];          // exact: the flags are those of the whole new layer
      c->layer_has_nonfinite[slot] = got[4];
      rc = build_tables(c, slot, &pr);
      if (rc != ARTP_OK) return rc;
      return publish_map_write(c);  // asynchronous like the rectangle updates: the other lanes wait on the device
    }
  }
  int rc = ensure_field_storage(c, slot, elems);
  if (rc != ARTP_OK) return rc;
  HIP_TRY(c, hipMemsetAsync(c->d_count, 0, sizeof(unsigned long long), c->stream));
  int* d_flags = reinterpret_cast<int*>(c->d_count);
  hipLaunchKernelGGL(flip_to_ode_layout_kernel, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, c->stream, d_layer,
                     rows, cols, c->field_data[slot], d_flags);
  HIP_TRY(c, hipGetLastError());
  int flags[2] = {0, 0};
  HIP_TRY(c, hipMemcpyAsync(flags, d_flags, sizeof(flags), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  rc = finish_layer(c, slot, rows, cols, len_x, len_y, pos_x, pos_y, flags[1], flags[0]);
  if (rc != ARTP_OK) return rc;
  return publish_map_write(c);
}

// Rectangle updates (config 5).  One call takes any number of rectangles of one slot: the patches go through ONE
// pinned staging buffer and ONE host-to-device copy, a scatter kernel writes them into the ODE-layout samples, the
// range / stride tables are rebuilt once, the partner table per dirty rectangle.  Everything is asynchronous on the
// context's stream (round 2 issued one hipMemcpyAsync per grid column -- 52 per rectangle, 0.9 ms of a 1.5 ms cycle --
// re-scanned the whole layer on the host and synchronised the stream per rectangle).
namespace {
struct RectDev { int row0, col0, nrows, ncols; unsigned offset; };  // offset: first float of the patch in the staging buffer

__global__ void __launch_bounds__(256)
scatter_rects_kernel(const float* __restrict__ staged, const RectDev* __restrict__ rects, int n_rects, int rows, int cols,
                     float* __restrict__ data) {
  // blockIdx.y = rectangle; the patch is column-major nrows x ncols; ODE sample (x, z) = layer(x, cols - 1 - z)
  const RectDev r = rects[blockIdx.y];
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= r.nrows * r.ncols) return;
  const int i = t % r.nrows, jj = t / r.nrows;
  const int row = r.row0 + i, col = r.col0 + jj;
  // the rectangles of a call apply IN ORDER: a cell that a later rectangle also covers is that rectangle's to write
  for (int k = (int)blockIdx.y + 1; k < n_rects; ++k) {
    const RectDev q = rects[k];
    if (row >= q.row0 && row < q.row0 + q.nrows && col >= q.col0 && col < q.col0 + q.ncols) return;
  }
  const int z = cols - 1 - col;
  data[(size_t)row + (size_t)z * rows] = staged[r.offset + t];
}
}  // namespace

int artp_update_layer_rects(artp_ctx* c, int slot, int n_rects, const float* const* patches, const int* rects) {
  if (!c || slot < 0 || slot > 1 || n_rects < 0 || (n_rects && (!patches || !rects))) return ARTP_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  if (!c->have_field[slot]) return ARTP_ERR_NO_MAP;
  if (n_rects == 0) return ARTP_OK;
  const int rows = c->field[slot].nW, cols = c->field[slot].nD;
  size_t total = 0;
  int max_cells = 0;
  for (int k = 0; k < n_rects; ++k) {
    const int row0 = rects[4 * k], col0 = rects[4 * k + 1], nrows = rects[4 * k + 2], ncols = rects[4 * k + 3];
    if (!patches[k] || row0 < 0 || col0 < 0 || nrows <= 0 || ncols <= 0 || row0 + nrows > rows || col0 + ncols > cols)
      return ARTP_ERR_INVALID_ARG;
    total += (size_t)nrows * ncols;
    max_cells = std::max(max_cells, nrows * ncols);
  }
  HIP_TRY(c, hipSetDevice(c->device));
  {
    const int rco = order_after_other_lanes(c);
    if (rco != ARTP_OK) return rco;
  }
  c->map_version.fetch_add(1, std::memory_order_release);
  // staging: pinned host buffer (patches + rectangle records), reused once the previous update's copy is done
  const size_t rec_bytes = ((size_t)n_rects * sizeof(RectDev) + 15) & ~(size_t)15;
  const size_t need = rec_bytes + total * sizeof(float);
  if (c->rect_stage_cap < need) {
    if (c->rect_stage_host) {
      HIP_TRY(c, hipEventSynchronize(c->rect_stage_done));
      HIP_TRY(c, hipHostFree(c->rect_stage_host));
      HIP_TRY(c, hipFree(c->rect_stage_dev));
    } else {
      HIP_TRY(c, hipEventCreateWithFlags(&c->rect_stage_done, hipEventDisableTiming));
    }
    c->rect_stage_host = nullptr;
    c->rect_stage_dev = nullptr;
    c->rect_stage_cap = 0;
    const size_t cap = need * 2;
    HIP_TRY(c, hipHostMalloc(&c->rect_stage_host, cap, hipHostMallocDefault));
    HIP_TRY(c, hipMalloc(&c->rect_stage_dev, cap));
    c->rect_stage_cap = cap;
  } else {
    HIP_TRY(c, hipEventSynchronize(c->rect_stage_done));
  }
  RectDev* h_rec = static_cast<RectDev*>(c->rect_stage_host);
  float* h_pat = reinterpret_cast<float*>(static_cast<char*>(c->rect_stage_host) + rec_bytes);
  int has_nan = c->field[slot].has_nan, has_nonfinite = c->layer_has_nonfinite[slot];
  size_t off = 0;
  for (int k = 0; k < n_rects; ++k) {
    const int row0 = rects[4 * k], col0 = rects[4 * k + 1], nrows = rects[4 * k + 2], ncols = rects[4 * k + 3];
    h_rec[k] = RectDev{row0, col0, nrows, ncols, (unsigned)off};
    const size_t cells = (size_t)nrows * ncols;
    std::memcpy(h_pat + off, patches[k], cells * sizeof(float));
    // the flags may only stay set conservatively (a patch can overwrite the layer's last NaN): "non-finite present"
    // selects the general code path, never a wrong one
    for (size_t i = 0; i < cells; ++i) {
      const float v = patches[k][i];
      has_nan |= (v != v);
      has_nonfinite |= !std::isfinite(v);
    }
    off += cells;
  }
  c->field[slot].has_nan = has_nan;
  c->layer_has_nonfinite[slot] = has_nonfinite;
  // partner table, step 1: what the OLD triangles of the rectangles contribute to the cells around them comes off the
  // counts while the old samples are still there (stream order: in front of the scatter)
  std::vector<int> dirty((size_t)4 * n_rects);
  for (int k = 0; k < n_rects; ++k) {
    const int row0 = rects[4 * k], col0 = rects[4 * k + 1], nrows = rects[4 * k + 2], ncols = rects[4 * k + 3];
    dirty[4 * k] = row0;
    dirty[4 * k + 1] = cols - (col0 + ncols);
    dirty[4 * k + 2] = row0 + nrows - 1;
    dirty[4 * k + 3] = cols - 1 - col0;
  }
  PartnerRects pr;
  {
    const int rcp = partner_update_begin(c, slot, dirty.data(), n_rects, &pr);
    if (rcp != ARTP_OK) return rcp;
  }
  HIP_TRY(c, hipMemcpyAsync(c->rect_stage_dev, c->rect_stage_host, need, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(scatter_rects_kernel, dim3((unsigned)((max_cells + 255) / 256), (unsigned)n_rects), dim3(256), 0, c->stream,
                     reinterpret_cast<const float*>(static_cast<const char*>(c->rect_stage_dev) + rec_bytes),
                     static_cast<const RectDev*>(c->rect_stage_dev), n_rects, rows, cols, c->field_data[slot]);
  HIP_TRY(c, hipGetLastError());
  // both staging buffers are free again once the scatter has run: the next update (whatever stream the context is on
  // by then) waits for this event before it touches them
  HIP_TRY(c, hipEventRecord(c->rect_stage_done, c->stream));
  // range / stride tables once (the whole map is ~1 MB: rebuilding beats tracking dirty blocks); the partner table,
  // step 2: the new triangles' contributions and a recount of the changed cells
  {
    const int rct = build_tables(c, slot, &pr);
    if (rct != ARTP_OK) return rct;
  }
  // the call returns with the device work queued: launches on the context's OTHER lanes are ordered behind it on the
  // device (map_written), this lane's by stream order
  return publish_map_write(c);
}

int artp_update_layer_rect(artp_ctx* c, int slot, const float* patch, int row0, int col0, int nrows, int ncols) {
  const int rect[4] = {row0, col0, nrows, ncols};
  return artp_update_layer_rects(c, slot, 1, &patch, rect);
}

int artp_check_boxes_dev(artp_ctx* c, int slot, const float box[3], const float* dposes, size_t n,
                         uint8_t* hit, uint8_t* exit_codes) {
  if (!c || !box || slot < 0 || slot > 1 || (n && (!dposes || !hit))) return ARTP_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  if (!c->have_field[slot]) return ARTP_ERR_NO_MAP;
  if (n == 0) return ARTP_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  hipLaunchKernelGGL(check_boxes_kernel<1>, dim3(grid_full(c, n)), dim3(64), lds_full(c), c->stream,
                     c->field[slot], box[0], box[1], box[2], dposes, n, hit, exit_codes, c->caps_full,
                     c->d_error);
  HIP_TRY(c, hipGetLastError());
  return ARTP_OK;
}

int artp_check_boxes(artp_ctx* c, int slot, const float box[3], const float* dposes, size_t n,
                     uint8_t* hit, uint8_t* exit_codes) {
  if (!c || (n && (!dposes || !hit))) return ARTP_ERR_INVALID_ARG;
  if (n == 0) return ARTP_OK;
  // one lock for the whole call: the staging buffers c->tmp[0..1] are shared by every host entry point
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  HIP_TRY(c, hipSetDevice(c->device));
  int rc = ensure_tmp(c, 0, n * 16 * sizeof(float));
  if (rc) return rc;
  rc = ensure_tmp(c, 1, n * 2);
  if (rc) return rc;
  HIP_TRY(c, hipMemcpyAsync(c->tmp[0], dposes, n * 16 * sizeof(float), hipMemcpyHostToDevice, c->stream));
  uint8_t* d_hit = static_cast<uint8_t*>(c->tmp[1]);
  uint8_t* d_ec = d_hit + n;
  rc = artp_check_boxes_dev(c, slot, box, static_cast<const float*>(c->tmp[0]), n, d_hit,
                            exit_codes ? d_ec : nullptr);
  if (rc) return rc;
  HIP_TRY(c, hipMemcpyAsync(hit, d_hit, n, hipMemcpyDeviceToHost, c->stream));
  if (exit_codes) HIP_TRY(c, hipMemcpyAsync(exit_codes, d_ec, n, hipMemcpyDeviceToHost, c->stream));
  return check_error_flag(c);
}

int artp_validate_states_dev(artp_ctx* c, const double* se3, size_t n, uint8_t* valid, int8_t* detail) {
  if (!c || (n && (!se3 || !valid))) return ARTP_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  if (!c->have_field[0] || !c->have_field[1]) return ARTP_ERR_NO_MAP;
  if (n == 0) return ARTP_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  if (!detail && n <= ARTP_FEW_STATES) {
    // tiny batches (the per-state isValid() of the host mirror): ONE launch, one workgroup per state with the
    // five boxes side by side; the labels are the same by construction (both are pinned to the oracle)
    hipLaunchKernelGGL(validate_few_kernel, dim3((unsigned)n), dim3(320), lds_few(c), c->stream, c->field[0],
                       c->field[1], c->geom, c->robot, se3, n, valid, c->caps_full, c->caps_foot_full, c->d_error, 0u);
    HIP_TRY(c, hipGetLastError());
    return ARTP_OK;
  }
  if (detail) {
    // per-box exit codes in the reference's evaluation order: wave-per-state kernel
    hipLaunchKernelGGL(validate_states_kernel<1>, dim3(grid_full(c, n)), dim3(64), lds_full(c), c->stream,
                       c->field[0], c->field[1], c->geom, c->robot, se3, n, valid, detail, c->caps_full,
                       c->d_error, (unsigned long long*)nullptr);
    HIP_TRY(c, hipGetLastError());
    return ARTP_OK;
  }
  return launch_validate_pipeline(c, se3, n, valid);
}

int artp_validate_states(artp_ctx* c, const double* se3, size_t n, uint8_t* valid, int8_t* detail) {
  if (!c || (n && (!se3 || !valid))) return ARTP_ERR_INVALID_ARG;
  if (n == 0) return ARTP_OK;
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  HIP_TRY(c, hipSetDevice(c->device));
  if (!detail && n <= ARTP_FEW_STATES) {
    // latency path: the kernel reads the states from / writes the labels to mapped pinned host memory -- one
    // launch, no copies.  The labels carry a "done" bit the host polls for (a stream synchronise costs more
    // than the kernel); ARTP_NO_POLL=1 or a poll that outlasts 2 ms falls back to hipStreamSynchronize.
    if (!c->have_field[0] || !c->have_field[1]) return ARTP_ERR_NO_MAP;
    // the resident workgroup (no launch at all) for the one-state call OMPL makes; it takes its states one after the other,
    // so from three states on the launch with a workgroup per state is the faster one (measured: 4 states 37 us against 17)
    if (c->svc_enabled && n <= ARTP_SVC_MAX_STATES) return svc_validate(c, se3, n, valid);
    std::memcpy(c->pin_states, se3, n * 7 * sizeof(double));
    for (size_t i = 0; i < n; ++i) c->pin_labels[i] = 0;
    const unsigned tag = 0x80u;
    hipLaunchKernelGGL(validate_few_kernel, dim3((unsigned)n), dim3(320), lds_few(c), c->stream, c->field[0],
                       c->field[1], c->geom, c->robot, (const double*)c->pin_states_dev, n,
                       (volatile uint8_t*)c->pin_labels_dev, c->caps_full, c->caps_foot_full, c->d_error, tag);
    HIP_TRY(c, hipGetLastError());
    bool done = false;
    if (c->poll_labels) {
      const auto t0 = std::chrono::steady_clock::now();
      for (unsigned spin = 0; !done; ++spin) {
        done = true;
        for (size_t i = 0; i < n; ++i) done = done && (c->pin_labels[i] & 0x80u);
        if (!done && (spin & 1023u) == 1023u &&
            std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2))
          break;
      }
    }
    if (!done) HIP_TRY(c, hipStreamSynchronize(c->stream));
    bool overflow = false;
    for (size_t i = 0; i < n; ++i) {
      valid[i] = c->pin_labels[i] & 1u;
      overflow = overflow || (c->pin_labels[i] & 2u);
    }
    if (overflow) {
      c->last_error = "a box window exceeded the LDS tile capacity";
      return ARTP_ERR_CAPACITY;
    }
    return ARTP_OK;
  }
  int rc = ensure_tmp(c, 0, n * 7 * sizeof(double));
  if (rc) return rc;
  rc = ensure_tmp(c, 1, n * 7);
  if (rc) return rc;
  HIP_TRY(c, hipMemcpyAsync(c->tmp[0], se3, n * 7 * sizeof(double), hipMemcpyHostToDevice, c->stream));
  uint8_t* d_valid = static_cast<uint8_t*>(c->tmp[1]);
  int8_t* d_detail = reinterpret_cast<int8_t*>(d_valid + n);
  rc = artp_validate_states_dev(c, static_cast<const double*>(c->tmp[0]), n, d_valid,
                                detail ? d_detail : nullptr);
  if (rc) return rc;
  HIP_TRY(c, hipMemcpyAsync(valid, d_valid, n, hipMemcpyDeviceToHost, c->stream));
  if (detail) HIP_TRY(c, hipMemcpyAsync(detail, d_detail, n * 6, hipMemcpyDeviceToHost, c->stream));
  return check_error_flag(c);
}

static int pack_sampler_tables(artp_ctx* c, int rows, int cols);

// Sampler layer + derived-table storage.  Reused while the grid size stays the same (the in-build re-weighting of
// artp_roadmap_build / _grow re-uploads the distribution every recompute_density_after_n_samples vertices: no
// hipFree / hipMalloc per round, and no free that is unordered against another lane still sampling).  On a size
// change every lane's stream is drained before the old buffers go.
static int ensure_sampler_storage(artp_ctx* c, int rows, int cols) {
  if (c->sampler_buf && c->sampler_pack && c->sampler_rows == rows && c->sampler_cols == cols) return ARTP_OK;
  const size_t e = (size_t)rows * cols;
  const int npiv = (cols + 15) / 16, pitch = npiv * 16, ppitch = (npiv + 3) & ~3;
  const size_t floats = (size_t)rows * pitch + (size_t)rows * ppitch + 8 * e + 64;
  for (int l = 0; l < ARTP_MAX_LANES; ++l)
    if (l != c->cur_lane && c->lanes[l].init) HIP_TRY(c, hipStreamSynchronize(c->lanes[l].stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->have_sampler = false;
  if (c->sampler_buf) HIP_TRY(c, hipFree(c->sampler_buf));
  c->sampler_buf = nullptr;
  if (c->sampler_pack) HIP_TRY(c, hipFree(c->sampler_pack));
  c->sampler_pack = nullptr;
  c->sampler_rows = c->sampler_cols = 0;
  HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->sampler_buf), (6 * e + rows) * sizeof(float)));
  HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->sampler_pack), floats * sizeof(float)));
  c->sampler_rows = rows;
  c->sampler_cols = cols;
  return ARTP_OK;
}

int artp_upload_sampler_layers(artp_ctx* c, const float* cum_prob, const float* cum_prob_rowwise,
                               const float* elevation, const float* normal_x, const float* normal_y,
                               const float* normal_z, const float* plane_fit_std_dev, int rows,
                               int cols, double len_x, double len_y, double pos_x, double pos_y) {
  if (!c || !cum_prob || !cum_prob_rowwise || !elevation || !normal_x || !normal_y || !normal_z ||
      !plane_fit_std_dev || rows < 1 || cols < 1)
    return ARTP_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  HIP_TRY(c, hipSetDevice(c->device));
  const size_t e = (size_t)rows * cols;
  c->map_version.fetch_add(1, std::memory_order_release);
  {
    const int rcs = ensure_sampler_storage(c, rows, cols);
    if (rcs != ARTP_OK) return rcs;
  }
  float* p = c->sampler_buf;
  const float* src[6] = {cum_prob, elevation, normal_x, normal_y, normal_z, plane_fit_std_dev};
  const float** dst[6] = {&c->sampler.cum_prob, &c->sampler.elevation, &c->sampler.normal_x,
                          &c->sampler.normal_y, &c->sampler.normal_z, &c->sampler.plane_fit_std_dev};
  for (int k = 0; k < 6; ++k) {
    HIP_TRY(c, hipMemcpyAsync(p, src[k], e * sizeof(float), hipMemcpyHostToDevice, c->stream));
    *dst[k] = p;
    p += e;
  }
  HIP_TRY(c, hipMemcpyAsync(p, cum_prob_rowwise, rows * sizeof(float), hipMemcpyHostToDevice, c->stream));
  c->sampler.cum_prob_rowwise = p;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  {
    const int rcp = pack_sampler_tables(c, rows, cols);
    if (rcp != ARTP_OK) return rcp;
  }
  if (!c->have_geom) {
    c->geom.len_x = len_x; c->geom.len_y = len_y; c->geom.pos_x = pos_x; c->geom.pos_y = pos_y;
    c->geom.rows = rows; c->geom.cols = cols; c->geom.res = len_x / rows;
    c->have_geom = true;
  }
  c->sampler.from_distribution = c->params.sample_from_distribution;
  c->have_sampler = true;
  return ARTP_OK;
}

// The derived sampler tables (SamplerDev::cum_prob_t / pivots / cells) from the six layers in c->sampler_buf.
static int pack_sampler_tables(artp_ctx* c, int rows, int cols) {
  const size_t e = (size_t)rows * cols;
  const int npiv = (cols + 15) / 16, pitch = npiv * 16, ppitch = (npiv + 3) & ~3;
  const size_t floats = (size_t)rows * pitch + (size_t)rows * ppitch + 8 * e + 64;
  // storage from ensure_sampler_storage (same size for the same rows x cols)
  HIP_TRY(c, hipMemsetAsync(c->sampler_pack, 0, floats * sizeof(float), c->stream));
  float* cells = c->sampler_pack;                       // 8 floats per cell, 32-byte aligned
  float* cdf_t = cells + 8 * e;
  float* piv = cdf_t + (size_t)rows * pitch;
  c->sampler.npiv = npiv;
  c->sampler.pitch = pitch;
  c->sampler.ppitch = ppitch;
  hipLaunchKernelGGL(sampler_pack_kernel, dim3((unsigned)((e + 255) / 256)), dim3(256), 0, c->stream, c->sampler, rows,
                     cols, cdf_t, piv, reinterpret_cast<float4*>(cells));
  HIP_TRY(c, hipGetLastError());
  c->sampler.cells = reinterpret_cast<const float4*>(cells);
  c->sampler.cum_prob_t = cdf_t;
  c->sampler.pivots = piv;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return ARTP_OK;
}

// artp_upload_sampler_layers from layers that already live in HBM (device-to-device copies)
static int upload_sampler_layers_from_device(artp_ctx* c, const float* cum_prob, const float* cum_prob_rowwise,
                                      const float* elevation, const float* normal_x, const float* normal_y,
                                      const float* normal_z, const float* plane_fit_std_dev, int rows, int cols,
                                      double len_x, double len_y, double pos_x, double pos_y) {
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  HIP_TRY(c, hipSetDevice(c->device));
  const size_t e = (size_t)rows * cols;
  c->map_version.fetch_add(1, std::memory_order_release);
  {
    const int rcs = ensure_sampler_storage(c, rows, cols);
    if (rcs != ARTP_OK) return rcs;
  }
  float* p = c->sampler_buf;
  const float* src[6] = {cum_prob, elevation, normal_x, normal_y, normal_z, plane_fit_std_dev};
  const float** dst[6] = {&c->sampler.cum_prob, &c->sampler.elevation, &c->sampler.normal_x,
                          &c->sampler.normal_y, &c->sampler.normal_z, &c->sampler.plane_fit_std_dev};
  for (int k = 0; k < 6; ++k) {
    HIP_TRY(c, hipMemcpyAsync(p, src[k], e * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    *dst[k] = p;
    p += e;
  }
  HIP_TRY(c, hipMemcpyAsync(p, cum_prob_rowwise, rows * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
  c->sampler.cum_prob_rowwise = p;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  {
    const int rcp = pack_sampler_tables(c, rows, cols);
    if (rcp != ARTP_OK) return rcp;
  }
  if (!c->have_geom) {
    c->geom.len_x = len_x; c->geom.len_y = len_y; c->geom.pos_x = pos_x; c->geom.pos_y = pos_y;
    c->geom.rows = rows; c->geom.cols = cols; c->geom.res = len_x / rows;
    c->have_geom = true;
  }
  c->sampler.from_distribution = c->params.sample_from_distribution;
  c->have_sampler = true;
  return ARTP_OK;
}

// min / max of the finite samples of a device layer (host result); false if there is none
static int finite_min_max_dev(artp_ctx* c, const float* d_layer, size_t n, float* lo, float* hi, bool* any) {
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  HIP_TRY(c, hipSetDevice(c->device));
  int init[2] = {0x7fffffff, (int)0x80000000};
  int* d_keys = reinterpret_cast<int*>(c->d_count);
  HIP_TRY(c, hipMemcpyAsync(d_keys, init, sizeof(init), hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(finite_min_max_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, (size_t)c->n_cus)), dim3(256), 0,
                     c->stream, d_layer, (int)n, d_keys);
  HIP_TRY(c, hipGetLastError());
  int keys[2];
  HIP_TRY(c, hipMemcpyAsync(keys, d_keys, sizeof(keys), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  *any = keys[0] <= keys[1];
  auto unkey = [](int k) {
    const int i = k >= 0 ? k : k ^ 0x7fffffff;
    float v;
    std::memcpy(&v, &i, 4);
    return v;
  };
  *lo = unkey(keys[0]);
  *hi = unkey(keys[1]);
  return ARTP_OK;
}

// recs != nullptr: also emit the PoseRecs of the validity pipeline (fused sample + validate)
static int launch_sampler(artp_ctx* c, uint64_t seed, uint64_t first_index, size_t n, double* se3_out, PoseRec* recs) {
  const size_t blocks = (n + 255) / 256;  // one wavefront per 64 states
  if (blocks > 0x7fffffffull) {
    c->last_error = "batch too large for one sampler launch";
    return ARTP_ERR_INVALID_ARG;
  }
  if (c->sampler.from_distribution)
    hipLaunchKernelGGL(sample_states_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, c->stream, c->sampler,
                       c->geom, c->robot, seed, first_index, n, se3_out, c->field[0], recs);
  else
    hipLaunchKernelGGL(sample_states_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, c->stream, c->sampler,
                       c->geom, c->robot, seed, first_index, n, se3_out, c->field[0], recs);
  HIP_TRY(c, hipGetLastError());
  return ARTP_OK;
}

int artp_sample_states_dev(artp_ctx* c, uint64_t seed, uint64_t first_index, size_t n, double* se3_out) {
  if (!c || (n && !se3_out)) return ARTP_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  if (!c->have_sampler) return ARTP_ERR_NO_MAP;
  if (n == 0) return ARTP_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  return launch_sampler(c, seed, first_index, n, se3_out, nullptr);
}

int artp_sample_states(artp_ctx* c, uint64_t seed, uint64_t first_index, size_t n, double* se3_out) {
  if (!c || (n && !se3_out)) return ARTP_ERR_INVALID_ARG;
  if (n == 0) return ARTP_OK;
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  HIP_TRY(c, hipSetDevice(c->device));
  int rc = ensure_tmp(c, 0, n * 7 * sizeof(double));
  if (rc) return rc;
  rc = artp_sample_states_dev(c, seed, first_index, n, static_cast<double*>(c->tmp[0]));
  if (rc) return rc;
  HIP_TRY(c, hipMemcpyAsync(se3_out, c->tmp[0], n * 7 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return ARTP_OK;
}

int artp_sample_and_validate_dev(artp_ctx* c, uint64_t seed, uint64_t first_index, size_t n,
                                 double* se3_out, uint8_t* valid_out, size_t* n_valid) {
  if (!c || (n && (!se3_out || !valid_out))) return ARTP_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  if (!c->have_sampler || !c->have_field[0] || !c->have_field[1]) return ARTP_ERR_NO_MAP;
  if (n == 0) {
    if (n_valid) *n_valid = 0;
    return ARTP_OK;
  }
  if (n >= (1ull << 32)) {
    c->last_error = "batch too large (state index is 32 bit)";
    return ARTP_ERR_INVALID_ARG;
  }
  HIP_TRY(c, hipSetDevice(c->device));
  // fused: the sampler hands the per-state part of the validity check (float pose, orthogonalised rotation in the
  // field frame) to the pipeline while the state is still in its registers
  int rc = ensure_recs(c, n);
  if (rc) return rc;
  if (n <= ARTP_FEW_STATES) {  // the few-state kernel reads the states themselves
    rc = launch_sampler(c, seed, first_index, n, se3_out, nullptr);
    if (rc) return rc;
    rc = artp_validate_states_dev(c, se3_out, n, valid_out, nullptr);
    if (rc) return rc;
  } else {
    rc = launch_sampler(c, seed, first_index, n, se3_out, static_cast<PoseRec*>(c->tmp[7]));
    if (rc) return rc;
  }
  if (n > ARTP_FEW_STATES) {
    const int rcv = launch_validate_pipeline(c, se3_out, n, valid_out, true);
    if (rcv) return rcv;
  }
  if (n_valid) {
    HIP_TRY(c, hipMemsetAsync(c->d_count, 0, sizeof(unsigned long long), c->stream));
    size_t blocks = (n / 16 + 255) / 256 + 1;
    if (blocks > (size_t)c->n_cus * 4) blocks = (size_t)c->n_cus * 4;
    hipLaunchKernelGGL(count_valid_kernel, dim3((unsigned)blocks), dim3(256), 0, c->stream,
                       (const uint8_t*)valid_out, n, c->d_count);
    HIP_TRY(c, hipGetLastError());
  }
  if (n_valid) {
    unsigned long long cnt = 0;
    HIP_TRY(c, hipMemcpyAsync(&cnt, c->d_count, sizeof(cnt), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    *n_valid = (size_t)cnt;
  }
  return ARTP_OK;
}

// Host-buffer form of the fused rejection-sampling step: states AND labels come back, so a caller that hands the
// states out one at a time (SE3FromSE2Sampler::sampleUniform of the host mirror) already knows their labels.
int artp_sample_and_validate(artp_ctx* c, uint64_t seed, uint64_t first_index, size_t n, double* se3_out,
                             uint8_t* valid_out, uint64_t* map_version) {
  if (!c || (n && (!se3_out || !valid_out))) return ARTP_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  if (map_version) *map_version = c->map_version.load(std::memory_order_acquire);  // fixed while `mu` is held
  if (n == 0) return ARTP_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  int rc = ensure_tmp(c, 0, n * 7 * sizeof(double));
  if (rc) return rc;
  rc = ensure_tmp(c, 1, n);
  if (rc) return rc;
  rc = artp_sample_and_validate_dev(c, seed, first_index, n, static_cast<double*>(c->tmp[0]),
                                    static_cast<uint8_t*>(c->tmp[1]), nullptr);
  if (rc) return rc;
  HIP_TRY(c, hipMemcpyAsync(se3_out, c->tmp[0], n * 7 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipMemcpyAsync(valid_out, c->tmp[1], n, hipMemcpyDeviceToHost, c->stream));
  return check_error_flag(c);
}

int artp_set_r3_extent(artp_ctx* c, double max_extent) {
  if (!c || !(max_extent >= 0.0)) return ARTP_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  c->r3_extent_override = max_extent;
  return ARTP_OK;
}

int artp_set_z_bounds(artp_ctx* c, double z_low, double z_high) {
  if (!c) return ARTP_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  c->z_low = z_low;
  c->z_high = z_high;
  c->have_z = true;
  return ARTP_OK;
}

// mode 0: checkMotion, mode 1: the 0.5 m interpolation rule.  last_t / last_state (mode 0 only, device, may be
// NULL): the lastValid pair of checkMotion's second overload.
static int run_edges_dev(artp_ctx* c, int mode, const double* s1, const double* s2, size_t n,
                         uint8_t* valid, uint32_t* aux_out, double* last_t = nullptr, double* last_state = nullptr) {
  if (!c || (n && (!s1 || !s2 || !valid))) return ARTP_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  if (!c->have_field[0] || !c->have_field[1]) return ARTP_ERR_NO_MAP;
  if (mode == 0 && !c->have_z) {
    c->last_error = "artp_set_z_bounds must be called before artp_check_motions";
    return ARTP_ERR_NO_MAP;
  }
  if (n == 0) return ARTP_OK;
  if (n + 1 > (size_t)0x7fffffff) {
    c->last_error = "edge batch too large (the scan is 32 bit)";
    return ARTP_ERR_CAPACITY;
  }
  HIP_TRY(c, hipSetDevice(c->device));
  // tmp[2]: counts (n+1) | offsets (n+1) | aux (n) | first_bad (n) | total64, overflow flag, total of pass 1, pad |
  //         slerp constants (n x 24 B) | counts of the coarse pass (n+1) | their offsets (n+1)
  int rc = ensure_tmp(c, 2, (6 * n + 4) * sizeof(uint32_t) + 64 + n * sizeof(SlerpEdge));
  if (rc) return rc;
  uint32_t* counts = static_cast<uint32_t*>(c->tmp[2]);
  uint32_t* offsets = counts + (n + 1);
  uint32_t* aux = aux_out ? aux_out : offsets + (n + 1);
  uint32_t* first_bad = offsets + (n + 1) + n;
  unsigned long long* d_total = reinterpret_cast<unsigned long long*>(
      (reinterpret_cast<uintptr_t>(first_bad + n) + 7) & ~(uintptr_t)7);
  int* d_overflow = reinterpret_cast<int*>(d_total + 1);
  unsigned long long* d_total1 = d_total + 2;
  SlerpEdge* d_slerp = reinterpret_cast<SlerpEdge*>(d_total + 4);  // 8-byte aligned like d_total
  uint32_t* counts1 = reinterpret_cast<uint32_t*>(d_slerp + n);
  uint32_t* offsets1 = counts1 + (n + 1);
  // checkMotion's first overload in two passes (kernels.h ARTP_COARSE_STRIDE): worth its second pipeline pass for real batches
  const bool want_last_ = (mode == 0) && last_t;
  const bool two_pass_possible = mode == 0 && !want_last_ && n >= 4096 && c->edge_two_pass;
  HIP_TRY(c, hipMemsetAsync(counts + n, 0, sizeof(uint32_t), c->stream));
  HIP_TRY(c, hipMemsetAsync(d_total, 0, 32, c->stream));
  size_t blocks = (n + 255) / 256;
  if (blocks > (size_t)c->n_cus * 4) blocks = (size_t)c->n_cus * 4;  // grid-stride; one atomic per workgroup on the total
  if (two_pass_possible) HIP_TRY(c, hipMemsetAsync(counts1 + n, 0, sizeof(uint32_t), c->stream));
  hipLaunchKernelGGL(motion_plan_kernel, dim3((unsigned)blocks), dim3(256), 0, c->stream, c->geom,
                     c->z_high - c->z_low, mode, s1, s2, n, counts, aux, valid, d_overflow, d_total, d_slerp,
                     two_pass_possible ? counts1 : (uint32_t*)nullptr, two_pass_possible ? d_total1 : (unsigned long long*)nullptr,
                     (uint32_t)c->edge_coarse_stride, c->r3_extent_override);
  HIP_TRY(c, hipGetLastError());
  size_t need = 0;
  HIP_TRY(c, hipcub::DeviceScan::ExclusiveSum(nullptr, need, counts, offsets, (int)(n + 1), c->stream));
  if (c->cub_cap < need) {
    if (c->cub_tmp) HIP_TRY(c, hipFree(c->cub_tmp));
    c->cub_tmp = nullptr;
    HIP_TRY(c, hipMalloc(&c->cub_tmp, need + 256));
    c->cub_cap = need + 256;
  }
  size_t cap = c->cub_cap;
  HIP_TRY(c, hipcub::DeviceScan::ExclusiveSum(c->cub_tmp, cap, counts, offsets, (int)(n + 1), c->stream));
  if (two_pass_possible) {
    cap = c->cub_cap;
    HIP_TRY(c, hipcub::DeviceScan::ExclusiveSum(c->cub_tmp, cap, counts1, offsets1, (int)(n + 1), c->stream));
  }
  // expand every interior state into one state batch, validate it with the standard pipeline, then
  // fold the labels back onto the edges
  struct { unsigned long long total; int overflow; int pad; unsigned long long total1; unsigned long long pad2; } plan{0, 0, 0, 0, 0};
  HIP_TRY(c, hipMemcpyAsync(&plan, d_total, 32, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (plan.overflow) {
    c->last_error = "an edge has non-finite states or needs more than 2^22 interpolation states";
    return ARTP_ERR_INVALID_ARG;
  }
  if (plan.total >= (1ull << 32)) {
    c->last_error = "edge batch expands to 2^32 or more states: split it";
    return ARTP_ERR_CAPACITY;
  }
  const uint32_t total = (uint32_t)plan.total;
  const bool want_last = want_last_;
  if (want_last) HIP_TRY(c, hipMemsetAsync(first_bad, 0xff, n * sizeof(uint32_t), c->stream));
  size_t eb = ((size_t)total + 255) / 256;
  if (eb > (size_t)c->n_cus * 32) eb = (size_t)c->n_cus * 32;
  if (two_pass_possible && plan.total >= 24ull * n) {
    // ---- two passes: s2 + every 8th interior state of every edge, then the rest of the edges still alive ----
    auto run_pass = [&](int pass, const uint32_t* offs, uint32_t tot) -> int {
      if (tot == 0) return ARTP_OK;
      size_t pb = ((size_t)tot + 255) / 256;
      if (pb > (size_t)c->n_cus * 32) pb = (size_t)c->n_cus * 32;
      int r = ensure_tmp(c, 3, (size_t)tot * (sizeof(uint32_t) + 1) + 64);
      if (r) return r;
      r = ensure_recs(c, tot);
      if (r) return r;
      uint32_t* edge_of = static_cast<uint32_t*>(c->tmp[3]);
      uint8_t* ex_valid = reinterpret_cast<uint8_t*>(edge_of + tot);
      hipLaunchKernelGGL(expand_edges_recs_kernel, dim3((unsigned)pb), dim3(256), 0, c->stream, c->field[0], mode, s1, s2, n,
                         offs, (const uint32_t*)aux, (const SlerpEdge*)d_slerp, static_cast<PoseRec*>(c->tmp[7]), edge_of,
                         pass, (uint32_t)c->edge_coarse_stride);
      HIP_TRY(c, hipGetLastError());
      r = launch_validate_pipeline(c, nullptr, tot, ex_valid, true);
      if (r) return r;
      hipLaunchKernelGGL(reduce_edges_kernel, dim3((unsigned)pb), dim3(256), 0, c->stream, (const uint8_t*)ex_valid,
                         (const uint32_t*)edge_of, offs, n, valid);
      HIP_TRY(c, hipGetLastError());
      return ARTP_OK;
    };
    rc = run_pass(1, offsets1, (uint32_t)plan.total1);
    if (rc) return rc;
    // pass 2: counts of what is left, for the edges still valid (reuses the pass-1 arrays)
    HIP_TRY(c, hipMemsetAsync(d_total1, 0, sizeof(unsigned long long), c->stream));
    hipLaunchKernelGGL(coarse_pass2_counts_kernel, dim3((unsigned)blocks), dim3(256), 0, c->stream, (const uint32_t*)aux,
                       (const uint8_t*)valid, n, (uint32_t)c->edge_coarse_stride, counts1, d_total1);
    HIP_TRY(c, hipGetLastError());
    cap = c->cub_cap;
    HIP_TRY(c, hipcub::DeviceScan::ExclusiveSum(c->cub_tmp, cap, counts1, offsets1, (int)(n + 1), c->stream));
    unsigned long long total2 = 0;
    HIP_TRY(c, hipMemcpyAsync(&total2, d_total1, sizeof(total2), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return run_pass(2, offsets1, (uint32_t)total2);
  }
  if (total != 0) {
    // tmp[3]: edge_of (total u32) | ex_valid (total bytes); the interior states themselves only exist as PoseRecs (tmp[7])
    rc = ensure_tmp(c, 3, (size_t)total * (sizeof(uint32_t) + 1) + 64);
    if (rc) return rc;
    rc = ensure_recs(c, total);
    if (rc) return rc;
    uint32_t* edge_of = static_cast<uint32_t*>(c->tmp[3]);
    uint8_t* ex_valid = reinterpret_cast<uint8_t*>(edge_of + total);
    hipLaunchKernelGGL(expand_edges_recs_kernel, dim3((unsigned)eb), dim3(256), 0, c->stream, c->field[0], mode, s1, s2, n,
                       (const uint32_t*)offsets, (const uint32_t*)aux, (const SlerpEdge*)d_slerp,
                       static_cast<PoseRec*>(c->tmp[7]), edge_of);
    HIP_TRY(c, hipGetLastError());
    rc = launch_validate_pipeline(c, nullptr, total, ex_valid, true);
    if (rc) return rc;
    hipLaunchKernelGGL(reduce_edges_kernel, dim3((unsigned)eb), dim3(256), 0, c->stream,
                       (const uint8_t*)ex_valid, (const uint32_t*)edge_of, (const uint32_t*)offsets, n, valid);
    if (want_last)
      hipLaunchKernelGGL(reduce_edges_first_bad_kernel, dim3((unsigned)eb), dim3(256), 0, c->stream,
                         (const uint8_t*)ex_valid, (const uint32_t*)edge_of, (const uint32_t*)offsets,
                         (const uint32_t*)aux, n, first_bad);
    HIP_TRY(c, hipGetLastError());
  }
  if (want_last) {
    hipLaunchKernelGGL(last_valid_kernel, dim3((unsigned)blocks), dim3(256), 0, c->stream, s1, s2, n,
                       (const uint32_t*)aux, (const uint32_t*)first_bad, last_t, last_state);
    HIP_TRY(c, hipGetLastError());
  }
  return ARTP_OK;
}

// Estimate of the wave-tasks of a small edge batch in plain host arithmetic: the total chooses the path (latency kernel or
// batch pipeline), the largest per-edge count sizes the latency kernel's grid.  Estimates only -- the kernels form the
// exact counts themselves, and an edge with more tasks than workgroups just loops.  Non-finite input -> huge, i.e. the
// batch path and its error.
static double few_edges_task_estimate(const artp_ctx* c, int mode, const double* s1, const double* s2, size_t n,
                                      double* max_per_edge) {
  double total = 0.0, mx = 1.0;
  const double ex = 2.0 * c->geom.len_x, ey = 2.0 * c->geom.len_y, ez = c->z_high - c->z_low;
  const double seg = (c->r3_extent_override > 0.0 ? c->r3_extent_override : std::sqrt(ex * ex + ey * ey + ez * ez)) * 0.01;
  for (size_t e = 0; e < n; ++e) {
    const double* a = s1 + 7 * e;
    const double* b = s2 + 7 * e;
    const double dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    double t;
    if (mode == 0) {
      const double r3 = std::sqrt(dx * dx + dy * dy + dz * dz) / seg;
      double dq = std::fabs(a[3] * b[3] + a[4] * b[4] + a[5] * b[5] + a[6] * b[6]);
      const double so3 = (dq < 1.0 ? std::acos(dq) : 0.0) / (0.005 * 3.14159265358979323846);
      t = (r3 > so3 ? r3 : so3) + 2.0;
    } else {
      t = std::sqrt(dx * dx + dy * dy) / 0.5 + 1.0;
    }
    if (!(t < 1e9)) return 1e30;
    total += t;
    mx = t > mx ? t : mx;
  }
  *max_per_edge = mx;
  return total;
}

// <= ARTP_FEW_EDGES edges in ONE launch (kernels.h check_motions_few_kernel).  s1 / s2: host pointers (host_io: staged
// through the mapped block, results read back from it) or device pointers (results to the caller's device arrays);
// either way the verdict's status bytes come back through mapped memory and the call returns when they are there.
static int run_edges_few(artp_ctx* c, int mode, bool host_io, const double* s1, const double* s2, size_t n, uint8_t* valid,
                         uint32_t* aux_out, double* last_t, double* last_state, double max_tasks_per_edge) {
  char* hb = c->pin_edges;
  char* db = c->pin_edges_dev;
  volatile uint8_t* status = reinterpret_cast<volatile uint8_t*>(hb + FEW_EDGE_STATUS);
  if (host_io) {
    std::memcpy(c->pin_edges_in + FEW_EDGE_S1, s1, n * 7 * sizeof(double));
    std::memcpy(c->pin_edges_in + FEW_EDGE_S2, s2, n * 7 * sizeof(double));
  }
  for (size_t i = 0; i < n; ++i) status[i] = 0;
  // workgroups per edge: one per task of the longest edge (estimated; <= ~100 for edges shorter than the map), ~2048
  // workgroups per launch at most
  unsigned chunks = (unsigned)(2048 / n), want = (unsigned)(max_tasks_per_edge < 128.0 ? max_tasks_per_edge : 128.0) + 1u;
  chunks = chunks > 128 ? 128 : (chunks < 16 ? 16 : chunks);
  chunks = chunks > want ? want : chunks;
  const unsigned tag = 0x80u;
  const bool want_last = mode == 0 && last_t;
  hipLaunchKernelGGL(check_motions_few_kernel, dim3(chunks, (unsigned)n), dim3(320), lds_few(c), c->stream, c->field[0],
                     c->field[1], c->geom, c->robot, c->z_high - c->z_low, c->r3_extent_override, mode,
                     host_io ? reinterpret_cast<const double*>(c->pin_edges_in_dev + FEW_EDGE_S1) : s1,
                     host_io ? reinterpret_cast<const double*>(c->pin_edges_in_dev + FEW_EDGE_S2) : s2, (uint32_t)n, c->d_few_sync,
                     reinterpret_cast<volatile uint8_t*>(db + FEW_EDGE_STATUS), host_io ? (uint8_t*)nullptr : valid,
                     !aux_out ? (uint32_t*)nullptr : host_io ? reinterpret_cast<uint32_t*>(db + FEW_EDGE_AUX) : aux_out,
                     !want_last ? (double*)nullptr : host_io ? reinterpret_cast<double*>(db + FEW_EDGE_LAST_T) : last_t,
                     !(want_last && last_state) ? (double*)nullptr
                         : host_io ? reinterpret_cast<double*>(db + FEW_EDGE_LAST_STATE) : last_state,
                     c->caps_full, c->caps_foot_full, tag);
  HIP_TRY(c, hipGetLastError());
  bool done = false;
  if (c->poll_labels) {
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spin = 0; !done; ++spin) {
      done = true;
      for (size_t i = 0; i < n; ++i) done = done && (status[i] & 0x80u);
      if (!done && (spin & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20))
        break;
    }
  }
  if (!done) HIP_TRY(c, hipStreamSynchronize(c->stream));
  std::atomic_thread_fence(std::memory_order_acquire);
  unsigned flags = 0;
  for (size_t i = 0; i < n; ++i) flags |= status[i];
  if (flags & 4u) {
    c->last_error = "an edge has non-finite states or needs more than 2^22 interpolation states";
    return ARTP_ERR_INVALID_ARG;
  }
  if (flags & 2u) {
    c->last_error = "a box window exceeded the LDS tile capacity";
    return ARTP_ERR_CAPACITY;
  }
  if (host_io) {
    for (size_t i = 0; i < n; ++i) valid[i] = status[i] & 1u;
    if (aux_out) std::memcpy(aux_out, hb + FEW_EDGE_AUX, n * sizeof(uint32_t));
    if (want_last) std::memcpy(last_t, hb + FEW_EDGE_LAST_T, n * sizeof(double));
    if (want_last && last_state) std::memcpy(last_state, hb + FEW_EDGE_LAST_STATE, n * 7 * sizeof(double));
  }
  return ARTP_OK;
}

// ---- resident pool for calls of one or two edges (kernels.h check_motions_pool_kernel) --------------------------------
static bool pool_any_exited(const artp_ctx* c) {
  for (unsigned w = 0; w < c->pool_wgs; ++w)
    if (c->pool_resp->exited[w]) return true;
  return false;
}

// Host stores into the request block have left the core's write-combining buffers (the block may be device memory behind
// the PCIe BAR) and are ordered against the stores that follow.
static inline void pool_store_fence() {
#if defined(__x86_64__)
  __builtin_ia32_sfence();
#endif
  std::atomic_thread_fence(std::memory_order_seq_cst);
}

// The pool needs a device whose memory the host can write (large BAR); otherwise edge calls keep their launch per call.
static bool pool_available(artp_ctx* c) {
  if (c->pool_bar < 0) {
    int large_bar = 0;
    void* p = nullptr;
    if (hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, c->device) == hipSuccess && large_bar &&
        hipExtMallocWithFlags(&p, sizeof(EdgeMailbox), hipDeviceMallocFinegrained) == hipSuccess) {
      // the request block: DEVICE memory the host writes through the PCIe BAR (kernels.h), cleared the same way
      c->pool_mb = static_cast<EdgeMailbox*>(p);
      for (size_t i = 0; i < sizeof(EdgeMailbox) / 8; ++i) reinterpret_cast<volatile uint64_t*>(p)[i] = 0;
      c->pool_bar = 1;
    } else {
      (void)hipGetLastError();
      c->pool_bar = 0;
    }
  }
  return c->pool_bar == 1;
}

static void pool_stop(artp_ctx* c) {
  if (!c->pool_mb || !c->pool_launched) return;
  // (line 0 is never READ by the host -- a read across the BAR is a microsecond: the request number is c->pool_seq)
  reinterpret_cast<volatile uint64_t*>(&c->pool_mb->line[0])[0] = ((uint64_t)0x100u << 32) | c->pool_seq;
  pool_store_fence();
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    bool all = true;
    for (unsigned w = 0; w < c->pool_wgs; ++w) all = all && c->pool_resp->exited[w];
    if (all || std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) break;
  }
  (void)hipStreamSynchronize(c->pool_stream);   // every workgroup has returned (at the latest after the lifetime limit)
  c->pool_launched = false;
}

static int pool_start(artp_ctx* c) {
  if (!c->pool_resp) {
    void *r = nullptr, *rd = nullptr;
    HIP_TRY(c, hipHostMalloc(&r, sizeof(PoolResponse), hipHostMallocMapped));
    c->pool_resp = static_cast<PoolResponse*>(r);
    HIP_TRY(c, hipHostGetDevicePointer(&rd, r, 0));
    c->pool_resp_dev = static_cast<PoolResponse*>(rd);
    std::memset(r, 0, sizeof(PoolResponse));
    HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->pool_ctl), sizeof(PoolCtl)));
    HIP_TRY(c, hipStreamCreateWithFlags(&c->pool_stream, hipStreamNonBlocking));
    HIP_TRY(c, hipEventCreateWithFlags(&c->pool_after_map, hipEventDisableTiming));
    HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(check_motions_pool_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_few(c)));
  }
  if (c->pool_seq >= 0xfffffff0u) {   // the request numbers wrap: every tag back to "never"
    reinterpret_cast<volatile uint64_t*>(&c->pool_mb->line[0])[0] = 0;
    std::memset(c->pool_resp, 0, sizeof(PoolResponse));
    c->pool_seq = 0;
  }
  HIP_TRY(c, hipEventRecord(c->pool_after_map, c->stream));          // behind the map writes already enqueued
  HIP_TRY(c, hipStreamWaitEvent(c->pool_stream, c->pool_after_map, 0));
  HIP_TRY(c, hipMemsetAsync(c->pool_ctl, 0, sizeof(PoolCtl), c->pool_stream));
  reinterpret_cast<volatile uint64_t*>(&c->pool_mb->line[0])[0] = c->pool_seq;   // word 0: nothing to do, do not leave
  for (unsigned w = 0; w < ARTP_POOL_MAX_WGS; ++w) c->pool_resp->exited[w] = 0;
  pool_store_fence();
  hipLaunchKernelGGL(check_motions_pool_kernel, dim3(c->pool_wgs), dim3(320), lds_few(c), c->pool_stream, c->field[0],
                     c->field[1], c->geom, c->robot, c->pool_mb, c->pool_resp_dev, c->pool_ctl, c->pool_seq, c->caps_full,
                     c->caps_foot_full);
  HIP_TRY(c, hipGetLastError());
  c->pool_launched = true;
  c->pool_map_version = c->map_version.load(std::memory_order_acquire);
  ++c->pool_launches;
  return ARTP_OK;
}

// One or two edges (host pointers) through the resident pool; results as run_edges_few's.  The z bounds and the R3 extent
// override travel with every request (setters change them without a map write).  The host reduces the workgroups' slots:
// smallest failing order over the pool = DiscreteMotionValidator's first invalid state.
static int run_edges_pool(artp_ctx* c, int mode, const double* s1, const double* s2, size_t n, uint8_t* valid,
                          uint32_t* aux_out, double* last_t, double* last_state) {
  const bool want_last = mode == 0 && last_t;
  for (int attempt = 0; attempt < 3; ++attempt) {
    const bool stale = c->pool_launched && c->pool_map_version != c->map_version.load(std::memory_order_acquire);
    if (stale || (c->pool_launched && (c->pool_seq >= 0xfffffff0u || pool_any_exited(c)))) pool_stop(c);
    if (!c->pool_launched) {
      const int rc = pool_start(c);
      if (rc) return rc;
    }
    EdgeMailbox* mb = c->pool_mb;
    const unsigned P = c->pool_wgs;
    // the block as the device will see it (c->pool_shadow: the host never reads across the BAR): payloads, header word, checksum
    const uint32_t seq = ++c->pool_seq;
    const uint32_t word = (uint32_t)n | (mode ? 0x200u : 0u) | (want_last ? 0x400u : 0u);
    uint64_t (*sh)[7] = c->pool_shadow;
    for (size_t i = 0; i < n; ++i) {
      std::memcpy(sh[i ? 3 : 0], s1 + 7 * i, 7 * sizeof(double));
      std::memcpy(sh[i ? 4 : 1], s2 + 7 * i, 7 * sizeof(double));
    }
    const double zr[2] = {c->z_high - c->z_low, c->r3_extent_override};
    std::memcpy(sh[2], zr, sizeof(zr));
    sh[2][2] = ((uint64_t)seq << 32) | word;
    uint64_t sum = pool_mix((uint64_t)seq, 64u);
    for (unsigned l = 0; l < 5; ++l)
      for (unsigned i = 0; i < 7; ++i)
        if (8 * l + 1 + i != ARTP_POOL_SUM_LANE) sum ^= pool_mix(sh[l][i], 8 * l + 1 + i);
    sh[2][6] = sum;
    for (unsigned l = 0; l < (n > 1 ? 5u : 3u); ++l) std::memcpy(mb->line[l].v, sh[l], 7 * sizeof(uint64_t));
    // the payload has left the write-combining buffers before the doorbell (line 0's number) is rung; what the device accepts
    // is decided by the checksum, not by this order
    pool_store_fence();
    reinterpret_cast<volatile uint64_t*>(&mb->line[0])[0] = seq;
    pool_store_fence();
    const auto t0 = std::chrono::steady_clock::now();
    // per edge: the slot of the workgroup task 0 fell to (it carries the edge's task count), then the slots of the
    // min(tasks, P) workgroups that had a task; the next edge's tasks continue the numbering
    bool gone = false, done = false;
    size_t e = 0;
    unsigned lead = 0, idx = 0, cnt = 1, flags = 0;
    uint64_t base = 0;
    uint32_t fbad = 0xffffffffu, who = 0, aux = 0;
    for (unsigned spin = 0; !done; ++spin) {
      for (;;) {
        const unsigned w = (lead + idx) % P;
        const volatile PoolSlot& sl = c->pool_resp->slot[e][w];
        if (sl.tag != seq) break;
        std::atomic_thread_fence(std::memory_order_acquire);
        // the slot as a whole: taken when its check word verifies (with the lastValid state it announces), else not there yet
        const uint32_t s_bad = sl.first_bad, s_flags = sl.flags, s_aux = sl.aux, s_check = sl.check;
        uint64_t state_xor = 0;
        if (want_last && s_bad != 0xffffffffu)
          for (int i = 0; i < 7; ++i) {
            uint64_t u;
            const double v = c->pool_resp->last_state[e][w][i];
            std::memcpy(&u, &v, 8);
            state_xor ^= u;
          }
        if (pool_slot_check(seq, s_bad, s_flags, s_aux, state_xor) != s_check) break;
        if (idx == 0) {
          const uint32_t tasks = s_flags >> 3;
          cnt = tasks < P ? (tasks ? tasks : 1u) : P;
          base += tasks;
          aux = s_aux;
        }
        flags |= s_flags & 7u;
        if (s_bad < fbad) {
          fbad = s_bad;
          who = w;
        }
        if (++idx < cnt) continue;
        // the edge is complete
        valid[e] = fbad == 0xffffffffu && !(flags & 6u);
        if (aux_out) aux_out[e] = aux;
        if (want_last) {   // check_motions_few_kernel's finalization rule
          if (fbad == 0xffffffffu) {
            last_t[e] = 1.0;
            if (last_state) std::memcpy(last_state + 7 * e, s2 + 7 * e, 7 * sizeof(double));
          } else {
            const int nd = (int)aux;
            last_t[e] = nd > 0 ? (double)fbad / (double)nd : (double)(nd - 1) / (double)nd;
            if (last_state)
              std::memcpy(last_state + 7 * e, const_cast<const double*>(c->pool_resp->last_state[e][who]), 7 * sizeof(double));
          }
        }
        if (++e == n) {
          done = true;
          break;
        }
        lead = (unsigned)(base % P);
        idx = 0;
        cnt = 1;
        fbad = 0xffffffffu;
      }
      if (!done && (spin & 255u) == 255u) {
        // Has the workgroup whose slot is awaited left (idle limit reached as the request went out)?  Only then is the
        // request posted again: workgroups WITHOUT a task of it may well leave while the others are still working on a slow
        // one (a 200 us request on a map full of unknown cells), and those others still answer.  Its slot is written in
        // front of its exit flag, so a flag without a verifying slot means it never saw the request.
        const unsigned w = (lead + idx) % P;
        if (c->pool_resp->exited[w]) {
          std::atomic_thread_fence(std::memory_order_acquire);
          if (c->pool_resp->slot[e][w].tag != seq) { gone = true; break; }
        }
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(50)) break;
      }
    }
    if (done) {
      ++c->pool_requests;
      if (flags & 4u) {
        c->last_error = "an edge has non-finite states or needs more than 2^22 interpolation states";
        return ARTP_ERR_INVALID_ARG;
      }
      if (flags & 2u) {
        c->last_error = "a box window exceeded the LDS tile capacity";
        return ARTP_ERR_CAPACITY;
      }
      return ARTP_OK;
    }
    if (gone) {   // post it again, under a NEW number, to a fresh pool: whatever the old one still wrote carries the old one
      pool_stop(c);
      continue;
    }
    c->last_error = "the resident edge pool did not answer within 50 ms";
    pool_stop(c);
    return ARTP_ERR_TIMEOUT;
  }
  c->last_error = "the resident edge pool could not be (re)started";
  return ARTP_ERR_TIMEOUT;
}

static int run_edges_host(artp_ctx* c, int mode, const double* s1, const double* s2, size_t n,
                          uint8_t* valid, uint32_t* aux_out, double* last_t = nullptr, double* last_state = nullptr) {
  if (!c || (n && (!s1 || !s2 || !valid))) return ARTP_ERR_INVALID_ARG;
  if (n == 0) return ARTP_OK;
  // one lock for the whole call: the staging buffers c->tmp[0..1] are shared by every host entry point
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  HIP_TRY(c, hipSetDevice(c->device));
  double few_max = 0.0;
  if (n <= ARTP_FEW_EDGES && c->few_edges && c->have_field[0] && c->have_field[1] && (mode != 0 || c->have_z) &&
      few_edges_task_estimate(c, mode, s1, s2, n, &few_max) <= 65536.0) {
    // (at most eight rounds per workgroup: one that has run out of tasks must not reach its idle limit while others work)
    if (c->svc_enabled && n <= ARTP_POOL_MAX_EDGES && few_max * n <= 8.0 * c->pool_wgs && pool_available(c))
      return run_edges_pool(c, mode, s1, s2, n, valid, aux_out, last_t, last_state);
    return run_edges_few(c, mode, true, s1, s2, n, valid, aux_out, last_t, last_state, few_max);
  }
  int rc = ensure_tmp(c, 0, 2 * n * 7 * sizeof(double));
  if (rc) return rc;
  // tmp[1]: last_t (n doubles) | last_state (7n doubles) | aux (n u32) | valid (n)
  rc = ensure_tmp(c, 1, n * 8 * sizeof(double) + n * 4 + n + 16);
  if (rc) return rc;
  HIP_TRY(c, hipMemcpyAsync(c->tmp[0], s1, n * 7 * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync(static_cast<double*>(c->tmp[0]) + 7 * n, s2, n * 7 * sizeof(double),
                            hipMemcpyHostToDevice, c->stream));
  double* d_last_t = static_cast<double*>(c->tmp[1]);
  double* d_last_state = d_last_t + n;
  uint32_t* d_aux = reinterpret_cast<uint32_t*>(d_last_state + 7 * n);
  uint8_t* d_valid = reinterpret_cast<uint8_t*>(d_aux + n);
  rc = run_edges_dev(c, mode, static_cast<const double*>(c->tmp[0]),
                     static_cast<const double*>(c->tmp[0]) + 7 * n, n, d_valid, d_aux,
                     last_t ? d_last_t : nullptr, last_state ? d_last_state : nullptr);
  if (rc) return rc;
  HIP_TRY(c, hipMemcpyAsync(valid, d_valid, n, hipMemcpyDeviceToHost, c->stream));
  if (aux_out) HIP_TRY(c, hipMemcpyAsync(aux_out, d_aux, n * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  if (last_t) HIP_TRY(c, hipMemcpyAsync(last_t, d_last_t, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  if (last_t && last_state)
    HIP_TRY(c, hipMemcpyAsync(last_state, d_last_state, n * 7 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  return check_error_flag(c);
}

int artp_set_edge_passes(artp_ctx* c, int two_pass, int coarse_stride) {
  if (!c || (coarse_stride != 0 && coarse_stride < 2)) return ARTP_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  c->edge_two_pass = two_pass != 0;
  c->edge_coarse_stride = coarse_stride ? coarse_stride : ARTP_COARSE_STRIDE;
  return ARTP_OK;
}
int artp_set_few_edges(artp_ctx* c, int enabled) {
  if (!c) return ARTP_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  c->few_edges = enabled != 0;
  return ARTP_OK;
}
int artp_check_motions_dev(artp_ctx* c, const double* s1, const double* s2, size_t n, uint8_t* valid) {
  return run_edges_dev(c, 0, s1, s2, n, valid, nullptr);
}
int artp_check_motions(artp_ctx* c, const double* s1, const double* s2, size_t n, uint8_t* valid) {
  return run_edges_host(c, 0, s1, s2, n, valid, nullptr);
}
int artp_check_motions_last_valid_dev(artp_ctx* c, const double* s1, const double* s2, size_t n, uint8_t* valid,
                                      double* last_valid_t, double* last_valid_se3) {
  if (n && !last_valid_t) return ARTP_ERR_INVALID_ARG;
  return run_edges_dev(c, 0, s1, s2, n, valid, nullptr, last_valid_t, last_valid_se3);
}
int artp_check_motions_last_valid(artp_ctx* c, const double* s1, const double* s2, size_t n, uint8_t* valid,
                                  double* last_valid_t, double* last_valid_se3) {
  if (n && !last_valid_t) return ARTP_ERR_INVALID_ARG;
  return run_edges_host(c, 0, s1, s2, n, valid, nullptr, last_valid_t, last_valid_se3);
}
int artp_check_edges_interp_dev(artp_ctx* c, const double* s1, const double* s2, size_t n,
                                uint8_t* valid, uint32_t* n_interp_out) {
  return run_edges_dev(c, 1, s1, s2, n, valid, n_interp_out);
}
int artp_check_edges_interp(artp_ctx* c, const double* s1, const double* s2, size_t n, uint8_t* valid,
                            uint32_t* n_interp_out) {
  return run_edges_host(c, 1, s1, s2, n, valid, n_interp_out);
}


namespace {
struct Se3Row { double v[7]; };
}

int artp_debug_pipeline_counters(artp_ctx* c, uint64_t out[8]) {
  if (!c || !out) return ARTP_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  if (!c->tmp[5]) return ARTP_ERR_NO_MAP;
  HIP_TRY(c, hipSetDevice(c->device));
  uint64_t raw[16 + 2 * ARTP_NSUB * 16];
  HIP_TRY(c, hipMemcpyAsync(raw, c->tmp[5], sizeof(raw), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  for (int k = 0; k < 8; ++k) out[k] = raw[k];
  out[0] = out[4] = out[2] = out[7] = 0;  // the torso / foot queues and the two fallback queues: sums over the sub-queues
  for (int s = 0; s < ARTP_NSUB; ++s) {
    out[0] += raw[16 + (size_t)s * 16];
    out[4] += raw[16 + (size_t)(ARTP_NSUB + s) * 16];
    out[2] += raw[16 + (size_t)s * 16 + 2];                  // queue 6: torso boxes for the staged pass
    out[7] += raw[16 + (size_t)(ARTP_NSUB + s) * 16 + 2];    // queue 4: foot boxes for the lane scan
  }
  return ARTP_OK;
}

int artp_debug_partner_table(artp_ctx* c, int slot, uint8_t* out, size_t out_bytes, int* radius) {
  if (!c || slot < 0 || slot > 1 || !radius) return ARTP_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  if (!c->have_field[slot]) return ARTP_ERR_NO_MAP;
  const FieldDev& f = c->field[slot];
  *radius = f.partner_flags ? f.partner_R : 0;
  if (!out || !f.partner_flags) return ARTP_OK;
  const size_t need = (size_t)f.nW * f.nD;
  if (out_bytes < need) return ARTP_ERR_INVALID_ARG;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipMemcpyAsync(out, f.partner_flags, need, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return ARTP_OK;
}

int artp_compact_valid_dev(artp_ctx* c, const double* se3, const uint8_t* valid, size_t n,
                           double* out_se3, uint64_t* n_out_dev) {
  if (!c || (n && (!se3 || !valid || !out_se3)) || !n_out_dev) return ARTP_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  HIP_TRY(c, hipSetDevice(c->device));
  if (n == 0) {
    HIP_TRY(c, hipMemsetAsync(n_out_dev, 0, sizeof(uint64_t), c->stream));
    return ARTP_OK;
  }
  const Se3Row* in = reinterpret_cast<const Se3Row*>(se3);
  Se3Row* out = reinterpret_cast<Se3Row*>(out_se3);
  unsigned long long* cnt = reinterpret_cast<unsigned long long*>(n_out_dev);
  size_t need = 0;
  HIP_TRY(c, hipcub::DeviceSelect::Flagged(nullptr, need, in, valid, out, cnt, (int)n, c->stream));
  if (c->cub_cap < need) {
    if (c->cub_tmp) HIP_TRY(c, hipFree(c->cub_tmp));
    c->cub_tmp = nullptr;
    HIP_TRY(c, hipMalloc(&c->cub_tmp, need + 256));
    c->cub_cap = need + 256;
  }
  size_t cap = c->cub_cap;
  HIP_TRY(c, hipcub::DeviceSelect::Flagged(c->cub_tmp, cap, in, valid, out, cnt, (int)n, c->stream));
  return ARTP_OK;
}

int artp_compact_valid_indices_dev(artp_ctx* c, const uint8_t* valid, size_t n, uint32_t* out_idx,
                                   uint64_t* n_out_dev) {
  if (!c || (n && (!valid || !out_idx)) || !n_out_dev || n >= (1ull << 32)) return ARTP_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  HIP_TRY(c, hipSetDevice(c->device));
  if (n == 0) {
    HIP_TRY(c, hipMemsetAsync(n_out_dev, 0, sizeof(uint64_t), c->stream));
    return ARTP_OK;
  }
  hipcub::CountingInputIterator<uint32_t> in(0u);
  unsigned long long* cnt = reinterpret_cast<unsigned long long*>(n_out_dev);
  size_t need = 0;
  HIP_TRY(c, hipcub::DeviceSelect::Flagged(nullptr, need, in, valid, out_idx, cnt, (int)n, c->stream));
  if (c->cub_cap < need) {
    if (c->cub_tmp) HIP_TRY(c, hipFree(c->cub_tmp));
    c->cub_tmp = nullptr;
    HIP_TRY(c, hipMalloc(&c->cub_tmp, need + 256));
    c->cub_cap = need + 256;
  }
  size_t cap = c->cub_cap;
  HIP_TRY(c, hipcub::DeviceSelect::Flagged(c->cub_tmp, cap, in, valid, out_idx, cnt, (int)n, c->stream));
  return ARTP_OK;
}

namespace {
// record e of the exchange = {u32 i, u32 j, f32 cost[3]} of the sel[e]-th edge
__global__ void __launch_bounds__(256)
gather_edge_records_kernel(const uint32_t* __restrict__ sel, const unsigned long long* __restrict__ count,
                           const uint32_t* __restrict__ edge_i, const uint32_t* __restrict__ edge_j,
                           const float* __restrict__ cost, uint32_t* __restrict__ out) {
  const size_t n = (size_t)(*count);
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
    const uint32_t k = sel[e];
    out[5 * e + 0] = edge_i[k];
    out[5 * e + 1] = edge_j[k];
    out[5 * e + 2] = __float_as_uint(cost[3 * k + 0]);
    out[5 * e + 3] = __float_as_uint(cost[3 * k + 1]);
    out[5 * e + 4] = __float_as_uint(cost[3 * k + 2]);
  }
}
}  // namespace

int artp_pack_edge_results_dev(artp_ctx* c, const uint8_t* valid, const uint32_t* edge_i, const uint32_t* edge_j,
                               const float* cost, size_t n, uint32_t* records_out, uint64_t* n_out_dev) {
  if (!c || !n_out_dev || (n && (!valid || !edge_i || !edge_j || !cost || !records_out)) || n >= (1ull << 31))
    return ARTP_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  HIP_TRY(c, hipSetDevice(c->device));
  if (n == 0) {
    HIP_TRY(c, hipMemsetAsync(n_out_dev, 0, sizeof(uint64_t), c->stream));
    return ARTP_OK;
  }
  int rc = ensure_tmp(c, 6, n * sizeof(uint32_t));
  if (rc) return rc;
  uint32_t* sel = static_cast<uint32_t*>(c->tmp[6]);
  rc = artp_compact_valid_indices_dev(c, valid, n, sel, n_out_dev);  // input order is kept
  if (rc) return rc;
  size_t blocks = (n + 255) / 256;
  if (blocks > (size_t)c->n_cus * 16) blocks = (size_t)c->n_cus * 16;
  hipLaunchKernelGGL(gather_edge_records_kernel, dim3((unsigned)blocks), dim3(256), 0, c->stream, (const uint32_t*)sel,
                     reinterpret_cast<const unsigned long long*>(n_out_dev), edge_i, edge_j, cost, records_out);
  HIP_TRY(c, hipGetLastError());
  return ARTP_OK;
}

namespace {
// 64 labels -> one word (bit k of word w = valid[64 w + k] != 0)
__global__ void __launch_bounds__(256)
pack_valid_bits_kernel(const uint8_t* __restrict__ valid, size_t n, unsigned long long* __restrict__ bits) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned long long b = __ballot(i < n && valid[i] != 0);
  if ((threadIdx.x & 63) == 0 && (i & ~(size_t)63) < n) bits[i >> 6] = b;
}
struct BitAt {
  const unsigned long long* bits;
  __host__ __device__ __forceinline__ uint8_t operator()(uint32_t i) const {
    return (uint8_t)((bits[i >> 6] >> (i & 63u)) & 1ull);
  }
};
}  // namespace

int artp_pack_valid_bits_dev(artp_ctx* c, const uint8_t* valid, size_t n, uint64_t* bits_out) {
  if (!c || (n && (!valid || !bits_out))) return ARTP_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  if (n == 0) return ARTP_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  const size_t padded = (n + 63) & ~(size_t)63;  // whole wavefronts: the ballot covers the tail
  hipLaunchKernelGGL(pack_valid_bits_kernel, dim3((unsigned)((padded + 255) / 256)), dim3(256), 0, c->stream, valid, n,
                     reinterpret_cast<unsigned long long*>(bits_out));
  HIP_TRY(c, hipGetLastError());
  return ARTP_OK;
}

int artp_indices_from_bits_dev(artp_ctx* c, const uint64_t* bits, size_t n, uint32_t* out_idx, uint64_t* n_out_dev) {
  if (!c || (n && (!bits || !out_idx)) || !n_out_dev || n >= (1ull << 31)) return ARTP_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  HIP_TRY(c, hipSetDevice(c->device));
  if (n == 0) {
    HIP_TRY(c, hipMemsetAsync(n_out_dev, 0, sizeof(uint64_t), c->stream));
    return ARTP_OK;
  }
  hipcub::CountingInputIterator<uint32_t> in(0u);
  hipcub::TransformInputIterator<uint8_t, BitAt, hipcub::CountingInputIterator<uint32_t>> flags(
      in, BitAt{reinterpret_cast<const unsigned long long*>(bits)});
  unsigned long long* cnt = reinterpret_cast<unsigned long long*>(n_out_dev);
  size_t need = 0;
  HIP_TRY(c, hipcub::DeviceSelect::Flagged(nullptr, need, in, flags, out_idx, cnt, (int)n, c->stream));
  if (c->cub_cap < need) {
    if (c->cub_tmp) HIP_TRY(c, hipFree(c->cub_tmp));
    c->cub_tmp = nullptr;
    HIP_TRY(c, hipMalloc(&c->cub_tmp, need + 256));
    c->cub_cap = need + 256;
  }
  size_t cap = c->cub_cap;
  HIP_TRY(c, hipcub::DeviceSelect::Flagged(c->cub_tmp, cap, in, flags, out_idx, cnt, (int)n, c->stream));
  return ARTP_OK;
}

int artp_materialise_from_bits_dev(artp_ctx* c, uint64_t seed, const uint64_t* bits, int n_ranks, size_t words_per_rank,
                                   size_t prefix_bits, const uint64_t* base_index, size_t cap, double* se3_out,
                                   uint64_t* counts_dev) {
  if (!c || !bits || !base_index || !counts_dev || n_ranks < 1 || n_ranks > 16 || (cap && !se3_out) ||
      prefix_bits > words_per_rank * 64 || prefix_bits >= (1ull << 32))
    return ARTP_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  if (!c->have_sampler) return ARTP_ERR_NO_MAP;
  HIP_TRY(c, hipSetDevice(c->device));
  const size_t words = (prefix_bits + 63) / 64;
  if (words == 0 || cap == 0) {
    HIP_TRY(c, hipMemsetAsync(counts_dev, 0, (size_t)n_ranks * sizeof(uint64_t), c->stream));
    return ARTP_OK;
  }
  const size_t n_tiles = (words + ARTP_BITS_TILE - 1) / ARTP_BITS_TILE;
  int rc = ensure_tmp(c, 6, (size_t)n_ranks * (words + n_tiles) * sizeof(unsigned));
  if (rc) return rc;
  unsigned* offsets = static_cast<unsigned*>(c->tmp[6]);
  unsigned* tile_tot = offsets + (size_t)n_ranks * words;
  HIP_TRY(c, hipMemsetAsync(counts_dev, 0, (size_t)n_ranks * sizeof(uint64_t), c->stream));
  hipLaunchKernelGGL(bits_word_offsets_kernel, dim3((unsigned)n_tiles, (unsigned)n_ranks), dim3(256), 0, c->stream,
                     reinterpret_cast<const unsigned long long*>(bits), words_per_rank, words, prefix_bits, offsets, tile_tot,
                     reinterpret_cast<unsigned long long*>(counts_dev));
  RankBases bases{};
  for (int r = 0; r < n_ranks; ++r) bases.base[r] = base_index[r];
  // a lane per output state: at most min(cap, prefix) of them per rank (the kernel reads the rank's count)
  const size_t max_out = cap < prefix_bits ? cap : prefix_bits;
  const dim3 grid((unsigned)((max_out + 255) / 256), (unsigned)n_ranks);
  if (c->sampler.from_distribution)
    hipLaunchKernelGGL(materialise_from_bits_kernel<true>, grid, dim3(256), 0, c->stream, c->sampler, c->geom, c->robot, seed,
                       bases, reinterpret_cast<const unsigned long long*>(bits), words_per_rank, words, prefix_bits,
                       (const unsigned*)offsets, (const unsigned*)tile_tot, (int)n_tiles, cap,
                       reinterpret_cast<const unsigned long long*>(counts_dev), se3_out);
  else
    hipLaunchKernelGGL(materialise_from_bits_kernel<false>, grid, dim3(256), 0, c->stream, c->sampler, c->geom, c->robot, seed,
                       bases, reinterpret_cast<const unsigned long long*>(bits), words_per_rank, words, prefix_bits,
                       (const unsigned*)offsets, (const unsigned*)tile_tot, (int)n_tiles, cap,
                       reinterpret_cast<const unsigned long long*>(counts_dev), se3_out);
  HIP_TRY(c, hipGetLastError());
  return ARTP_OK;
}

int artp_sample_states_at_dev(artp_ctx* c, uint64_t seed, uint64_t base_index, const uint32_t* idx,
                              const uint64_t* count_dev, size_t cap, double* se3_out) {
  if (!c || !idx || !count_dev || (cap && !se3_out)) return ARTP_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  if (!c->have_sampler) return ARTP_ERR_NO_MAP;
  if (cap == 0) return ARTP_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  size_t blocks = (cap + 255) / 256;
  if (blocks > (size_t)c->n_cus * 16) blocks = (size_t)c->n_cus * 16;
  if (c->sampler.from_distribution)
    hipLaunchKernelGGL(sample_states_at_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, c->stream, c->sampler, c->geom,
                     c->robot, seed, base_index, idx, reinterpret_cast<const unsigned long long*>(count_dev), cap,
                     se3_out);
  else
    hipLaunchKernelGGL(sample_states_at_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, c->stream, c->sampler, c->geom,
                     c->robot, seed, base_index, idx, reinterpret_cast<const unsigned long long*>(count_dev), cap,
                     se3_out);
  HIP_TRY(c, hipGetLastError());
  return ARTP_OK;
}

int artp_algorithmic_vertices_dev(artp_ctx* c, const double* se3, size_t n, uint64_t* total) {
  if (!c || (n && !se3) || !total) return ARTP_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  if (!c->have_field[0] || !c->have_field[1]) return ARTP_ERR_NO_MAP;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipMemsetAsync(c->d_count, 0, sizeof(unsigned long long), c->stream));
  if (n) {
    size_t blocks = (n + 255) / 256;
    if (blocks > (size_t)c->n_cus * 32) blocks = (size_t)c->n_cus * 32;
    hipLaunchKernelGGL(alg_vertices_kernel, dim3((unsigned)blocks), dim3(256), 0, c->stream, c->field[0],
                       c->field[1], c->geom, c->robot, se3, n, c->d_count);
    HIP_TRY(c, hipGetLastError());
  }
  unsigned long long v = 0;
  HIP_TRY(c, hipMemcpyAsync(&v, c->d_count, sizeof(v), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  *total = v;
  return ARTP_OK;
}

}  // extern "C"

// ---- learned motion cost (R8 / R9) ------------------------------------------------------------------
namespace {

struct ConvSpec { int cout, cin, kh, kw, nt; };
// layers 2..6 of network.CNNpart (network_light.py:23-36)
const ConvSpec kConv[5] = {{24, 24, 3, 3, 2}, {48, 24, 3, 3, 3}, {48, 48, 3, 3, 3}, {48, 48, 3, 3, 3}, {48, 48, 15, 15, 3}};

size_t cost_blob_floats() {
  size_t n = 24 * 9 + 24;
  for (const ConvSpec& s : kConv) n += (size_t)s.cout * s.cin * s.kh * s.kw + s.cout;
  return n + FcWeights::TOTAL;
}

inline uint16_t f32_to_f16_bits(float f) {  // round-to-nearest-even, host side
  _Float16 h = (_Float16)f;
  uint16_t u;
  std::memcpy(&u, &h, 2);
  return u;
}

// The FC part (FcWeights blob: tar0, out0, three heads, three outputs) in the fragment order fc_cost_mfma_kernel reads
// (cost_kernels.h): tar0 composed into out0 in double, every weight as a half-float hi / lo pair.
inline void fc_mfma_pack(const float* w, std::vector<unsigned char>* out) {
  out->assign(FcMfma::TOTAL, 0);
  auto put16 = [&](size_t byte_off, float v) {
    const uint16_t b = f32_to_f16_bits(v);
    std::memcpy(out->data() + byte_off, &b, 2);
  };
  auto f16_value = [](float v) {  // the float a half-float rounding of v stands for
    const uint16_t h = f32_to_f16_bits(v);
    const uint32_t sign = (h >> 15) & 1u, ex = (h >> 10) & 31u, man = h & 1023u;
    double r;
    if (ex == 0) r = std::ldexp((double)man, -24);
    else if (ex == 31) r = man ? NAN : INFINITY;
    else r = std::ldexp((double)(man | 1024u), (int)ex - 25);
    return (float)(sign ? -r : r);
  };
  auto split = [&](double v, float* hi, float* lo) {
    *hi = f16_value((float)v);
    *lo = f16_value((float)(v - (double)*hi));
  };
  // out0 o tar0: k < 48 the map features, 48 .. 57 the ten geometric inputs, 58 the bias (input 1.0)
  double w0c[48][64];  // 24 KB of stack: no shared state between contexts loading weights on different threads
  for (int o = 0; o < 48; ++o) {
    for (int k = 0; k < 64; ++k) w0c[o][k] = 0.0;
    for (int k = 0; k < 48; ++k) w0c[o][k] = w[FcWeights::OUT0_W + o * 64 + k];
    double bias = w[FcWeights::OUT0_B + o];
    for (int m = 0; m < 16; ++m) {
      const double a = w[FcWeights::OUT0_W + o * 64 + 48 + m];
      for (int k = 0; k < 10; ++k) w0c[o][48 + k] += a * (double)w[FcWeights::TAR0_W + m * 10 + k];
      bias += a * (double)w[FcWeights::TAR0_B + m];
    }
    w0c[o][58] = bias;
  }
  auto hidden_of_row = [](int r) {  // accumulator row of the first GEMM -> hidden unit (see the kernel's header)
    if (r >= 32) return r;
    const int t = r / 16, g = (r % 16) / 4, i = r % 4;
    return 8 * g + 4 * t + i;
  };
  for (int s = 0; s < 2; ++s)
    for (int t = 0; t < 3; ++t)
      for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 8; ++j) {
          const int o = hidden_of_row(16 * t + (l & 15)), k = 32 * s + 8 * (l >> 4) + j;
          float hi, lo;
          split(w0c[o][k], &hi, &lo);
          put16(FcMfma::G1 + (size_t)((s * 3 + t) * 2 + 0) * 1024 + l * 16 + j * 2, hi);
          put16(FcMfma::G1 + (size_t)((s * 3 + t) * 2 + 1) * 1024 + l * 16 + j * 2, lo);
        }
  auto head_w = [&](int u, int k) -> double {
    if (u < 24) return w[FcWeights::H1_W + u * 48 + k];
    if (u < 48) return w[FcWeights::H2_W + (u - 24) * 48 + k];
    if (u < 84) return w[FcWeights::H3_W + (u - 48) * 48 + k];
    return 0.0;
  };
  for (int t = 0; t < 6; ++t)
    for (int l = 0; l < 64; ++l) {
      const int u = 16 * t + (l & 15);
      for (int j = 0; j < 8; ++j) {
        float hi, lo;
        split(head_w(u, 8 * (l >> 4) + j), &hi, &lo);
        put16(FcMfma::G2A + (size_t)(t * 2 + 0) * 1024 + l * 16 + j * 2, hi);
        put16(FcMfma::G2A + (size_t)(t * 2 + 1) * 1024 + l * 16 + j * 2, lo);
      }
      for (int j = 0; j < 4; ++j) {
        float hi, lo;
        split(head_w(u, 32 + 4 * (l >> 4) + j), &hi, &lo);
        put16(FcMfma::G2B + (size_t)(t * 2 + 0) * 512 + l * 8 + j * 2, hi);
        put16(FcMfma::G2B + (size_t)(t * 2 + 1) * 512 + l * 8 + j * 2, lo);
      }
    }
  float* bias2 = reinterpret_cast<float*>(out->data() + FcMfma::BIAS2);
  float* outw = reinterpret_cast<float*>(out->data() + FcMfma::OUT);
  float* ob = reinterpret_cast<float*>(out->data() + FcMfma::OB);
  for (int u = 0; u < 96; ++u) {
    bias2[u] = u < 24 ? w[FcWeights::H1_B + u] : (u < 48 ? w[FcWeights::H2_B + u - 24] : (u < 84 ? w[FcWeights::H3_B + u - 48] : 0.f));
    outw[0 * 96 + u] = u < 24 ? w[FcWeights::O1_W + u] : 0.f;
    outw[1 * 96 + u] = (u >= 24 && u < 48) ? w[FcWeights::O2_W + u - 24] : 0.f;
    outw[2 * 96 + u] = (u >= 48 && u < 84) ? w[FcWeights::O3_W + u - 48] : 0.f;
  }
  ob[0] = w[FcWeights::O1_B];
  ob[1] = w[FcWeights::O2_B];
  ob[2] = w[FcWeights::O3_B];
}

}  // namespace

// A 512-edge probe batch on a 6 x 6 pseudo-random feature map through fc_cost_mfma_kernel and fc_cost_kernel (same
// weights, same gather): fills fc_selfcheck / fc_selfcheck_err, clears fc_mfma when they differ by more than the hi / lo
// split's own error (a few 1e-6 in practice; 1e-4 absolute + 1e-4 relative allowed).
static int cost_fc_selfcheck(artp_ctx* c) {
  constexpr int F = 6, B = 512;
  std::vector<uint16_t> feat((size_t)F * F * 48);
  std::vector<float> edges((size_t)B * 6);
  uint32_t s = 0x9E3779B9u;
  auto rnd = [&]() {
    s = s * 1664525u + 1013904223u;
    return (float)((s >> 8) & 0xFFFFu) / 65536.0f;   // [0, 1)
  };
  for (auto& v : feat) v = f32_to_f16_bits(2.0f * rnd() - 0.5f);
  for (int e = 0; e < B; ++e) {
    const float sx = 1.0f + 4.0f * rnd(), sy = 1.0f + 4.0f * rnd();
    edges[6 * e + 3] = sx;
    edges[6 * e + 4] = sy;
    edges[6 * e + 5] = 6.0f * rnd() - 3.0f;
    edges[6 * e + 0] = sx + rnd() - 0.5f;
    edges[6 * e + 1] = sy + rnd() - 0.5f;
    edges[6 * e + 2] = 6.0f * rnd() - 3.0f;
  }
  CostMapGeom g;
  g.Fh = g.Fw = F;
  g.feat_res = 1.0;
  g.row_bias = g.col_bias = 0;
  g.cx = g.cy = 0.0;
  char* d = nullptr;
  const size_t fb = feat.size() * 2, eb = edges.size() * 4, cb = (size_t)B * 3 * 4;
  HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&d), fb + eb + 2 * cb + 64));
  half_t* d_feat = reinterpret_cast<half_t*>(d);
  float* d_edges = reinterpret_cast<float*>(d + ((fb + 15) & ~(size_t)15));
  float* d_c0 = d_edges + edges.size();
  float* d_c1 = d_c0 + (size_t)B * 3;
  hipError_t e1 = hipMemcpyAsync(d_feat, feat.data(), fb, hipMemcpyHostToDevice, c->stream);
  if (e1 == hipSuccess) e1 = hipMemcpyAsync(d_edges, edges.data(), eb, hipMemcpyHostToDevice, c->stream);
  std::vector<float> c0((size_t)B * 3), c1((size_t)B * 3);
  if (e1 == hipSuccess) {
    hipLaunchKernelGGL(fc_cost_mfma_kernel, dim3(2), dim3(256), 0, c->stream, (const float*)d_edges, (size_t)B, (const half_t*)d_feat, g,
                       (const char*)c->d_fc_mfma, d_c0);
    hipLaunchKernelGGL(fc_cost_kernel, dim3((B + 255) / 256), dim3(256), 0, c->stream, (const float*)d_edges, (size_t)B,
                       (const half_t*)d_feat, g, (const float*)c->d_fc, d_c1);
    e1 = hipGetLastError();
  }
  if (e1 == hipSuccess) e1 = hipMemcpyAsync(c0.data(), d_c0, cb, hipMemcpyDeviceToHost, c->stream);
  if (e1 == hipSuccess) e1 = hipMemcpyAsync(c1.data(), d_c1, cb, hipMemcpyDeviceToHost, c->stream);
  if (e1 == hipSuccess) e1 = hipStreamSynchronize(c->stream);
  (void)hipFree(d);
  HIP_TRY(c, e1);
  float worst = 0.f;
  bool ok = true;
  for (size_t i = 0; i < c0.size(); ++i) {
    const float diff = std::fabs(c0[i] - c1[i]);
    if (!(diff <= 1e-4f + 1e-4f * std::fabs(c1[i]))) ok = false;   // also catches NaN
    if (diff > worst || diff != diff) worst = diff;
  }
  c->fc_selfcheck = ok ? 1 : 0;
  c->fc_selfcheck_err = worst;
  if (!ok) {
    c->fc_mfma = 0;
    c->last_error = "motion cost: the MFMA form of the per-edge MLP disagreed with the fp32 kernel on the probe batch; "
                    "using the fp32 VALU kernels (see cost_kernels.h FCM_SHAPE_CHANGE)";
  }
  return ARTP_OK;
}

extern "C" {

size_t artp_cost_blob_bytes(void) { return 8 + cost_blob_floats() * sizeof(float); }

int artp_cost_load_weights(artp_ctx* c, const void* blob, size_t bytes) {
  if (!c || !blob) return ARTP_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  const unsigned char* p = static_cast<const unsigned char*>(blob);
  if (bytes != artp_cost_blob_bytes() || std::memcmp(p, "ARMC", 4) != 0 || p[4] != 1) {
    c->last_error = "motion-cost blob: bad magic, version or size";
    return ARTP_ERR_INVALID_ARG;
  }
  HIP_TRY(c, hipSetDevice(c->device));
  const float* w = reinterpret_cast<const float*>(p + 8);
  w += 216 + 24;  // conv1 (composed with conv2 below)
  for (int l = 0; l < 5; ++l) {
    const ConvSpec& s = kConv[l];
    const int krow = s.kw * s.cin, ksteps = (krow + 31) / 32;
    const size_t nfrag = l == 4 ? (size_t)s.kh * ksteps * s.nt * 64 : 0;  // per-row packing: the 15 x 15 layer only
    std::vector<uint16_t> packed(nfrag * 8, 0);
    // B fragment of v_mfma_f32_16x16x32_f16: lane l holds B[k = (l>>4)*8 + j][n = l&15]
    for (int kh = 0; l == 4 && kh < s.kh; ++kh)
      for (int ks = 0; ks < ksteps; ++ks)
        for (int nt = 0; nt < s.nt; ++nt)
          for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 8; ++j) {
              const int co = nt * 16 + (l & 15);
              const int kidx = ks * 32 + (l >> 4) * 8 + j;
              float v = 0.f;
              if (co < s.cout && kidx < krow) {
                const int kw = kidx / s.cin, ci = kidx % s.cin;
                v = w[(((size_t)co * s.cin + ci) * s.kh + kh) * s.kw + kw];  // torch [cout][cin][kh][kw]
              }
              packed[((((size_t)kh * ksteps + ks) * s.nt + nt) * 64 + l) * 8 + j] = f32_to_f16_bits(v);
            }
#ifdef ARTP_VARIANTS
    if (l == 4) {
      // the same layer for conv15_pair32_kernel: [kernel row -1 .. 15][k-step][k-half][channel][k-quad][8], kernel rows
      // -1 and 15 all zero (Conv15P32Cfg); element j of (channel co, quad kq) = W[co][k = 32 ks + 16 sh + 8 kq + j]
      using P32 = Conv15P32Cfg<6>;
      std::vector<uint16_t> p32(P32::W_FRAGS * 8, 0);
      for (int kh = 0; kh < s.kh; ++kh)
        for (int ks = 0; ks < ksteps; ++ks)
          for (int sh = 0; sh < 2; ++sh)
            for (int co = 0; co < s.cout; ++co)
              for (int kq = 0; kq < 2; ++kq)
                for (int j = 0; j < 8; ++j) {
                  const int kidx = ks * 32 + sh * 16 + kq * 8 + j;
                  if (kidx >= krow) continue;
                  const int kw = kidx / s.cin, ci = kidx % s.cin;
                  const float v = w[(((size_t)co * s.cin + ci) * s.kh + kh) * s.kw + kw];
                  p32[((size_t)(kh + 1) * P32::KH_FR + (size_t)ks * P32::KS_FR + sh * P32::SH_FR + co * 2 + kq) * 8 + j] =
                      f32_to_f16_bits(v);
                }
      if (c->d_convw_p32) HIP_TRY(c, hipFree(c->d_convw_p32));
      c->d_convw_p32 = nullptr;
      HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->d_convw_p32), p32.size() * 2));
      HIP_TRY(c, hipMemcpy(c->d_convw_p32, p32.data(), p32.size() * 2, hipMemcpyHostToDevice));
    }
#endif
    w += (size_t)s.cout * s.cin * s.kh * s.kw;
    if (c->d_convw[l]) HIP_TRY(c, hipFree(c->d_convw[l]));
    c->d_convw[l] = nullptr;
    if (c->d_convb[l]) HIP_TRY(c, hipFree(c->d_convb[l]));
    c->d_convb[l] = nullptr;
    if (l == 4) {
      HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->d_convw[l]), packed.size() * 2));
      HIP_TRY(c, hipMemcpy(c->d_convw[l], packed.data(), packed.size() * 2, hipMemcpyHostToDevice));
    }
    HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->d_convb[l]), 64 * sizeof(float)));
    float bias[64] = {0};
    std::memcpy(bias, w, s.cout * sizeof(float));
    HIP_TRY(c, hipMemcpy(c->d_convb[l], bias, sizeof(bias), hipMemcpyHostToDevice));
    w += s.cout;
  }
  {
    // conv1 (+BN) followed by conv2 (+BN) without an activation in between (network_light.py:84-87) is one 5 x 5
    // convolution: Wc[co][a+c][b+d] += W2[co][ci][a][b] * W1[ci][c][d], bias b2[co] + sum W2[co][ci][a][b] * b1[ci]
    const float* base = reinterpret_cast<const float*>(p + 8);
    const float* w1 = base;            // [24][3][3]
    const float* b1 = base + 216;      // [24]
    const float* w2 = base + 240;      // [24][24][3][3]
    const float* b2 = w2 + 24 * 24 * 9;
    double wc[24][25], bc[24];
    for (int co = 0; co < 24; ++co) {
      for (int k = 0; k < 25; ++k) wc[co][k] = 0.0;
      bc[co] = b2[co];
      for (int ci = 0; ci < 24; ++ci)
        for (int a = 0; a < 3; ++a)
          for (int b = 0; b < 3; ++b) {
            const double v2 = w2[((co * 24 + ci) * 3 + a) * 3 + b];
            bc[co] += v2 * b1[ci];
            for (int cc = 0; cc < 3; ++cc)
              for (int d = 0; d < 3; ++d) wc[co][(a + cc) * 5 + (b + d)] += v2 * (double)w1[ci * 9 + cc * 3 + d];
          }
    }
    float h12[24 * 25 + 24];
    for (int co = 0; co < 24; ++co) {
      for (int k = 0; k < 25; ++k) h12[co * 25 + k] = (float)wc[co][k];
      h12[600 + co] = (float)bc[co];
    }
    if (!c->d_c12) HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->d_c12), sizeof(h12)));
    HIP_TRY(c, hipMemcpy(c->d_c12, h12, sizeof(h12), hipMemcpyHostToDevice));
    {
      // conv12_mfma_kernel's operand: [channel tile t][hi, lo][lane][8 half floats] + bias[32].  Lane l = (m = l & 15, g = l >> 4)
      // holds channel 16 t + m, K slots 8 g .. 8 g + 7; slot s < 30 = window row s / 6, tap s % 6 (tap 5 and slots 30, 31: 0).
      // w = hi + lo, both half floats, split from the double (22 bits).
      std::vector<unsigned char> blob((size_t)C12M_FRAG_HALFS * 2 + 32 * sizeof(float), 0);
      for (int t = 0; t < 2; ++t)
        for (int l = 0; l < 64; ++l)
          for (int j = 0; j < 8; ++j) {
            const int ch = 16 * t + (l & 15), sl = 8 * (l >> 4) + j;
            double v = 0.0;
            if (ch < 24 && sl < 30 && sl % 6 < 5) v = wc[ch][(sl / 6) * 5 + sl % 6];
            const _Float16 hi = (_Float16)v;
            const _Float16 lo = (_Float16)(v - (double)hi);
            std::memcpy(blob.data() + ((((size_t)t * 2 + 0) * 64 + l) * 8 + j) * 2, &hi, 2);
            std::memcpy(blob.data() + ((((size_t)t * 2 + 1) * 64 + l) * 8 + j) * 2, &lo, 2);
          }
      float b32[32] = {0};
      for (int co = 0; co < 24; ++co) b32[co] = (float)bc[co];
      std::memcpy(blob.data() + (size_t)C12M_FRAG_HALFS * 2, b32, sizeof(b32));
      if (!c->d_c12m) HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->d_c12m), blob.size()));
      HIP_TRY(c, hipMemcpy(c->d_c12m, blob.data(), blob.size(), hipMemcpyHostToDevice));
    }
    // conv3..5 for conv345_kernel: K walked in 16-byte chunks over the whole (kh, kw, cin) window; fragment
    // [ks][nt][lane][8]: lane l, element j holds B[k][n] with chunk q = 4 ks + (l >> 4), k = 8 q + j, n = 16 nt + (l & 15)
    const float* wl = w2 + 24 * 24 * 9 + 24;
    for (int l = 0; l < 3; ++l) {
      const ConvSpec& s = kConv[l + 1];
      const int cpr = 3 * s.cin / 8, q_tot = 3 * cpr, ks_tot = (q_tot + 3) / 4;
      std::vector<uint16_t> packed((size_t)ks_tot * 3 * 64 * 8, 0);
      for (int ks = 0; ks < ks_tot; ++ks)
        for (int nt = 0; nt < 3; ++nt)
          for (int ln = 0; ln < 64; ++ln)
            for (int j = 0; j < 8; ++j) {
              const int q = ks * 4 + (ln >> 4), co = nt * 16 + (ln & 15);
              float v = 0.f;
              if (q < q_tot && co < s.cout) {
                const int kh = q / cpr, i = (q % cpr) * 8 + j, kw = i / s.cin, ci = i % s.cin;
                v = wl[(((size_t)co * s.cin + ci) * 3 + kh) * 3 + kw];
              }
              packed[(((size_t)ks * 3 + nt) * 64 + ln) * 8 + j] = f32_to_f16_bits(v);
            }
      if (c->d_convw_chunk[l]) HIP_TRY(c, hipFree(c->d_convw_chunk[l]));
      c->d_convw_chunk[l] = nullptr;
      HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->d_convw_chunk[l]), packed.size() * 2));
      HIP_TRY(c, hipMemcpy(c->d_convw_chunk[l], packed.data(), packed.size() * 2, hipMemcpyHostToDevice));
      wl += (size_t)s.cout * s.cin * 9 + s.cout;
    }
  }
  if (!c->d_fc) HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->d_fc), FcWeights::TOTAL * sizeof(float)));
  HIP_TRY(c, hipMemcpy(c->d_fc, w, FcWeights::TOTAL * sizeof(float), hipMemcpyHostToDevice));
  {
    std::vector<unsigned char> blob;
    fc_mfma_pack(w, &blob);
    if (!c->d_fc_mfma) HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->d_fc_mfma), blob.size()));
    HIP_TRY(c, hipMemcpy(c->d_fc_mfma, blob.data(), blob.size(), hipMemcpyHostToDevice));
    c->fc_mfma = c->fc_mfma_wanted;   // artp_cost_set_fc_path
  }
  c->have_weights = true;
  // The MFMA form of the MLP depends on a software-managed hazard of the matrix pipe (cost_kernels.h FCM_SHAPE_CHANGE) that
  // only the device can confirm for the toolchain that built this library: a probe batch goes through it and through the
  // fp32 VALU kernel; if they disagree the context falls back to the fp32 kernels and says so (artp_cost_fc_path).
  if (c->fc_mfma) {
    const int rc = cost_fc_selfcheck(c);
    if (rc != ARTP_OK) return rc;
  }
  return ARTP_OK;
}

int artp_cost_set_fc_path(artp_ctx* c, int mfma) {
  if (!c) return ARTP_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  c->fc_mfma_wanted = mfma != 0;
  if (!c->have_weights) return ARTP_OK;       // takes effect at artp_cost_load_weights
  if (!mfma) {
    c->fc_mfma = 0;
    return ARTP_OK;
  }
  if (c->fc_mfma) return ARTP_OK;
  c->fc_mfma = 1;
  return cost_fc_selfcheck(c);                // back to the matrix cores only through the probe batch
}
int artp_cost_fc_path(artp_ctx* c, int* mfma, int* selfcheck, float* max_abs_diff) {
  if (!c) return ARTP_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  if (mfma) *mfma = c->fc_mfma;
  if (selfcheck) *selfcheck = c->fc_selfcheck;
  if (max_abs_diff) *max_abs_diff = c->fc_selfcheck_err;
  return ARTP_OK;
}

static int cost_run_cnn(artp_ctx* c, const float* d_map, int H, int W) {
  // shapes: network_light.py:84-107
  const int h1 = H - 2, w1 = W - 2, h2 = h1 - 2, w2 = w1 - 2, hp = h2 / 2, wpp = w2 / 2;
  const int h3 = hp - 2, w3 = wpp - 2, h4 = h3 - 2, w4 = w3 - 2, hq = h4 - 2, wq = w4 - 2;
  const int h5 = hq - 2, w5 = wq - 2, hf = h5 - 14, wf = w5 - 14;
  if (hf < 3 || wf < 3) {
    c->last_error = "map too small for the motion-cost feature extractor";
    return ARTP_ERR_INVALID_ARG;
  }
  const size_t act_bytes = (size_t)h1 * w1 * 24 * 2 + 8192;
  if (c->act_cap < act_bytes) {
    for (int l = 0; l < 2; ++l) {
      if (c->d_act[l]) HIP_TRY(c, hipFree(c->d_act[l]));
      c->d_act[l] = nullptr;
      HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->d_act[l]), act_bytes));
      HIP_TRY(c, hipMemset(c->d_act[l], 0, act_bytes));
    }
    c->act_cap = act_bytes;
  }
  const size_t feat_bytes = (size_t)hf * wf * 48 * 2 + 256;
  if (c->feat_cap < feat_bytes) {
    if (c->d_feat) HIP_TRY(c, hipFree(c->d_feat));
    c->d_feat = nullptr;
    HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->d_feat), feat_bytes));
    c->feat_cap = feat_bytes;
  }
  half_t* A = c->d_act[0];
  half_t* Bf = c->d_act[1];
  hipStream_t st = c->stream;
  bool fuse12 = false;
  {
    // round 3: three launches.  (A) conv1 o conv2 + lrelu + pool2 straight from the f32 map -> A [hp][wpp][24];
    // (B) conv3 -> conv4 -> pool3 -> conv5 with LDS-resident halo tiles -> Bf [h5][w5][48]; (C) the 15 x 15 layer.
    // round 5: (A) on the matrix cores -- inside (B)'s patch phase (conv345_kernel<T, true, true>: no launch, no 24-channel
    // image), or as a launch of its own ($ARTP_CONV12_FUSED=0: conv12_mfma_kernel); the VALU form stays behind $ARTP_CONV12_MFMA=0
    fuse12 = true;   // the product: conv1 o conv2 inside conv345's patch phase, XCD-aware tile numbering
#ifdef ARTP_VARIANTS
    bool c12m = true, xcd = true;
    // tuning / tests, read at every update: conv1 o conv2 as a launch of its own ($ARTP_CONV12_FUSED=0: conv12_mfma_kernel),
    // its VALU form ($ARTP_CONV12_MFMA=0: conv12_pool_kernel), launch-order tile numbering ($ARTP_CNN_XCD=0)
    const char* ecm = std::getenv("ARTP_CONV12_MFMA");
    const char* ecf = std::getenv("ARTP_CONV12_FUSED");
    const char* ecx = std::getenv("ARTP_CNN_XCD");
    if (ecx) xcd = ecx[0] != '0';
    c12m = ecm ? ecm[0] != '0' : c->conv12_mfma != 0;
    fuse12 = c12m && (ecf ? ecf[0] != '0' : true) && xcd;
    if (fuse12) {
    } else if (c12m) {
      const unsigned blocks_m = (unsigned)(((wpp + C12M_PX - 1) / C12M_PX) * ((hp + C12M_PY - 1) / C12M_PY));
      hipLaunchKernelGGL(conv12_mfma_kernel, dim3(blocks_m), dim3(256), 0, st, d_map, H, W, (const half8*)c->d_c12m,
                         (const float*)(c->d_c12m + (size_t)C12M_FRAG_HALFS * 2), A);
    } else {
      const unsigned blocks_a = (unsigned)(((wpp + C12_PT - 1) / C12_PT) * ((hp + C12_PT - 1) / C12_PT));
      hipLaunchKernelGGL(conv12_pool_kernel, dim3(blocks_a), dim3(256), 0, st, d_map, H, W,
                         (const float*)c->d_c12, (const float*)(c->d_c12 + 600), A);
    }
#endif
    // tile edge of (B): one workgroup per CU, and a partly filled last round costs a full round -- rounds x patch
    // area decides (400 x 400: 144 tiles of 16 = one round; 800 x 800: 484 tiles of 18 = two rounds against three of 16)
    auto rounds_cost = [&](int t) {
      const long tiles = (long)((w5 + t - 1) / t) * ((h5 + t - 1) / t);
      return ((tiles + c->n_cus - 1) / c->n_cus) * (long)(t + 8) * (t + 8);
    };
    auto launch_b = [&](auto k345, int lds, int t) -> int {
      HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k345), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      const unsigned blocks_b = (unsigned)(((w5 + t - 1) / t) * ((h5 + t - 1) / t));
      hipLaunchKernelGGL(k345, dim3(blocks_b), dim3(C345_NT), lds, st, (const half_t*)A, hp, wpp,
                         (const half8*)c->d_convw_chunk[0], (const float*)c->d_convb[1], (const half8*)c->d_convw_chunk[1],
                         (const float*)c->d_convb[2], (const half8*)c->d_convw_chunk[2], (const float*)c->d_convb[3], Bf,
                         (const float*)d_map, H, W, (const half8*)c->d_c12m,
                         (const float*)(c->d_c12m + (size_t)C12M_FRAG_HALFS * 2));
      return ARTP_OK;
    };
    // (round 4: 12 added -- 400 x 400 is 256 tiles of 12 = ONE round on all 256 CUs against 144 tiles of 16 on 144 CUs)
    int t_best = 16;
    for (int t : {12, 18})
      if (rounds_cost(t) < rounds_cost(t_best)) t_best = t;
#ifdef ARTP_VARIANTS
    if (const char* ev = std::getenv("ARTP_C345_T")) t_best = std::atoi(ev);  // tuning
#endif
    int rcb;
    if (fuse12)
      rcb = t_best == 18   ? launch_b(conv345_kernel<18, true, true>, C345Cfg<18>::LDS_BYTES, 18)
            : t_best == 12 ? launch_b(conv345_kernel<12, true, true>, C345Cfg<12>::LDS_BYTES, 12)
                           : launch_b(conv345_kernel<16, true, true>, C345Cfg<16>::LDS_BYTES, 16);
#ifdef ARTP_VARIANTS
    else
      rcb = xcd ? (t_best == 18   ? launch_b(conv345_kernel<18>, C345Cfg<18>::LDS_BYTES, 18)
                   : t_best == 12 ? launch_b(conv345_kernel<12>, C345Cfg<12>::LDS_BYTES, 12)
                                  : launch_b(conv345_kernel<16>, C345Cfg<16>::LDS_BYTES, 16))
                : (t_best == 18   ? launch_b(conv345_kernel<18, false>, C345Cfg<18>::LDS_BYTES, 18)
                   : t_best == 12 ? launch_b(conv345_kernel<12, false>, C345Cfg<12>::LDS_BYTES, 12)
                                  : launch_b(conv345_kernel<16, false>, C345Cfg<16>::LDS_BYTES, 16));
#else
    else
      return ARTP_ERR_INVALID_ARG;   // not reachable: the product always fuses
#endif
    if (rcb != ARTP_OK) return rcb;
    HIP_TRY(c, hipGetLastError());
    A = Bf;  // the 15 x 15 layer below reads conv5's output
  }
  {
    // Tile height of the 15 x 15 layer: two workgroups per CU run at once, and the launch's last, partly filled
    // round costs a full round's time.  Take the candidate height with the fewest output rows computed per CU
    // (rounds x height); ties go to the smaller tile (less padding).
    const int slots = 2 * c->n_cus;
    const int cands[3] = {8, 9, 10};  // (4- and 6-row tiles were measured at 400^2: 54.5 / 60.0 us against 52.7 for 8)
    int best = 8;
    long best_cost = -1;
    for (int tr : cands) {
      const long tiles = (long)((wf + 15) / 16) * ((hf + tr - 1) / tr);
      const long cost = ((tiles + slots - 1) / slots) * tr;
      if (best_cost < 0 || cost < best_cost) {
        best_cost = cost;
        best = tr;
      }
    }
#ifndef ARTP_VARIANTS
    // the product: conv_ksplit_kernel, XCD-aware tile order.  No more 8-row tiles than CUs (C3: 242): one 8-wavefront
    // workgroup per CU, two wavefronts per SIMD; else 4-wavefront workgroups, two per CU, of the tile height chosen above
    auto launch = [&](auto kfn, int lds, int tr, int threads = 256) -> int {
      HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      const unsigned blocks = (unsigned)(((wf + 15) / 16) * ((hf + tr - 1) / tr));
      hipLaunchKernelGGL(kfn, dim3(blocks), dim3(threads), lds, st, (const half_t*)A, h5, w5, (const half8*)c->d_convw[4],
                         (const float*)c->d_convb[4], c->d_feat);
      return ARTP_OK;
    };
    int rcl;
    const long tiles8 = (long)((wf + 15) / 16) * ((hf + 7) / 8);
    if (tiles8 <= c->n_cus)
      rcl = launch(conv_ksplit_kernel<15, 15, 48, 48, 3, true, 8, 8>, ConvKsplitCfg<15, 15, 48, 48, 3, true, 8, 8>::LDS_BYTES, 8, 512);
    else if (best == 9)
      rcl = launch(conv_ksplit_kernel<15, 15, 48, 48, 3, true, 9>, ConvKsplitCfg<15, 15, 48, 48, 3, true, 9>::LDS_BYTES, 9);
    else if (best == 10)
      rcl = launch(conv_ksplit_kernel<15, 15, 48, 48, 3, true, 10>, ConvKsplitCfg<15, 15, 48, 48, 3, true, 10>::LDS_BYTES, 10);
    else
      rcl = launch(conv_ksplit_kernel<15, 15, 48, 48, 3, true, 8>, ConvKsplitCfg<15, 15, 48, 48, 3, true, 8>::LDS_BYTES, 8);
#else
    // the variants build: every form that was built and measured, behind its environment switch (read at every update)
    auto launch = [&](auto kfn, int lds, int tr, int threads = 256) -> int {
      if (const char* evp = std::getenv("ARTP_KSPLIT_ONE_PER_CU")) {   // experiment: a workgroup on its own (LDS padded past half a CU's)
        if (evp[0] != '0' && lds < 96 * 1024) lds = 96 * 1024;
      }
      HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      const unsigned blocks = (unsigned)(((wf + 15) / 16) * ((hf + tr - 1) / tr));
      hipLaunchKernelGGL(kfn, dim3(blocks), dim3(threads), lds, st, (const half_t*)A, h5, w5, (const half8*)c->d_convw[4],
                         (const float*)c->d_convb[4], c->d_feat);
      return ARTP_OK;
    };
    int rcl;
    // round 5: the persistent strip-walking form (conv_kwalk_kernel): one workgroup per CU for the whole launch, a run
    // of vertically adjacent tiles each.  Tile height = the candidate with the shortest longest run (rows per CU).
    const char* evk = std::getenv("ARTP_KWALK");
    const bool kwalk = evk ? std::atoi(evk) != 0 : false;   // measured (profiles/r05_cnn_variants.txt): fewer cycles, lower clock, no faster
    if (kwalk) {
      int trk = 8;
      long costk = -1;
      for (int tr : cands) {
        const long tiles = (long)((wf + 15) / 16) * ((hf + tr - 1) / tr);
        const long g = tiles < c->n_cus ? tiles : c->n_cus;
        const long cost = ((tiles + g - 1) / g) * tr;
        if (costk < 0 || cost < costk) {
          costk = cost;
          trk = tr;
        }
      }
      if (const char* ev = std::getenv("ARTP_KWALK_TR")) trk = std::atoi(ev);  // tuning
      auto launch_w = [&](auto kfn, int lds, int tr, int threads) -> int {
        HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        const int tps = (hf + tr - 1) / tr, tiles = ((wf + 15) / 16) * tps;
        const int g = tiles < c->n_cus ? tiles : c->n_cus;
        hipLaunchKernelGGL(kfn, dim3((unsigned)g), dim3(threads), lds, st, (const half_t*)A, h5, w5, (const half8*)c->d_convw[4],
                           (const float*)c->d_convb[4], c->d_feat, tps, tiles);
        return ARTP_OK;
      };
      int variant = 0;   // 0: BD 5, rows early; 1: BD 5, rows late; 2: BD 15, rows late  (tuning)
      if (const char* ev = std::getenv("ARTP_KWALK_VARIANT")) variant = std::atoi(ev);
#define ARTP_KW_LAUNCH(TRv, BDv, EARLYv) launch_w(conv_kwalk_kernel<TRv, BDv, EARLYv>, KwalkCfg<TRv, BDv>::LDS_BYTES, TRv, KwalkCfg<TRv, BDv>::NTH)
      if (variant == 2)
        rcl = trk == 9 ? ARTP_KW_LAUNCH(9, 15, false) : trk == 10 ? ARTP_KW_LAUNCH(10, 15, false) : ARTP_KW_LAUNCH(8, 15, false);
      else if (variant == 1)
        rcl = trk == 9 ? ARTP_KW_LAUNCH(9, 5, false) : trk == 10 ? ARTP_KW_LAUNCH(10, 5, false) : ARTP_KW_LAUNCH(8, 5, false);
      else
        rcl = trk == 9 ? ARTP_KW_LAUNCH(9, 5, true) : trk == 10 ? ARTP_KW_LAUNCH(10, 5, true) : ARTP_KW_LAUNCH(8, 5, true);
#undef ARTP_KW_LAUNCH
      if (rcl != ARTP_OK) return rcl;
      HIP_TRY(c, hipGetLastError());
      c->feat_h = hf;
      c->feat_w = wf;
      return ARTP_OK;
    }
    // no more 8-row tiles than CUs (C3: 242): one 8-wavefront workgroup per CU, two wavefronts per SIMD
    const long tiles8 = (long)((wf + 15) / 16) * ((hf + 7) / 8);
    const char* ev8 = std::getenv("ARTP_KSPLIT_NWV");
    const bool wide = ev8 ? std::atoi(ev8) == 8 : tiles8 <= c->n_cus;
    const char* evx2 = std::getenv("ARTP_CNN_XCD");   // tuning: 0 = tiles in launch order (rounds 3-4)
    const bool xcd2 = evx2 ? std::atoi(evx2) != 0 : true;
    // $ARTP_KSPLIT_MS=2 (more tiles than CUs, 800^2): 18-row tiles, one 8-wavefront workgroup per CU, K slice x row half
    // (conv_ksplit_kernel MS = 2).  Built to keep two wavefronts per SIMD in the main loop at all times; measured the same
    // (launch 111.6 us against 108-111: the main loop runs at 17 cycles per MFMA either way, at ~1.65 GHz), so not the default.
    const char* evm = std::getenv("ARTP_KSPLIT_MS");
    const bool ms2 = !wide && xcd2 && evm && std::atoi(evm) == 2;
    // round 6: more tiles than CUs -> the row-pair form on v_mfma_f32_32x32x16_f16, one 4-wavefront workgroup per CU; tile
    // height 6 or 8 by rounds x height (800^2: 63 x 12 = 756 tiles of 6 rows = 2.95 rounds of 256)
    const char* ev32 = std::getenv("ARTP_CONV15_PAIR32");   // A/B against conv_ksplit_kernel (tuning)
    const bool pair32 = !wide && xcd2 && (ev32 && std::atoi(ev32) != 0);
    if (pair32) {
      auto cost32 = [&](int tr) {
        const long tiles = (long)((wf + 31) / 32) * ((hf + tr - 1) / tr);
        return ((tiles + c->n_cus - 1) / c->n_cus) * (long)tr;
      };
      auto launch32 = [&](auto kfn, int lds, int tr) -> int {
        HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        const unsigned blocks = (unsigned)(((wf + 31) / 32) * ((hf + tr - 1) / tr));
        hipLaunchKernelGGL(kfn, dim3(blocks), dim3(256), lds, st, (const half_t*)A, h5, w5, (const half8*)c->d_convw_p32,
                           (const float*)c->d_convb[4], c->d_feat);
        return ARTP_OK;
      };
      rcl = cost32(6) <= cost32(8) ? launch32(conv15_pair32_kernel<6>, Conv15P32Cfg<6>::LDS_BYTES, 6)
                                   : launch32(conv15_pair32_kernel<8>, Conv15P32Cfg<8>::LDS_BYTES, 8);
    } else if (wide && !xcd2)
      rcl = launch(conv_ksplit_kernel<15, 15, 48, 48, 3, true, 8, 8, false>, ConvKsplitCfg<15, 15, 48, 48, 3, true, 8, 8>::LDS_BYTES, 8, 512);
    else if (wide)
      rcl = launch(conv_ksplit_kernel<15, 15, 48, 48, 3, true, 8, 8>, ConvKsplitCfg<15, 15, 48, 48, 3, true, 8, 8>::LDS_BYTES, 8, 512);
    else if (ms2)
      rcl = launch(conv_ksplit_kernel<15, 15, 48, 48, 3, true, 9, 8, true, 3, 2>, ConvKsplitCfg<15, 15, 48, 48, 3, true, 9, 8, 2>::LDS_BYTES, 18, 512);
    else if (best == 9 && !xcd2)
      rcl = launch(conv_ksplit_kernel<15, 15, 48, 48, 3, true, 9, 4, false>, ConvKsplitCfg<15, 15, 48, 48, 3, true, 9>::LDS_BYTES, 9);
    else if (best == 9)
      rcl = launch(conv_ksplit_kernel<15, 15, 48, 48, 3, true, 9>, ConvKsplitCfg<15, 15, 48, 48, 3, true, 9>::LDS_BYTES, 9);
    else if (best == 10)
      rcl = launch(conv_ksplit_kernel<15, 15, 48, 48, 3, true, 10>, ConvKsplitCfg<15, 15, 48, 48, 3, true, 10>::LDS_BYTES, 10);
    else
      rcl = launch(conv_ksplit_kernel<15, 15, 48, 48, 3, true, 8>, ConvKsplitCfg<15, 15, 48, 48, 3, true, 8>::LDS_BYTES, 8);
#endif
    if (rcl != ARTP_OK) return rcl;
  }
  HIP_TRY(c, hipGetLastError());
  c->feat_h = hf;
  c->feat_w = wf;
  return ARTP_OK;
}

static int cost_update_map_impl(artp_ctx* c, const float* elev_xy, bool on_device, int rows, int cols, double res,
                                double len_x, double len_y, double cx, double cy) {
  if (!c || !elev_xy || rows < 1 || cols < 1 || !(res > 0)) return ARTP_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  if (!c->have_weights) return ARTP_ERR_NO_WEIGHTS;
  HIP_TRY(c, hipSetDevice(c->device));
  const size_t n = (size_t)rows * cols;
  const float* d_map = elev_xy;  // a device array is read where it lies (stream-ordered, like every _dev argument)
  if (!on_device) {
    if (c->map_cap < n) {
      if (c->d_map_f32) HIP_TRY(c, hipFree(c->d_map_f32));
      c->d_map_f32 = nullptr;
      HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->d_map_f32), n * sizeof(float)));
      c->map_cap = n;
    }
    HIP_TRY(c, hipMemcpyAsync(c->d_map_f32, elev_xy, n * sizeof(float), hipMemcpyHostToDevice, c->stream));
    d_map = c->d_map_f32;
  }
  const int rc = cost_run_cnn(c, d_map, rows, cols);
  if (rc) return rc;
  // CostQuery.setMapParams (cost_query.py:26-35): featureResFactor = 2, mapClip = 24
  CostMapGeom& g = c->cost_geom;
  g.Fh = c->feat_h;  // predictor.features.shape[2] / shape[3] (cost_query.py:54-55): maps need not be square
  g.Fw = c->feat_w;
  g.feat_res = res * 2;
  g.row_bias = (int)((len_x / res - 2 * 24) / 2 * 0.5);
  g.col_bias = (int)((len_y / res - 2 * 24) / 2 * 0.5);
  g.cx = cx;
  g.cy = cy;
  if (!on_device) HIP_TRY(c, hipStreamSynchronize(c->stream));  // the caller's host buffer is free again
  c->have_features = true;
  return ARTP_OK;
}

int artp_cost_update_map(artp_ctx* c, const float* elev_xy, int rows, int cols, double res, double len_x,
                         double len_y, double cx, double cy) {
  return cost_update_map_impl(c, elev_xy, false, rows, cols, res, len_x, len_y, cx, cy);
}

// the same with the map array already in HBM; asynchronous on the context's stream
int artp_cost_update_map_dev(artp_ctx* c, const float* elev_xy_dev, int rows, int cols, double res, double len_x,
                             double len_y, double cx, double cy) {
  return cost_update_map_impl(c, elev_xy_dev, true, rows, cols, res, len_x, len_y, cx, cy);
}

// cost_query_server.py:46-74 receives the planner's grid_map layer and stores it as
// np.rot90(layer_as_sent, 2).transpose(); for the Eigen matrix layer(i, j) (column-major) that is the array
// a[r][c] = layer(rows-1-r, cols-1-c): index r grows along world x, c along world y.
int cost_update_map_layer_filled(artp_ctx* c, const float* layer, int rows, int cols, double res, double len_x,
                                 double len_y, double pos_x, double pos_y);

int artp_cost_update_map_layer(artp_ctx* c, const float* layer, int rows, int cols, double res, double len_x,
                               double len_y, double pos_x, double pos_y) {
  if (!c || !layer || rows < 1 || cols < 1) return ARTP_ERR_INVALID_ARG;
  std::vector<float> a((size_t)rows * cols);
  for (int r = 0; r < rows; ++r)
    for (int k = 0; k < cols; ++k) {
      const float v = layer[(size_t)(rows - 1 - r) + (size_t)(cols - 1 - k) * rows];
      if (!std::isfinite(v) && c->cost_fill_holes)
        return cost_update_map_layer_filled(c, layer, rows, cols, res, len_x, len_y, pos_x, pos_y);
      if (!std::isfinite(v)) {
        // the server inpaints holes (cost_query_server.py:90-111): artp_cost_set_hole_filling(ctx, 1) does it here
        c->last_error = "elevation layer has holes (NaN / inf): inpaint first or enable artp_cost_set_hole_filling";
        return ARTP_ERR_INVALID_ARG;
      }
      a[(size_t)r * cols + k] = v;
    }
  return artp_cost_update_map(c, a.data(), rows, cols, res, len_x, len_y, pos_x, pos_y);
}

int artp_cost_query_dev(artp_ctx* c, const float* edges, size_t b, float* cost) {
  if (!c || (b && (!edges || !cost))) return ARTP_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  if (!c->have_weights) {
    c->last_error = "artp_cost_load_weights has not been called";
    return ARTP_ERR_NO_WEIGHTS;
  }
  if (!c->have_features) {
    c->last_error = "artp_cost_update_map has not been called";
    return ARTP_ERR_NO_MAP;
  }
  if (b == 0) return ARTP_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  // up to 2^16 edges (a roadmap update's query): four lanes per edge; above that a lane per edge fills the GPU.  Both
  // kernels accumulate every unit in the same order: the same bits
  if (c->fc_mfma) {
    // the MLP as MFMA tiles (cost_kernels.h fc_cost_mfma_kernel): 64 edges per wavefront, 256 per workgroup
    size_t blocks = (b + 255) / 256;
    if (blocks > (size_t)c->n_cus * 3) blocks = (size_t)c->n_cus * 3;  // three workgroups per CU fit (50 KB of LDS each)
    hipLaunchKernelGGL(fc_cost_mfma_kernel, dim3((unsigned)blocks), dim3(256), 0, c->stream, edges, b,
                       (const half_t*)c->d_feat, c->cost_geom, (const char*)c->d_fc_mfma, cost);
  } else if (b <= (1u << 16))
    hipLaunchKernelGGL(fc_cost_split_kernel, dim3((unsigned)((b + FC_SPLIT_EDGES - 1) / FC_SPLIT_EDGES)), dim3(256), 0, c->stream,
                       edges, b, (const half_t*)c->d_feat, c->cost_geom, (const float*)c->d_fc, cost);
  else
    hipLaunchKernelGGL(fc_cost_kernel, dim3((unsigned)((b + 255) / 256)), dim3(256), 0, c->stream, edges, b,
                       (const half_t*)c->d_feat, c->cost_geom, (const float*)c->d_fc, cost);
  HIP_TRY(c, hipGetLastError());
  return ARTP_OK;
}

int artp_cost_set_external_query(artp_ctx* c, artp_cost_query_fn fn, void* user) {
  if (!c) return ARTP_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  c->ext_cost_fn = fn;
  c->ext_cost_user = fn ? user : nullptr;
  return ARTP_OK;
}

// The roadmap's learned-cost batch (device edge matrix in, device costs out, on the context's stream): the caller's
// MotionCostFunc when one is installed -- the edge matrix goes to the host, through the function and back, exactly the
// [B x 6] -> [B x 3] call of PRMMotionCostMaintainer::updateEdges (prm_motion_cost.cpp:27-73) -- else the device network.
int roadmap_cost_query_dev(artp_ctx* c, const float* d_edges, size_t b, float* d_cost) {
  if (!c->ext_cost_fn) return artp_cost_query_dev(c, d_edges, b, d_cost);
  if (b == 0) return ARTP_OK;
  std::vector<float> em(b * 6), c3(b * 3, 0.0f);
  HIP_TRY(c, hipMemcpyAsync(em.data(), d_edges, b * 6 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (c->ext_cost_fn(c->ext_cost_user, em.data(), b, c3.data()) != 0) {
    c->last_error = "Motion cost call failed";   // motion_cost_objective.cpp:81
    return ARTP_ERR_COST_FUNC;
  }
  HIP_TRY(c, hipMemcpyAsync(d_cost, c3.data(), b * 3 * sizeof(float), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));   // c3 leaves scope
  return ARTP_OK;
}

// diagnostics: the feature-map cell CostQuery.__call__ gathers for every edge's start (cost_query.py:54-55), from the
// very device function the cost kernels use
int artp_cost_debug_query_cells(artp_ctx* c, const float* edges, size_t b, int32_t* rows_out, int32_t* cols_out) {
  if (!c || (b && (!edges || !rows_out || !cols_out))) return ARTP_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  if (!c->have_features) return ARTP_ERR_NO_MAP;
  if (b == 0) return ARTP_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  int rc = ensure_tmp(c, 0, b * 6 * sizeof(float));
  if (rc) return rc;
  rc = ensure_tmp(c, 1, 2 * b * sizeof(int));
  if (rc) return rc;
  int* d_rc = static_cast<int*>(c->tmp[1]);
  HIP_TRY(c, hipMemcpyAsync(c->tmp[0], edges, b * 6 * sizeof(float), hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(cost_query_cells_kernel, dim3((unsigned)((b + 255) / 256)), dim3(256), 0, c->stream,
                     static_cast<const float*>(c->tmp[0]), b, c->cost_geom, d_rc, d_rc + b);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipMemcpyAsync(rows_out, d_rc, b * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipMemcpyAsync(cols_out, d_rc + b, b * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return ARTP_OK;
}

int artp_cost_query(artp_ctx* c, const float* edges, size_t b, float* cost) {
  if (!c || (b && (!edges || !cost))) return ARTP_ERR_INVALID_ARG;
  if (b == 0) return ARTP_OK;
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  HIP_TRY(c, hipSetDevice(c->device));
  int rc = ensure_tmp(c, 0, b * 6 * sizeof(float));
  if (rc) return rc;
  rc = ensure_tmp(c, 1, b * 3 * sizeof(float));
  if (rc) return rc;
  HIP_TRY(c, hipMemcpyAsync(c->tmp[0], edges, b * 6 * sizeof(float), hipMemcpyHostToDevice, c->stream));
  rc = artp_cost_query_dev(c, static_cast<const float*>(c->tmp[0]), b, static_cast<float*>(c->tmp[1]));
  if (rc) return rc;
  HIP_TRY(c, hipMemcpyAsync(cost, c->tmp[1], b * 3 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return ARTP_OK;
}

// Copy the feature map out (tests / diagnostics): NHWC fp16 -> float [F][F][48]
int artp_cost_get_features(artp_ctx* c, float* out, int* fh, int* fw) {
  if (!c || !fh || !fw) return ARTP_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  if (!c->have_features) return ARTP_ERR_NO_MAP;
  *fh = c->feat_h;
  *fw = c->feat_w;
  if (!out) return ARTP_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  const size_t n = (size_t)c->feat_h * c->feat_w * 48;
  std::vector<uint16_t> tmp(n);
  HIP_TRY(c, hipMemcpy(tmp.data(), c->d_feat, n * 2, hipMemcpyDeviceToHost));
  for (size_t i = 0; i < n; ++i) {
    _Float16 h;
    std::memcpy(&h, &tmp[i], 2);
    out[i] = (float)h;
  }
  return ARTP_OK;
}

}  // extern "C"

#include "roadmap.h"
#include "preprocess.h"
#include "group.h"

#ifdef ARTP_STAGE_TIMING
extern "C" int artp_debug_few_trace(unsigned long long* out40) {
  return hipMemcpyFromSymbol(out40, HIP_SYMBOL(artp::g_few_trace), 40 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
extern "C" int artp_debug_pool_trace(unsigned long long* out16) {
  return hipMemcpyFromSymbol(out16, HIP_SYMBOL(artp::g_pool_trace), 16 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
extern "C" int artp_debug_stage_cycles(unsigned long long* out20, int reset) {
  if (out20 && hipMemcpyFromSymbol(out20, HIP_SYMBOL(artp::g_stage_cycles), 20 * sizeof(unsigned long long)) != hipSuccess)
    return -1;
  if (out20 && reset >= 3 &&
      hipMemcpyFromSymbol(out20, HIP_SYMBOL(artp::g_feet_cycles), 4 * sizeof(unsigned long long)) != hipSuccess)
    return -1;
  if (reset) {
    unsigned long long z4[4] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(artp::g_feet_cycles), z4, sizeof(z4)) != hipSuccess) return -1;
  }
  if (reset == 6) {  // conv_ksplit_kernel's per-workgroup records: out20[0 .. 6143] (1024 x {HW_ID, XCC_ID, 4 x s_memrealtime})
    return out20 && hipMemcpyFromSymbol(out20, HIP_SYMBOL(artp::g_ks_trace), 1024 * 6 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
  }
  if (reset == 5) {  // the strip-walking 15 x 15 kernel: out20[0..47] = cycles per (wavefront, phase) summed over the workgroups
    static unsigned long long all[256 * 12 * 4];
    if (hipMemcpyFromSymbol(all, HIP_SYMBOL(artp::g_kwalk_cycles), sizeof(all)) != hipSuccess) return -1;
    if (out20)
      for (int k = 0; k < 48; ++k) {
        out20[k] = 0;
        for (int b = 0; b < 256; ++b) out20[k] += all[b * 48 + k];
      }
    std::memset(all, 0, sizeof(all));
    return hipMemcpyToSymbol(HIP_SYMBOL(artp::g_kwalk_cycles), all, sizeof(all)) == hipSuccess ? 0 : -1;
  }
  if (reset == 4) {  // the feature extractor's phase counters (read + reset)
    if (out20 && hipMemcpyFromSymbol(out20, HIP_SYMBOL(artp::g_cnn_cycles), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    unsigned long long z16[16] = {};
    return hipMemcpyToSymbol(HIP_SYMBOL(artp::g_cnn_cycles), z16, sizeof(z16)) == hipSuccess ? 0 : -1;
  }
  if (out20 && reset == 2 &&
      hipMemcpyFromSymbol(out20, HIP_SYMBOL(artp::g_classify_cycles), 16 * sizeof(unsigned long long)) != hipSuccess)
    return -1;
  if (reset) {
    unsigned long long z16[16] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(artp::g_classify_cycles), z16, sizeof(z16)) != hipSuccess) return -1;
    unsigned long long z[20] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(artp::g_stage_cycles), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
#endif
