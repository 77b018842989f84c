// kernels.h -- gfx950 kernels of the validity / sampling / edge hot path.
//
//   check_boxes_kernel      R3  HeightMapBoxChecker::checkCollision, one wavefront per dPose
//   validate_states_kernel  R1+R2 StateValidityChecker::isValid, one wavefront per state
//   sample_states_kernel    R6  SE3FromSE2Sampler::sampleUniform, one lane per sample
//   motion_plan_kernel / expanded_validate_kernel
//                           R7  DiscreteMotionValidator::checkMotion and the 0.5 m edge
//                               interpolation of PRMMotionCost::addValidMilestone: every interior
//                               state becomes one wave-task of the same validity code
#pragma once

#include "box_check.h"

namespace artp {

struct RobotDev {  // float views of the artp_params numbers, converted once on the host
  float torso[3];  // HeightMapBoxChecker(torso.length, torso.width, torso.height) ctor args (float)
  float foot[3];   // reach.x, reach.y, reach.z
  float torso_off[3];  // (torso.offset.x, torso.offset.y, torso.offset.z - feet.offset.z) as float
  float feet_off_x, feet_off_y;
  int unknown_space_untraversable;
  double reach_z;
  double max_pitch_pert, max_roll_pert;
};

struct MapGeom {  // grid_map geometry (doubles, as grid_map stores them)
  double len_x, len_y, pos_x, pos_y, res;
  int rows, cols;
};

struct SamplerDev {
  int from_distribution;          // Params::sampler.sample_from_distribution
  const float* cum_prob;          // col-major rows x cols (grid_map storage, as uploaded)
  const float* cum_prob_rowwise;  // rows
  const float* elevation;
  const float* normal_x;
  const float* normal_y;
  const float* normal_z;
  const float* plane_fit_std_dev;
  // Derived at upload (sampler_pack_kernel) for the batch sampler, which is bound by the number of L2 requests
  // per sample, not by arithmetic: the per-row CDF contiguous (row-major), one pivot per 16 columns, and the five
  // per-cell numbers of a sample in one 32-byte record -- 4 requests per sample instead of 14.
  const float* cum_prob_t;        // row-major, rows x pitch floats (pitch = cols rounded up to 16: aligned groups)
  const float* pivots;            // rows x ppitch: cum_prob(row, 16 j + 15) for j < npiv - 1, +inf behind them
  const float4* cells;            // 2 float4 per cell (index row + col * rows): {elev, nx, ny, nz}, {std, -, -, -}
  int npiv;
  int pitch;
  int ppitch;                     // floats per pivot row (npiv rounded up to 4: aligned 16-byte loads)
};

// one lane per cell: transposed CDF, pivots, packed cell records
__global__ void __launch_bounds__(256)
sampler_pack_kernel(SamplerDev sm, int rows, int cols, float* __restrict__ cum_prob_t, float* __restrict__ pivots,
                    float4* __restrict__ cells) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * cols) return;
  const int row = i % rows, col = i / rows;
  const float v = sm.cum_prob[i];
  cum_prob_t[(size_t)row * sm.pitch + col] = v;
  // the last pivot of a row is never looked at ("first of [0, cols-2] that exceeds u, else cols-1"): it and the
  // padding behind it hold +inf, so "number of pivots that do not exceed u" is the group index
  if (col == cols - 1) {
    for (int j = col >> 4; j < sm.ppitch; ++j) pivots[(size_t)row * sm.ppitch + j] = INFINITY;
  } else if ((col & 15) == 15) {
    pivots[(size_t)row * sm.ppitch + (col >> 4)] = v;
  }
  cells[2 * (size_t)i] = make_float4(sm.elevation[i], sm.normal_x[i], sm.normal_y[i], sm.normal_z[i]);
  cells[2 * (size_t)i + 1] = make_float4(sm.plane_fit_std_dev[i], 0.0f, 0.0f, 0.0f);
}

#define ARTP_WAVES_PER_BLOCK 2

// ---- per-wave LDS carve --------------------------------------------------------------------------
struct ScratchCaps {  // sized on the host from the box diagonals / sample spacing
  int cand;   // bytes for the corner-candidate arrays (36 B per candidate; multiple of 16)
  int verts;  // heights tile, floats (multiple of 4)
  int tris;   // kept-triangle list, u16 (multiple of 8); 0 = stage has no list
  int tab;    // hash table entries, u32 (power of two); 0 = stage has no table
};

__host__ __device__ __forceinline__ size_t scratch_bytes_per_wave(const ScratchCaps& c) {
  return (size_t)c.cand + (size_t)c.verts * 4 + (size_t)c.tab * 4 + (size_t)c.tris * 2;
}

__device__ __forceinline__ WaveScratch carve_scratch(char* smem, int wave_in_block, const ScratchCaps& c) {
  char* base = smem + scratch_bytes_per_wave(c) * wave_in_block;
  WaveScratch s;
  s.cand = reinterpret_cast<float*>(base);
  base += c.cand;
  s.h = reinterpret_cast<float*>(base);
  s.tab = reinterpret_cast<unsigned*>(base + (size_t)c.verts * 4);
  s.tri = reinterpret_cast<unsigned short*>(base + (size_t)c.verts * 4 + (size_t)c.tab * 4);
  s.cap_verts = c.verts;
  s.cap_tris = c.tris;
  s.tab_size = c.tab;
  return s;
}

// grid_map isInside (checkIfPositionWithinMap): t = -((p - c) - L/2), 0 <= t < L, in double.
__device__ __forceinline__ bool map_is_inside(const MapGeom& g, double px, double py) {
  const double tx = -((px - g.pos_x) - 0.5 * g.len_x);
  const double ty = -((py - g.pos_y) - 0.5 * g.len_y);
  return tx >= 0.0 && ty >= 0.0 && tx < g.len_x && ty < g.len_y;
}

// ---- R3: HeightMapBoxChecker::checkCollision, one wavefront per dPose -----------------------------
template <int WAVES>
__global__ void __launch_bounds__(64 * WAVES)
check_boxes_kernel(FieldDev f, float sx, float sy, float sz, const float* __restrict__ poses,
                   size_t n, uint8_t* __restrict__ hit, uint8_t* __restrict__ exit_codes,
                   ScratchCaps caps, int* __restrict__ error_flag) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave_in_block = threadIdx.x >> 6;
  const WaveScratch s = carve_scratch(smem, wave_in_block, caps);
  const size_t wave0 = (size_t)blockIdx.x * WAVES + wave_in_block;
  const size_t stride = (size_t)gridDim.x * WAVES;
  for (size_t i = wave0; i < n; i += stride) {
    float pose[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) pose[k] = poses[16 * i + k];
    BoxHF b;
    setup_box(f, pose, sx, sy, sz, b);
    int ec;
    const int r = wave_check_box(f, b, s, lane, &ec);
    if (r < 0) {
      if (lane == 0) atomicExch(error_flag, 1);
    }
    if (lane == 0) {
      hit[i] = (uint8_t)(r > 0);
      if (exit_codes) exit_codes[i] = (uint8_t)(ec < 0 ? 255 : ec);
    }
    wave_lds_sync();
  }
}

// ---- R1 + R2 ---------------------------------------------------------------------------------------
// Pose3FromSE3 (art_planner/include/art_planner/utils.h:25-38): Eigen::Quaternionf(w,x,y,z)
// .toRotationMatrix(); R row-major 3x3.
// Eigen::Quaternionf(w, x, y, z).toRotationMatrix(), row-major
__device__ __forceinline__ void rot_from_quat(float x, float y, float z, float w, float R[9]) {
  const float tx = 2.0f * x, ty = 2.0f * y, tz = 2.0f * z;
  const float twx = tx * w, twy = ty * w, twz = tz * w;
  const float txx = tx * x, txy = ty * x, txz = tz * x;
  const float tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1.0f - (tyy + tzz);
  R[1] = txy - twz;
  R[2] = txz + twy;
  R[3] = txy + twz;
  R[4] = 1.0f - (txx + tzz);
  R[5] = tyz - twx;
  R[6] = txz - twy;
  R[7] = tyz + twx;
  R[8] = 1.0f - (txx + tyy);
}

__device__ __forceinline__ void pose3_from_se3(const double* se3, float t[3], float R[9]) {
  t[0] = (float)se3[0];
  t[1] = (float)se3[1];
  t[2] = (float)se3[2];
  rot_from_quat((float)se3[3], (float)se3[4], (float)se3[5], (float)se3[6], R);
}

// What the five boxes of a state share, computed ONCE per state (by the sampler, or by pose_rec_kernel for
// caller-provided states) instead of once per (state, box) lane of the classify stage: the float pose of
// Pose3FromSE3 (translation + quaternion; the rotation matrix is 27 flops away) and the box rotation in the
// field frame -- dxOrthogonalizeR (5 IEEE divisions, 2 square roots) + Rt^T R, the same for every box of the
// state and for both layers (the field rotation is a constant of HeightMapBoxChecker, height_map_box_checker.cpp:22).
struct __attribute__((aligned(16))) PoseRec {  // 64 bytes = one cache line per state
  float t[3];
  float q[4];   // x y z w
  float bR[9];  // box_rotation_in_field
};

__device__ __forceinline__ void make_pose_rec(const FieldDev& f, const double* se3, float4 out[4]) {
  float t[3], R[9], bR[9];
  pose3_from_se3(se3, t, R);
  box_rotation_in_field(f, R, bR);
  out[0] = make_float4(t[0], t[1], t[2], (float)se3[3]);
  out[1] = make_float4((float)se3[4], (float)se3[5], (float)se3[6], bR[0]);
  out[2] = make_float4(bR[1], bR[2], bR[3], bR[4]);
  out[3] = make_float4(bR[5], bR[6], bR[7], bR[8]);
}

// A wavefront's 64 PoseRecs through LDS: lane l parks the four 16-byte chunks of its record, then the wavefront
// writes the 4 KB block as four fully coalesced 1 KB rows.  Chunk k of lane l sits at slot 4 l + ((k + (l >> 2)) & 3):
// un-skewed, lanes l and l + 4 of one 8-lane ds_write_b128 group would hit the same banks.
__device__ __forceinline__ int rec_stage_slot(int l, int k) { return 4 * l + ((k + (l >> 2)) & 3); }
__device__ __forceinline__ void stage_pose_rec(float4* rw, int lane, const float4 r[4]) {
#pragma unroll
  for (int k = 0; k < 4; ++k) rw[rec_stage_slot(lane, k)] = r[k];
}
// live = records of this wavefront that exist (64 except at the tail); call between two wave_lds_sync()
__device__ __forceinline__ void flush_pose_recs(const float4* rw, int lane, size_t live, PoseRec* __restrict__ dst_recs) {
  float4* dst = reinterpret_cast<float4*>(dst_recs);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int j = k * 64 + lane;  // chunk j of the block = chunk (j & 3) of lane j >> 2
    // streamed once, read once by the next kernel: non-temporal, so that 0.27 GB of records per batch do not push the
    // sampler's 6 MB of tables out of the L2s
    typedef float native_f4 __attribute__((ext_vector_type(4)));
    if ((size_t)j < live * 4)
      __builtin_nontemporal_store(reinterpret_cast<const native_f4*>(rw)[rec_stage_slot(j >> 2, j & 3)],
                                  reinterpret_cast<native_f4*>(dst) + j);
  }
}

__global__ void __launch_bounds__(256)
pose_rec_kernel(FieldDev f, const double* __restrict__ se3, size_t n, PoseRec* __restrict__ recs) {
  // both directions through LDS: a wavefront's 64 states are seven coalesced 512-byte rows on the way in (a lane
  // reading its own 56 bytes touches 28 lines per load instruction), its 64 PoseRecs four 1 KB rows on the way out
  __shared__ double stage[4][64 * 8];
  const int lane = threadIdx.x & 63;
  double* sw = stage[threadIdx.x >> 6];
  const size_t i0 = (size_t)blockIdx.x * blockDim.x + (threadIdx.x - lane);  // wave-uniform
  if (i0 >= n) return;
  const size_t live = n - i0 < 64 ? n - i0 : 64;
  const double* in = se3 + 7 * i0;
  // seven loads in flight, then seven LDS stores: a load under a condition becomes a branch that hipcc waits for before the
  // next one (round 5: seven serial trips to memory per wavefront) -- so the loads are unconditional, from a clamped index
  double v7[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) v7[k] = in[(size_t)(k * 64 + lane) < live * 7 ? k * 64 + lane : 0];
#pragma unroll
  for (int k = 0; k < 7; ++k)
    if ((size_t)(k * 64 + lane) < live * 7) sw[k * 64 + lane] = v7[k];
  wave_lds_sync();
  float4 r[4];
  const bool have = (size_t)lane < live;
  if (have) {
    double st[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) st[j] = sw[lane * 7 + j];
    make_pose_rec(f, st, r);
  }
  wave_lds_sync();
  float4* rw = reinterpret_cast<float4*>(sw);
  if (have) stage_pose_rec(rw, lane, r);
  wave_lds_sync();
  flush_pose_recs(rw, lane, live, recs + i0);
}

// One full StateValidityChecker::isValid for the state held (wave-uniformly) in se3[7].
// detail6 (global memory, may be nullptr) receives the per-box exit codes.  Returns 0/1, or -1 on
// scratch overflow.
__device__ __forceinline__ int wave_state_valid(const FieldDev& fb, const FieldDev& ff,
                                                const MapGeom& g, const RobotDev& rb,
                                                const double* se3, const WaveScratch& s, int lane,
                                                int8_t* detail6) {
  float t[3], R[9];
  pose3_from_se3(se3, t, R);
  if (detail6 && lane == 0) {
#pragma unroll
    for (int k = 0; k < 5; ++k) detail6[k] = -2;
    detail6[5] = 0;
  }
  int valid = 1;
  int err = 0;
  // box 0 = torso against the body layer, boxes 1..4 = feet against the masked layer, in the
  // reference order (+,+),(+,-),(-,+),(-,-) (validity_checker_feet.cpp:64-68); the loop stops at the
  // first failing box like the reference's short-circuit.
  for (int k = 0; k < 5 && valid; ++k) {
    const bool body = (k == 0);
    const float ox = body ? rb.torso_off[0] : ((k <= 2) ? rb.feet_off_x : -rb.feet_off_x);
    const float oy = body ? rb.torso_off[1] : ((k & 1) ? rb.feet_off_y : -rb.feet_off_y);
    const float oz = body ? rb.torso_off[2] : 0.0f;
    // pose * Pose3FromXYZ(o): Eigen affine product, translation = R*o + t with the 3-term dot
    // summed as x0 + (x1 + x2) (Eigen's unrolled redux).
    float pose[16];
    pose[0] = (R[0] * ox + (R[1] * oy + R[2] * oz)) + t[0];
    pose[1] = (R[3] * ox + (R[4] * oy + R[5] * oz)) + t[1];
    pose[2] = (R[6] * ox + (R[7] * oy + R[8] * oz)) + t[2];
    pose[3] = 0.0f;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      pose[4 + 4 * r + 0] = R[3 * r + 0];
      pose[4 + 4 * r + 1] = R[3 * r + 1];
      pose[4 + 4 * r + 2] = R[3 * r + 2];
      pose[4 + 4 * r + 3] = 0.0f;
    }
    const bool inside = map_is_inside(g, (double)pose[0], (double)pose[1]);
    int ok;
    int ec = -1;
    if (!inside) {
      // body: outside -> valid (validity_checker_body.cpp:29-32);
      // feet: outside -> !unknown_space_untraversable (validity_checker_feet.cpp:34-37)
      ok = body ? 1 : !rb.unknown_space_untraversable;
    } else {
      const FieldDev& fk = body ? fb : ff;
      BoxHF b;
      setup_box(fk, pose, body ? rb.torso[0] : rb.foot[0], body ? rb.torso[1] : rb.foot[1],
                body ? rb.torso[2] : rb.foot[2], b);
      int r = wave_check_box(fk, b, s, lane, &ec);
      wave_lds_sync();
      if (r < 0) {
        err = 1;
        r = 0;
      }
      ok = body ? !r : r;
    }
    if (detail6 && lane == 0) detail6[k] = (int8_t)ec;
    valid = valid && ok;
  }
  return err ? -1 : valid;
}

// v1: one wavefront per state, boxes in the reference's order with its short-circuit.  Kept for the
// `detail` output (per-box exit codes exactly as the reference would evaluate them); the throughput
// path is the classify -> resolve -> plane-stage pipeline below.
template <int WAVES>
__global__ void __launch_bounds__(64 * WAVES)
validate_states_kernel(FieldDev fb, FieldDev ff, MapGeom g, RobotDev rb,
                       const double* __restrict__ se3, size_t n, uint8_t* __restrict__ valid,
                       int8_t* __restrict__ detail, ScratchCaps caps,
                       int* __restrict__ error_flag, unsigned long long* __restrict__ n_valid) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave_in_block = threadIdx.x >> 6;
  const WaveScratch s = carve_scratch(smem, wave_in_block, caps);
  const size_t wave0 = (size_t)blockIdx.x * WAVES + wave_in_block;
  const size_t stride = (size_t)gridDim.x * WAVES;
  unsigned long long local_valid = 0;
  for (size_t i = wave0; i < n; i += stride) {
    double st[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) st[k] = se3[7 * i + k];
    const int v = wave_state_valid(fb, ff, g, rb, st, s, lane, detail ? detail + 6 * i : nullptr);
    if (v < 0 && lane == 0) atomicExch(error_flag, 1);
    if (lane == 0) {
      valid[i] = (uint8_t)(v > 0);
      local_valid += (v > 0);
    }
  }
  if (n_valid && lane == 0 && local_valid) atomicAdd(n_valid, local_valid);
}

// Latency path of a handful of states (the per-state isValid() of the host mirror: one OMPL call = one state).
// One workgroup per state, five wavefronts = the five boxes side by side (torso against the body layer, four
// feet against the masked layer), each with its own LDS scratch, so the call costs the slowest box instead of
// their sum.  The label is the AND over the boxes (the reference's short-circuit only skips work).  States and
// labels may live in mapped host memory: one launch, no copies.
// One box (k = 0 torso, 1..4 feet in the reference order) of one state on one wavefront, with the wavefront's own LDS
// scratch: 1 / 0 = this box passes / fails the state, -1 = the window exceeded the scratch.
__device__ __forceinline__ int few_box_ok(const FieldDev& fb, const FieldDev& ff, const MapGeom& g, const RobotDev& rb,
                                          const double* st, int k, const WaveScratch& s, int lane) {
  const bool body = (k == 0);
  float t[3], R[9];
  pose3_from_se3(st, t, R);
  const float ox = body ? rb.torso_off[0] : ((k <= 2) ? rb.feet_off_x : -rb.feet_off_x);
  const float oy = body ? rb.torso_off[1] : ((k & 1) ? rb.feet_off_y : -rb.feet_off_y);
  const float oz = body ? rb.torso_off[2] : 0.0f;
  float pose[16];  // pose * Pose3FromXYZ(o), Eigen's x0 + (x1 + x2) dot (see wave_state_valid)
  pose[0] = (R[0] * ox + (R[1] * oy + R[2] * oz)) + t[0];
  pose[1] = (R[3] * ox + (R[4] * oy + R[5] * oz)) + t[1];
  pose[2] = (R[6] * ox + (R[7] * oy + R[8] * oz)) + t[2];
  pose[3] = 0.0f;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    pose[4 + 4 * r + 0] = R[3 * r + 0];
    pose[4 + 4 * r + 1] = R[3 * r + 1];
    pose[4 + 4 * r + 2] = R[3 * r + 2];
    pose[4 + 4 * r + 3] = 0.0f;
  }
  if (!map_is_inside(g, (double)pose[0], (double)pose[1]))
    return body ? 1 : !rb.unknown_space_untraversable;  // validity_checker_body.cpp:29-32, _feet.cpp:34-37
  BoxHF b;
  int ec, r;
  if (body) {
    setup_box(fb, pose, rb.torso[0], rb.torso[1], rb.torso[2], b);
    r = wave_check_box(fb, b, s, lane, &ec);
  } else {
    setup_box(ff, pose, rb.foot[0], rb.foot[1], rb.foot[2], b);
    r = wave_check_box(ff, b, s, lane, &ec);
  }
  return (r < 0) ? -1 : (body ? !r : r);
}

__global__ void __launch_bounds__(320)
validate_few_kernel(FieldDev fb, FieldDev ff, MapGeom g, RobotDev rb, const double* __restrict__ se3, size_t n,
                    volatile uint8_t* valid, ScratchCaps caps_torso, ScratchCaps caps_foot,
                    int* __restrict__ error_flag, unsigned done_tag) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int box_ok[5];
  const int lane = threadIdx.x & 63;
  const int k = threadIdx.x >> 6;  // box index, wave-uniform
  const size_t i = blockIdx.x;
  if (i >= n) return;
  WaveScratch s;
  if (k == 0) {
    s = carve_scratch(smem, 0, caps_torso);
  } else {
    s = carve_scratch(smem + scratch_bytes_per_wave(caps_torso), k - 1, caps_foot);
  }
  double st[7];
#pragma unroll
  for (int j = 0; j < 7; ++j) st[j] = se3[7 * i + j];
  const int ok = few_box_ok(fb, ff, g, rb, st, k, s, lane);
  if (ok < 0 && lane == 0 && !done_tag) atomicExch(error_flag, 1);
  if (lane == 0) box_ok[k] = ok;
  __syncthreads();
  if (threadIdx.x == 0) {
    bool err = false, v = true;
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      err = err || box_ok[q] < 0;
      v = v && box_ok[q] > 0;
    }
    // done_tag != 0 (host call through mapped memory): the host polls for the tag bit instead of synchronising
    // the stream; bit 1 = a window exceeded the LDS scratch (ARTP_ERR_CAPACITY)
    valid[i] = (uint8_t)((v && !err ? 1u : 0u) | (done_tag && err ? 2u : 0u) | done_tag);
    if (done_tag) __threadfence_system();
  }
}

// ---- persistent latency service for isValid (round 6; opt-in: artp_set_persistent_latency) -------------------------------
// validate_few_kernel costs ~8 us of kernel and ~8 us of launch + completion per call.  The service removes the launch: ONE
// resident workgroup (five wavefronts = the five boxes of a state) polls a mailbox in mapped host memory; the host writes the
// states and bumps req_seq, the workgroup validates them (few_box_ok: the very function of the launch-per-call path) and
// answers with the labels and resp_seq.  It never outlives its usefulness: it leaves when told to (quit), after
// ARTP_SVC_IDLE_TICKS (200 us) without a request, or after ARTP_SVC_LIFE_TICKS (2 s) whatever happens (s_memrealtime ticks of 10 ns) --
// a host that died or a protocol bug costs a bounded wait, never a hung GPU.  The host restarts it on demand and ALWAYS
// after a map write (the field arguments are captured at launch; artp_map_version).
struct SvcMailbox {   // mapped (coherent) host memory
  // request: ONE 64-byte line the workgroup polls with one 16-lane load -- the host writes the state first and the sequence
  // number last (x86 stores become visible in order), the device reads the line again once it has seen the new number
  volatile uint32_t req_seq;
  volatile uint32_t n_quit;        // bits 0-7 number of states (1 or 2), bit 8 = leave
  double state0[7];
  double state1[7];                // second state of a two-state call: read AFTER the new sequence number was seen
  uint32_t pad0[34];               // the response starts on its own 128-byte line (offset 256)
  // response: ONE 32-bit word {sequence number's low 24 bits << 8 | labels} (label of state i in bits 2i (valid) and 2i + 1
  // (LDS scratch overflow)): a single dword store certainly arrives in one piece, number and labels together
  volatile uint32_t resp;
  uint32_t pad_resp;
  volatile uint32_t running;       // 1 while the kernel is resident
  volatile uint32_t served;        // requests answered by this incarnation
  uint32_t pad1[28];
};
static_assert(sizeof(SvcMailbox) == 384, "request line, second state, response line");
// 200 us without a request: a burst of isValid() calls from a host loop arrives every 10-20 us; once it ends the workgroup
// is gone before anything else can trip over it (hipFree / hipDeviceSynchronize wait for EVERY stream of the device: a
// resident kernel would stall them for as long as it stays)
#define ARTP_SVC_IDLE_TICKS 20000ull
#define ARTP_SVC_LIFE_TICKS 200000000ull   // 2 s in all
__global__ void __launch_bounds__(320)
validate_service_kernel(FieldDev fb, FieldDev ff, MapGeom g, RobotDev rb, SvcMailbox* mb, uint32_t last_seq,
                        ScratchCaps caps_torso, ScratchCaps caps_foot) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int box_ok[5];
  __shared__ uint32_t s_line[16 + 14];   // the request line, then the second state
  __shared__ uint32_t s_exit;
  const int lane = threadIdx.x & 63;
  const int k = threadIdx.x >> 6;
  WaveScratch s;
  if (k == 0) {
    s = carve_scratch(smem, 0, caps_torso);
  } else {
    s = carve_scratch(smem + scratch_bytes_per_wave(caps_torso), k - 1, caps_foot);
  }
  const unsigned long long t_start = wall_clock64();
  unsigned long long t_idle = t_start;
  uint32_t served = 0;
  const volatile uint32_t* line = reinterpret_cast<const volatile uint32_t*>(mb);
  for (;;) {
    if (k == 0) {   // wavefront 0 polls: lanes 0..15 fetch the request line, one PCIe read per poll
      uint32_t leave = 0, w = 0, seq;
      for (;;) {
        if (lane < 16) w = line[lane];
        seq = (uint32_t)__builtin_amdgcn_readfirstlane((int)w);
        if (seq != last_seq) break;
        const uint32_t nq = (uint32_t)__builtin_amdgcn_readlane((int)w, 1);
        const unsigned long long now = wall_clock64();   // s_memrealtime: a scalar, the same for every lane
        if ((nq & 0x100u) || now - t_idle > ARTP_SVC_IDLE_TICKS || now - t_start > ARTP_SVC_LIFE_TICKS) {
          leave = 1;
          break;
        }
        __builtin_amdgcn_s_sleep(4);
      }
      // The poll's 64 bytes may reach the host as more than one read (32-byte sectors), so the sector with the sequence number
      // could be newer than the one with the state's tail: fetch the line ONCE MORE now that the number is known to be there
      // (the host wrote the state before it).  A precaution worth one PCIe round trip, not a measured failure.
      if (!leave && lane < 16) w = line[lane];
      if (lane < 16) s_line[lane] = w;
      if (!leave && ((uint32_t)__builtin_amdgcn_readlane((int)w, 1) & 0xffu) > 1u && lane < 14)
        s_line[16 + lane] = reinterpret_cast<const volatile uint32_t*>(mb->state1)[lane];   // after the sequence number
      if (lane == 0) s_exit = leave;
    }
    __syncthreads();
    if (s_exit) {
      if (threadIdx.x == 0) {
        mb->running = 0u;
        __threadfence_system();
      }
      return;
    }
    const uint32_t seq = s_line[0];
    uint32_t n = s_line[1] & 0xffu;
    n = n < 1u ? 1u : (n > 2u ? 2u : n);
    uint32_t bits = 0;   // (meaningful in thread 0)
    for (uint32_t i = 0; i < n; ++i) {
      double st[7];
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        const uint32_t lo = s_line[(i ? 16 : 2) + 2 * j], hi = s_line[(i ? 16 : 2) + 2 * j + 1];
        st[j] = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
      }
      const int ok = few_box_ok(fb, ff, g, rb, st, k, s, lane);
      wave_lds_sync();
      if (lane == 0) box_ok[k] = ok;
      __syncthreads();
      if (threadIdx.x == 0) {
        bool err = false, v = true;
#pragma unroll
        for (int q = 0; q < 5; ++q) {
          err = err || box_ok[q] < 0;
          v = v && box_ok[q] > 0;
        }
        bits |= ((v && !err ? 1u : 0u) | (err ? 2u : 0u)) << (2 * i);
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      mb->resp = (seq << 8) | bits;   // ONE dword: number and labels arrive together
      __threadfence_system();
      ++served;
      mb->served = served;
    }
    last_seq = seq;               // every thread: the pollers are ALL lanes of wavefront 0, and the loop's exit conditions must be
    t_idle = wall_clock64();      // wave-uniform (a lane with a stale idle clock left the poll loop alone, with the PREVIOUS request's words)
    __syncthreads();
  }
}

// ---- R6 ------------------------------------------------------------------------------------------
// Counter-based uniform01 (replaces ompl::RNG::uniform01): splitmix64 finaliser over
// (seed, index, k) -- integer-exact, identical on host and device.
ARTP_HD double uniform01(uint64_t seed, uint64_t index, unsigned k) {
  uint64_t x = seed + 0x9E3779B97F4A7C15ULL * (index * 8u + (uint64_t)k + 1u);
  x ^= x >> 30;
  x *= 0xBF58476D1CE4E5B9ULL;
  x ^= x >> 27;
  x *= 0x94D049BB133111EBULL;
  x ^= x >> 31;
  x += seed;
  x ^= x >> 30;
  x *= 0xBF58476D1CE4E5B9ULL;
  x ^= x >> 27;
  x *= 0x94D049BB133111EBULL;
  x ^= x >> 31;
  return (double)(x >> 11) * (1.0 / 9007199254740992.0);
}

// sin and cos of a half Euler angle, |x| <= ~3 (the sampler's yaw/2, roll/2, pitch/2): one Cody-Waite step to
// the nearest multiple of pi/2 (two-part constant, exact for the handful of quadrants that can occur), then the
// classic minimax kernels on [-pi/4, pi/4] -- within one ulp of libm, without the large-argument reduction of the
// general sincos (which alone costs this kernel 40 registers).
__device__ __forceinline__ void sincos_half_angle(double x, double* sn, double* cs) {
  const double kf = rint(x * 6.36619772367581382433e-01);  // x * 2/pi
  const int k = (int)kf;
  double r = fma(-kf, 1.57079632673412561417e+00, x);
  r = fma(-kf, 6.07710050650619224932e-11, r);
  const double z = r * r;
  const double ps = fma(z, fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08),
                                                 2.75573137070700676789e-06),
                                          -1.98412698298579493134e-04),
                                   8.33333333332248946124e-03),
                        -1.66666666666666324348e-01);
  const double pc = fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09),
                                                 -2.75573143513906633035e-07),
                                          2.48015872894767294178e-05),
                                   -1.38888888888741095749e-03),
                        4.16666666666666019037e-02);
  const double s0 = fma(r * z, ps, r);
  const double c0 = fma(z * z, pc, fma(-0.5, z, 1.0));
  const bool swap = (k & 1) != 0;
  const double sa = swap ? c0 : s0, ca = swap ? s0 : c0;
  *sn = (k & 2) ? -sa : sa;
  *cs = ((k + 1) & 2) ? -ca : ca;
}

// SE3FromSE2Sampler::sampleUniform (art_planner/src/sampler.cpp:82-131) with
// samplePositionInMapFromDist (:56-78).  The two linear CDF scans become binary searches for the
// same "first index whose cumulative value exceeds u, else the last index".
// row_cdf: the row CDF (cum_prob_rowwise) in LDS or global memory.
template <bool FROM_DIST>
__device__ __forceinline__ void sample_one(const SamplerDev& sm, const MapGeom& g, const RobotDev& rb,
                                           uint64_t seed, uint64_t index, double out[7], const float* row_cdf) {
  double px, py;
  int cell_row = 0, cell_col = 0;
  if (FROM_DIST) {  // samplePositionInMapFromDist (sampler.cpp:56-78)
    const double samp_col = uniform01(seed, index, 0);
    const double samp_row = uniform01(seed, index, 1);
    int lo = 0, hi = g.rows - 1;  // answer in [0, rows-1]
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if ((double)row_cdf[mid] > samp_row) hi = mid; else lo = mid + 1;
    }
    const int row = lo;
    // "first column of [0, cols-2] whose cumulative value exceeds u, else cols-1" in two levels (the CDF of a row
    // is non-decreasing, so the first group of 16 columns whose LAST value exceeds u holds the answer):
    // the pivots of the row (<= 2 cache lines), then the group's 16 values (one line, one request).
    // pivots of the row: all of them in one round trip (<= 8 aligned 16-byte loads, two cache lines for a 400-column
    // map) and a count, instead of a binary search's five dependent requests -- the kernel is bound by its L2
    // requests times their latency (the L1's miss queue), not by arithmetic
    const float* prow = sm.pivots + (size_t)row * sm.ppitch;
    const int nch = sm.ppitch >> 2;
    if (nch <= 8) {
      // (double)p > u  <=>  p > (u rounded DOWN to float): between two neighbouring floats every u compares the same
      const float uf = __double2float_rd(samp_col);
      const float4* p4 = reinterpret_cast<const float4*>(prow);
      lo = 0;
#pragma unroll
      for (int h = 0; h < 2; ++h) {  // two batches of four loads: eight float4 in flight at once spill registers
        float4 pv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (4 * h + k < nch) pv[k] = p4[4 * h + k];
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (4 * h + k < nch)
            lo += (!(pv[k].x > uf) ? 1 : 0) + (!(pv[k].y > uf) ? 1 : 0) + (!(pv[k].z > uf) ? 1 : 0) +
                  (!(pv[k].w > uf) ? 1 : 0);
      }
    } else {
      lo = 0;
      hi = sm.npiv - 1;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if ((double)prow[mid] > samp_col) hi = mid; else lo = mid + 1;
      }
    }
    const int c0 = lo << 4;
    const int c1 = (c0 + 15 < g.cols - 1) ? c0 + 15 : g.cols - 1;
    const float4* grp = reinterpret_cast<const float4*>(sm.cum_prob_t + (size_t)row * sm.pitch + c0);
    int below = 0;  // values of the group that do not exceed u (they come first)
    const float uf_col = __double2float_rd(samp_col);  // (double)v > u  <=>  v > u rounded down to float
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const float4 v = grp[q4];
      below += (c0 + 4 * q4 + 0 <= c1 && !(v.x > uf_col)) ? 1 : 0;
      below += (c0 + 4 * q4 + 1 <= c1 && !(v.y > uf_col)) ? 1 : 0;
      below += (c0 + 4 * q4 + 2 <= c1 && !(v.z > uf_col)) ? 1 : 0;
      below += (c0 + 4 * q4 + 3 <= c1 && !(v.w > uf_col)) ? 1 : 0;
    }
    const int col = (c0 + below < c1) ? c0 + below : c1;
    cell_row = row;
    cell_col = col;
    // grid_map getPosition: (c + (L/2 - res/2)) + res * (-i)
    px = (g.pos_x + (0.5 * g.len_x - 0.5 * g.res)) + g.res * (double)(-row);
    py = (g.pos_y + (0.5 * g.len_y - 0.5 * g.res)) + g.res * (double)(-col);
  } else {
    // samplePositionInMap (sampler.cpp:38-50): uniform over the SE3 bounds pos -+ length
    // (planner.cpp:146-156), three draws per attempt (z unused), until map_->isInside(pos)
    const double lx = g.pos_x - g.len_x, hx = g.pos_x + g.len_x;
    const double ly = g.pos_y - g.len_y, hy = g.pos_y + g.len_y;
    for (unsigned a = 0;; ++a) {
      px = (hx - lx) * uniform01(seed, index, 8 + 3 * a) + lx;
      py = (hy - ly) * uniform01(seed, index, 8 + 3 * a + 1) + ly;
      if (map_is_inside(g, px, py) || a >= 255) break;
    }
  }
  // getIndexOfPosition (sampler.cpp:95).  From the distribution the position is the centre of cell (row, col):
  // the quotient is row + 0.5 up to rounding in the 13th digit, its truncation is row -- no division needed.
  int ri, ci;
  if (FROM_DIST) {
    ri = cell_row;
    ci = cell_col;
  } else {
    ri = (int)(-(((px - 0.5 * g.len_x) - g.pos_x) / g.res));
    ci = (int)(-(((py - 0.5 * g.len_y) - g.pos_y) / g.res));
  }
  const size_t ind = (size_t)ri + (size_t)ci * g.rows;
  const float4 ca = sm.cells[2 * ind], cb = sm.cells[2 * ind + 1];  // one 32-byte record: one request
  double v0 = px, v1 = py, v2 = (double)ca.x;
  const double nwx = (double)ca.y;
  const double nwy = (double)ca.z;
  const double nwz = (double)ca.w;
  const float sd = cb.x;
  const float sd_min = (0.5f < sd) ? 0.5f : sd;  // std::min(std, 0.5f)
  const double u_pert = uniform01(seed, index, 2);
  const double pert = ((1.0 - (-1.0)) * u_pert + (-1.0)) * (double)sd_min * rb.reach_z;
  v0 += nwx * pert;
  v1 += nwy * pert;
  v2 += nwz * pert;
  out[0] = v0;
  out[1] = v1;
  out[2] = v2;
  const double pi = 3.14159265358979323846;
  double rpy0 = pi * (-2.0 * uniform01(seed, index, 3) + 1.0);
  double rpy1 = acos(1.0 - 2.0 * uniform01(seed, index, 4)) - pi / 2.0;
  const double rpy2 = pi * (-2.0 * uniform01(seed, index, 5) + 1.0);
  // sin and cos of yaw/2 are needed twice (the yaw quaternion here, setSO3FromRPY below): one sincos
  double sy2s, sy2c, qw, qz;
  sincos_half_angle(0.5 * rpy2, &sy2s, &sy2c);
  {
    // normal_b = Quaterniond(AngleAxisd(yaw, Z)).inverse() * normal_w (sampler.cpp:120-123).  The quaternion is
    // (cos(yaw/2), 0, 0, sin(yaw/2)): its inverse divides by |q|^2 = 1 +- 1e-16 and the rotation of a vector by it is
    // the plane rotation by -yaw -- written out, four f64 divisions and two thirds of the multiplications fall away
    // (the sampler is bound by its f64 instruction count).  Against the literal form (oracle/artp_oracle.c) the
    // result moves in the 16th digit; the sampler's parity bar is 1e-12 ("same seed" contract).
    qw = sy2c;
    qz = sy2s;
    const double cyaw = qw * qw - qz * qz, syaw = (qw + qw) * qz;
    const double nbx = cyaw * nwx + syaw * nwy;
    const double nby = cyaw * nwy - syaw * nwx;
    const double nbz = nwz;
    rpy0 = -atan2(nby, nbz) + rpy0 * (rb.max_roll_pert * (1.0 / 1.57079632679489661923));
    rpy1 = atan2(nbx, nbz) + rpy1 * (rb.max_pitch_pert * (1.0 / 0.78539816339744830962));
  }
  {  // setSO3FromRPY (utils.h:101-115)
    double cr, cp, sr, sp;
    sincos_half_angle(rpy0 * 0.5, &sr, &cr);
    sincos_half_angle(rpy1 * 0.5, &sp, &cp);
    const double cy = sy2c, sy = sy2s;
    out[6] = cy * cp * cr + sy * sp * sr;
    out[3] = cy * cp * sr - sy * sp * cr;
    out[4] = sy * cp * sr + cy * sp * cr;
    out[5] = sy * cp * cr - cy * sp * sr;
  }
}

// The row CDF of the map in LDS (rows <= ARTP_ROW_CDF_LDS; larger maps search it in global memory).
#define ARTP_ROW_CDF_LDS 2048
__device__ __forceinline__ const float* stage_row_cdf(const SamplerDev& sm, const MapGeom& g, float* lds) {
  if (!sm.from_distribution || g.rows > ARTP_ROW_CDF_LDS) return sm.cum_prob_rowwise;
  for (int i = threadIdx.x; i < g.rows; i += blockDim.x) lds[i] = sm.cum_prob_rowwise[i];
  __syncthreads();
  return lds;
}

#ifndef ARTP_SAMPLER_WAVES
#define ARTP_SAMPLER_WAVES 5
#endif
template <bool FROM_DIST>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(ARTP_SAMPLER_WAVES, ARTP_SAMPLER_WAVES)))
sample_states_kernel(SamplerDev sm, MapGeom g, RobotDev rb, uint64_t seed, uint64_t first_index,
                     size_t n, double* __restrict__ se3_out, FieldDev f, PoseRec* __restrict__ recs) {
  // A lane's 7 doubles are 56 bytes apart from its neighbour's: written directly, every store instruction
  // touches 28 cache lines with 8 useful bytes in each 56.  The wavefront's 64 states go through LDS instead and
  // leave as seven fully coalesced 512-byte rows.  recs (may be null): the per-state PoseRec of the validity
  // pipeline, produced here while the state is in registers (fused sample + validate).
  __shared__ double stage[4][64 * 8];  // 64 states x 7 doubles, then 64 PoseRecs x 64 bytes
  __shared__ float row_cdf_lds[ARTP_ROW_CDF_LDS];
  const float* row_cdf = stage_row_cdf(sm, g, row_cdf_lds);
  const int lane = threadIdx.x & 63;
  double* sw = stage[threadIdx.x >> 6];
  // One wavefront = 64 consecutive states, no grid-stride loop: in a loop the compiler hoists the f64 polynomial
  // constants of acos / atan2 / sincos out of it into SGPRs, runs out of them and parks them in VGPR lanes
  // (v_writelane / v_readlane were 11 % of the kernel's VALU instructions); straight-line code takes them as
  // literals on the scalar unit.
  const size_t i0 = (size_t)blockIdx.x * blockDim.x + (threadIdx.x - lane);  // wave-uniform
  if (i0 >= n) return;
  const size_t i = i0 + lane;
  float4 r[4];
  if (i < n) {
    double st[7];
    sample_one<FROM_DIST>(sm, g, rb, seed, first_index + i, st, row_cdf);
#pragma unroll
    for (int k = 0; k < 7; ++k) sw[lane * 7 + k] = st[k];
    if (recs) make_pose_rec(f, st, r);
  }
  wave_lds_sync();
  const size_t live = n - i0 < 64 ? n - i0 : 64;
  const size_t cnt = live * 7;
  double* out = se3_out + 7 * i0;
#pragma unroll
  for (int k = 0; k < 7; ++k)
    if ((size_t)(k * 64 + lane) < cnt) __builtin_nontemporal_store(sw[k * 64 + lane], &out[k * 64 + lane]);
  if (recs) {
    // the PoseRecs leave the same way: a lane storing its own 64-byte record issues four 16-byte stores 64 bytes apart
    // (each store instruction touches 64 half-written lines: the L2 fetched 270 MB per batch to merge them); through
    // the staging area (the states are out) the wavefront's 4 KB go as four fully coalesced 1 KB rows
    wave_lds_sync();
    float4* rw = reinterpret_cast<float4*>(sw);
    if (i < n) stage_pose_rec(rw, lane, r);
    wave_lds_sync();
    flush_pose_recs(rw, lane, live, recs + i0);
  }
}

// States of the global sample stream at explicit indices base + idx[j], j < *count (device counter):
// how a rank materialises the accepted states of another rank from the 4-byte indices it received
// (a state is a pure function of (seed, index), so only indices need to cross xGMI).
template <bool FROM_DIST>
__global__ void __launch_bounds__(256)
sample_states_at_kernel(SamplerDev sm, MapGeom g, RobotDev rb, uint64_t seed, uint64_t base_index,
                        const uint32_t* __restrict__ idx, const unsigned long long* __restrict__ count,
                        size_t cap, double* __restrict__ se3_out) {
  __shared__ float row_cdf_lds[ARTP_ROW_CDF_LDS];
  const float* row_cdf = stage_row_cdf(sm, g, row_cdf_lds);
  const size_t n = (size_t)(*count) < cap ? (size_t)(*count) : cap;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    double st[7];
    sample_one<FROM_DIST>(sm, g, rb, seed, base_index + idx[i], st, row_cdf);
#pragma unroll
    for (int k = 0; k < 7; ++k) se3_out[7 * i + k] = st[k];
  }
}

// ---- accepted states of EVERY rank from the gathered validity bitmaps, two launches whatever the world size --------
// (1) one workgroup per rank: exclusive prefix sum of the popcounts of the rank's first `words` bitmap words -> the
//     rank of the first set bit of every word, and the rank's count (bits beyond `prefix_bits` are ignored);
// (2) one lane per bit: a set bit whose rank is below `cap` is the rank-th accepted state of that rank's batch:
//     the lane re-samples it from (seed, base[r] + bit index) and writes it to out[r][rank].
// (Per rank, hipcub's select + a sampling launch cost 73 us; at 8 ranks that was 0.58 ms on a 1.29 ms step.)
// grid (tiles of 1024 words, ranks), 256 threads x 4 consecutive words (coalesced 16-byte loads): offsets[r][w] = set
// bits in front of word w WITHIN its tile, tile_tot[r][tile] = set bits of the tile, counts[r] += that (zeroed by the caller)
constexpr int ARTP_BITS_TILE = 1024;
__device__ __forceinline__ unsigned long long masked_word(const unsigned long long* __restrict__ b, size_t w, size_t words,
                                                          size_t prefix_bits) {
  if (w >= words) return 0ull;
  unsigned long long v = b[w];
  const size_t lo = w * 64;
  if (lo + 64 > prefix_bits) v = lo >= prefix_bits ? 0ull : (v & (~0ull >> (64 - (prefix_bits - lo))));
  return v;
}

__global__ void __launch_bounds__(256)
bits_word_offsets_kernel(const unsigned long long* __restrict__ bits, size_t words_per_rank, size_t words, size_t prefix_bits,
                         unsigned* __restrict__ offsets /*[ranks][words]*/, unsigned* __restrict__ tile_tot /*[ranks][tiles]*/,
                         unsigned long long* __restrict__ counts) {
  __shared__ unsigned wave_sum[4];
  const int r = blockIdx.y, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const unsigned long long* b = bits + (size_t)r * words_per_rank;
  const size_t w0 = (size_t)blockIdx.x * ARTP_BITS_TILE + (size_t)t * 4;
  unsigned c[4], run = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    c[i] = run;                                   // exclusive within the thread
    run += (unsigned)__popcll(masked_word(b, w0 + i, words, prefix_bits));
  }
  unsigned incl = run;                            // inclusive scan of the threads' sums across the wavefront
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned v = __shfl_up(incl, d, 64);
    if (lane >= d) incl += v;
  }
  if (lane == 63) wave_sum[wave] = incl;
  __syncthreads();
  unsigned base = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (k < wave) base += wave_sum[k];
  const unsigned excl = base + incl - run;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (w0 + i < words) offsets[(size_t)r * words + w0 + i] = excl + c[i];
  if (t == 255) {
    const unsigned tot = base + incl;
    tile_tot[(size_t)r * gridDim.x + blockIdx.x] = tot;
    atomicAdd(&counts[r], (unsigned long long)tot);
  }
}

struct RankBases { unsigned long long base[16]; };  // first global sample index of every rank's batch

// position of the k-th (0-based) set bit of a 64-bit word; k < popcount
__device__ __forceinline__ int select_bit64(unsigned long long v, int k) {
  int pos = 0;
#pragma unroll
  for (int w = 32; w >= 1; w >>= 1) {
    const int cnt = __popcll(v & ((1ull << w) - 1ull));
    if (k >= cnt) {
      k -= cnt;
      v >>= w;
      pos += w;
    }
  }
  return pos;
}

// One lane per OUTPUT state: lane j of rank r finds the bitmap word that holds the rank's j-th accepted candidate -- the
// tile by walking the (<= 64) tile totals, the word by a binary search over the tile's 1024 exclusive offsets, the bit by
// select -- and re-samples that candidate.  Every lane has a state to make and consecutive lanes write consecutive
// rows (the earlier form, a wavefront per pair of bitmap words, kept ~47 of 64 lanes busy on the words it needed and
// launched two thirds of its wavefronts only to find them beyond the requested prefix: 60 us for 65 536 states).
template <bool FROM_DIST>
__global__ void __launch_bounds__(256)
materialise_from_bits_kernel(SamplerDev sm, MapGeom g, RobotDev rb, uint64_t seed, RankBases bases,
                             const unsigned long long* __restrict__ bits, size_t words_per_rank, size_t words,
                             size_t prefix_bits, const unsigned* __restrict__ offsets,
                             const unsigned* __restrict__ tile_tot, int n_tiles, size_t cap,
                             const unsigned long long* __restrict__ counts, double* __restrict__ out /*[ranks][cap][7]*/) {
  __shared__ float row_cdf_lds[ARTP_ROW_CDF_LDS];
  const float* row_cdf = stage_row_cdf(sm, g, row_cdf_lds);
  const int r = blockIdx.y;
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t have = counts[r] < cap ? (size_t)counts[r] : cap;
  if (j >= have) return;
  unsigned rest = (unsigned)j;
  int tile = 0;
  for (; tile < n_tiles - 1; ++tile) {
    const unsigned tt = tile_tot[(size_t)r * n_tiles + tile];
    if (rest < tt) break;
    rest -= tt;
  }
  // the last word of the tile whose exclusive offset is <= rest (it holds the bit: the next offset is larger)
  const size_t w_first = (size_t)tile * ARTP_BITS_TILE;
  const size_t w_end = w_first + ARTP_BITS_TILE < words ? w_first + ARTP_BITS_TILE : words;
  const unsigned* off = offsets + (size_t)r * words;
  size_t lo = w_first, hi = w_end;  // invariant: off[lo] <= rest, answer in [lo, hi)
  while (hi - lo > 1) {
    const size_t mid = (lo + hi) >> 1;
    if (off[mid] <= rest) lo = mid; else hi = mid;
  }
  const unsigned long long word = masked_word(bits + (size_t)r * words_per_rank, lo, words, prefix_bits);
  const int bit = select_bit64(word, (int)(rest - off[lo]));
  double st[7];
  sample_one<FROM_DIST>(sm, g, rb, seed, bases.base[r] + lo * 64 + (unsigned)bit, st, row_cdf);
  double* o = out + ((size_t)r * cap + j) * 7;
#pragma unroll
  for (int q = 0; q < 7; ++q) o[q] = st[q];
}

// ---- R7 ------------------------------------------------------------------------------------------
#define ARTP_MAX_QUATERNION_NORM_ERROR 1e-9

__device__ __forceinline__ double so3_arc_length(const double* q1, const double* q2) {
  const double dq = fabs(q1[0] * q2[0] + q1[1] * q2[1] + q1[2] * q2[2] + q1[3] * q2[3]);
  if (dq > 1.0 - ARTP_MAX_QUATERNION_NORM_ERROR) return 0.0;
  return acos(dq);
}

// OMPL SE3StateSpace::interpolate: R^3 lerp + SO3 slerp.  The part of the slerp that depends on the two end states only
// (arc length: an acos; 1 / sin(theta); the sign of the quaternion dot product) is its own function: an edge has ~50
// interior states, and the expansion kernel takes these three numbers per edge from the planning kernel instead of
// forming them per state (the same operations on the same inputs: the same bits).
struct SlerpEdge { double theta, inv_sin, sgn; };  // inv_sin / sgn only meaningful when theta > DBL_EPSILON
__device__ __forceinline__ SlerpEdge slerp_edge(const double* q1, const double* q2) {
  SlerpEdge e;
  e.theta = so3_arc_length(q1, q2);
  e.inv_sin = 0.0;
  e.sgn = 1.0;
  if (e.theta > 2.220446049250313e-16) {
    e.inv_sin = 1.0 / sin(e.theta);
    const double dq = q1[0] * q2[0] + q1[1] * q2[1] + q1[2] * q2[2] + q1[3] * q2[3];
    if (dq < 0) e.sgn = -1.0;
  }
  return e;
}
__device__ __forceinline__ void se3_interpolate_pre(const double* a, const double* b, double t, const SlerpEdge& e,
                                                    double* out) {
#pragma unroll
  for (int i = 0; i < 3; ++i) out[i] = a[i] + (b[i] - a[i]) * t;
  const double* q1 = a + 3;
  const double* q2 = b + 3;
  if (e.theta > 2.220446049250313e-16) {
    const double d = e.inv_sin;
    const double s0 = sin((1.0 - t) * e.theta);
    double s1 = sin(t * e.theta);
    if (e.sgn < 0) s1 = -s1;
    out[3] = (q1[0] * s0 + q2[0] * s1) * d;
    out[4] = (q1[1] * s0 + q2[1] * s1) * d;
    out[5] = (q1[2] * s0 + q2[2] * s1) * d;
    out[6] = (q1[3] * s0 + q2[3] * s1) * d;
  } else {
    out[3] = q1[0];
    out[4] = q1[1];
    out[5] = q1[2];
    out[6] = q1[3];
  }
}
__device__ __forceinline__ void se3_interpolate(const double* a, const double* b, double t,
                                                double* out) {
  se3_interpolate_pre(a, b, t, slerp_edge(a + 3, b + 3), out);
}

// checkMotion in two passes (first overload only; ARTP_COARSE_STRIDE): an edge is valid iff ALL its states are, so the
// order in which they are looked at is free.  Pass 1 validates s2 and every S-th interior state of every edge, pass 2 the
// rest -- of the edges pass 1 left alive only.  An invalid edge usually fails in a run of consecutive states (a third of
// its states on the bench's batch), so the subsample catches five in six of them and their other states are never
// expanded.  Task j of an edge in a pass <-> task k of the edge:
//   pass 0 (single pass): k = j;   pass 1: j = 0 -> k = 0 (s2), j >= 1 -> k = S j;   pass 2: k = j + j / (S - 1) + 1
#define ARTP_COARSE_STRIDE 8   // default S (artp_ctx::edge_coarse_stride; 4 .. 16 measured: see DESIGN 4.3)
__device__ __forceinline__ uint32_t edge_task_of_pass(int pass, uint32_t j, uint32_t S) {
  if (pass == 1) return j * S;
  if (pass == 2) return j + j / (S - 1u) + 1u;
  return j;
}
// tasks of an edge with nd segments in a pass (nd >= 2: interior states 1 .. nd - 1)
__device__ __forceinline__ uint32_t edge_tasks_in_pass(int pass, uint32_t nd, uint32_t S) {
  const uint32_t interior = nd >= 2 ? nd - 1 : 0u, coarse = interior / S;
  return pass == 1 ? 1u + coarse : interior - coarse;
}

// mode 0: DiscreteMotionValidator::checkMotion -> tasks = 1 (s2) + max(nd-1, 0), nd = validSegmentCount
// mode 1: PRMMotionCost::addValidMilestone     -> tasks = n_interp = floor(lateral / 0.5)
// counts[e] = number of wave-tasks of edge e, aux[e] = nd (mode 0) or n_interp (mode 1).
// Per-edge task counts are capped at ARTP_MAX_EDGE_TASKS (an edge needing more -- non-finite states, a
// degenerate state-space extent -- is reported through *overflow and the call fails with ARTP_ERR_INVALID_ARG
// instead of wrapping the 32-bit scan); *total64 receives the sum of all counts.
#define ARTP_MAX_EDGE_TASKS (1u << 22)
__device__ __forceinline__ unsigned clamp_task_count(double x, int* overflow) {
  if (!(x >= 0.0) || !(x <= (double)ARTP_MAX_EDGE_TASKS)) {  // NaN, negative, huge
    *overflow = 1;
    return 0u;
  }
  return (unsigned)x;
}

// tasks of one edge: mode 0 -> *aux = nd = CompoundStateSpace::validSegmentCount, tasks = 1 (s2) + max(nd - 1, 0);
// mode 1 -> *aux = tasks = n_interp = floor(lateral distance / 0.5).  ONE definition for the batch planner
// (motion_plan_kernel) and the latency kernel (check_motions_few_kernel): the same operations, the same bits.
__device__ __forceinline__ uint32_t edge_task_count(const MapGeom& g, double z_extent, double r3_extent_override, int mode,
                                                    const double* a, const double* b, uint32_t* aux, int* overflow) {
  if (mode == 0) {
    // CompoundStateSpace::validSegmentCount: max over R^3 and SO3 of ceil(dist / (0.01*maxExtent));
    // R^3 bounds = map centre -/+ FULL length (art_planner/src/planner.cpp:146-156).
    const double ex = (g.pos_x + g.len_x) - (g.pos_x - g.len_x);
    const double ey = (g.pos_y + g.len_y) - (g.pos_y - g.len_y);
    double ext = 0.0;
    ext += ex * ex;
    ext += ey * ey;
    ext += z_extent * z_extent;
    // artp_set_r3_extent: the R^3 maxExtent frozen at an earlier map's bounds (OMPL keeps longestValidSegment_ from
    // the first StateSpace::setup(), planner.cpp:146-163 never re-runs it)
    const double seg_r3 = (r3_extent_override > 0.0 ? r3_extent_override : sqrt(ext)) * 0.01;
    double d2 = 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const double diff = a[i] - b[i];
      d2 += diff * diff;
    }
    const unsigned n_r3 = clamp_task_count(ceil(sqrt(d2) / seg_r3), overflow);
    const double seg_so3 = (0.5 * 3.14159265358979323846) * 0.01;
    const unsigned n_so3 = clamp_task_count(ceil(so3_arc_length(a + 3, b + 3) / seg_so3), overflow);
    const unsigned nd = n_r3 > n_so3 ? n_r3 : n_so3;
    *aux = nd;
    return 1u + (nd >= 2 ? nd - 1 : 0u);
  }
  const double dx = b[0] - a[0];
  const double dy = b[1] - a[1];
  const double dist = sqrt(dx * dx + dy * dy);
  const unsigned n_interp = clamp_task_count(floor(dist / 0.5), overflow);
  *aux = n_interp;
  return n_interp;
}

__global__ void __launch_bounds__(256)
motion_plan_kernel(MapGeom g, double z_extent, int mode, const double* __restrict__ s1,
                   const double* __restrict__ s2, size_t n, uint32_t* __restrict__ counts,
                   uint32_t* __restrict__ aux, uint8_t* __restrict__ valid, int* __restrict__ overflow,
                   unsigned long long* __restrict__ total64, SlerpEdge* __restrict__ slerp,
                   uint32_t* __restrict__ counts_pass1 = nullptr, unsigned long long* __restrict__ total_pass1 = nullptr,
                   uint32_t coarse_stride = ARTP_COARSE_STRIDE, double r3_extent_override = 0.0) {
  unsigned long long my_total = 0, my_total1 = 0;
  int my_overflow = 0;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n;
       e += (size_t)gridDim.x * blockDim.x) {
    const double* a = s1 + 7 * e;
    const double* b = s2 + 7 * e;
    uint32_t ax;
    const uint32_t cnt = edge_task_count(g, z_extent, r3_extent_override, mode, a, b, &ax, &my_overflow);
    counts[e] = cnt;
    aux[e] = ax;
    valid[e] = 1;
    if (slerp) slerp[e] = slerp_edge(a + 3, b + 3);
    my_total += cnt;
    if (counts_pass1) {  // mode 0: s2 + every S-th interior state
      const uint32_t c1 = edge_tasks_in_pass(1, ax, coarse_stride);
      counts_pass1[e] = c1;
      my_total1 += c1;
    }
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) my_total += __shfl_xor(my_total, m, 64);
  // one atomic per workgroup (a returning atomic per wavefront on one word: 4096 of them for 2^18 edges were most of
  // this kernel's 55 us)
  __shared__ unsigned long long wave_total[4];
  if ((threadIdx.x & 63) == 0) wave_total[threadIdx.x >> 6] = my_total;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long tot = wave_total[0] + wave_total[1] + wave_total[2] + wave_total[3];
    if (tot) atomicAdd(total64, tot);
  }
  if (my_overflow) atomicExch(overflow, 1);
  if (total_pass1) {  // wave-uniform
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) my_total1 += __shfl_xor(my_total1, m, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) wave_total[threadIdx.x >> 6] = my_total1;
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned long long tot = wave_total[0] + wave_total[1] + wave_total[2] + wave_total[3];
      if (tot) atomicAdd(total_pass1, tot);
    }
  }
}

// offsets = exclusive scan of counts (n+1 entries, offsets[n] = total).  One lane per task (edge e, interior state k):
// the interpolated state of DiscreteMotionValidator::checkMotion (mode 0: k = 0 is s2, then t = k / nd) or of
// addValidMilestone's 0.5 m rule (mode 1: t = (k + 1) / (n_interp + 1)), and the edge it belongs to.
// The interpolated state never leaves the registers: the validity pipeline only reads its PoseRec, so
// the kernel emits that (coalesced through LDS) instead of 56 bytes of f64 state per lane for pose_rec_kernel to read
// back (13 M states per 2^18 checkMotion edges: 0.74 GB written with 56-byte strides, read again, and a launch).
// counts of pass 2: the remaining interior states of the edges pass 1 left alive; their sum to *total64
__global__ void __launch_bounds__(256)
coarse_pass2_counts_kernel(const uint32_t* __restrict__ aux, const uint8_t* __restrict__ valid, size_t n, uint32_t S,
                           uint32_t* __restrict__ counts, unsigned long long* __restrict__ total64) {
  unsigned long long mine = 0;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
    const uint32_t c = valid[e] ? edge_tasks_in_pass(2, aux[e], S) : 0u;
    counts[e] = c;
    mine += c;
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) mine += __shfl_xor(mine, m, 64);
  __shared__ unsigned long long ws[4];
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = mine;
  __syncthreads();
  if (threadIdx.x == 0 && (ws[0] + ws[1] + ws[2] + ws[3])) atomicAdd(total64, ws[0] + ws[1] + ws[2] + ws[3]);
}

__global__ void __launch_bounds__(256)
expand_edges_recs_kernel(FieldDev f, int mode, const double* __restrict__ s1, const double* __restrict__ s2, size_t n,
                         const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ aux,
                         const SlerpEdge* __restrict__ slerp, PoseRec* __restrict__ recs, uint32_t* __restrict__ edge_of,
                         int pass = 0, uint32_t coarse_stride = ARTP_COARSE_STRIDE) {
  __shared__ float4 stage[4][64 * 4];
  const int lane = threadIdx.x & 63;
  float4* rw = stage[threadIdx.x >> 6];
  const size_t total = offsets[n];
  for (size_t w0 = (size_t)blockIdx.x * blockDim.x + (threadIdx.x - lane); w0 < total;
       w0 += (size_t)gridDim.x * blockDim.x) {  // w0: wave-uniform
    const size_t w = w0 + lane;
    // edge e with offsets[e] <= w < offsets[e+1].  The 64 tasks of a wavefront are consecutive: ONE search (wave-uniform,
    // scalar loads) finds the edge of its first task, the next 64 offsets are held one per lane, and each lane counts
    // how many of them lie at or below its task with six shuffles instead of log2(n) dependent loads of its own.
    uint32_t e0;
    {
      const uint32_t w0u = __builtin_amdgcn_readfirstlane((uint32_t)w0);
      uint32_t lo = 0, hi = (uint32_t)n;
      while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (offsets[mid] <= w0u) lo = mid; else hi = mid;
      }
      e0 = lo;
    }
    const size_t probe = (size_t)e0 + 1 + lane;
    const uint32_t mine = offsets[probe < n ? probe : n];  // offsets[n] = total > every task
    uint32_t cnt = 0;  // number of j in [0, 64) with offsets[e0 + 1 + j] <= w (the sequence is non-decreasing)
#pragma unroll
    for (int step = 32; step >= 1; step >>= 1) {
      const uint32_t v = __shfl(mine, (int)(cnt + step - 1), 64);
      if (v <= (uint32_t)w) cnt += step;
    }
    {
      const uint32_t last = __shfl(mine, 63, 64);
      if (cnt == 63 && last <= (uint32_t)w) cnt = 64;
    }
    if (w < total) {
      size_t e = (size_t)e0 + cnt;
      if (cnt == 64) {  // more than 64 edges (empty ones) start inside this wavefront's tasks: search on from there
        size_t lo = e, hi = n;
        while (hi - lo > 1) {
          const size_t mid = (lo + hi) >> 1;
          if (offsets[mid] <= w) lo = mid; else hi = mid;
        }
        e = lo;
      }
      const uint32_t k = edge_task_of_pass(pass, (uint32_t)(w - offsets[e]), coarse_stride);
      const double* a = s1 + 7 * e;
      const double* b = s2 + 7 * e;
      double st[7];
      if (mode == 0) {
        if (k == 0) {
#pragma unroll
          for (int i = 0; i < 7; ++i) st[i] = b[i];
        } else {
          const uint32_t nd = aux[e];
          se3_interpolate_pre(a, b, (double)k / (double)nd, slerp[e], st);
        }
      } else {
        const uint32_t n_interp = aux[e];
        const double n_interp_div = 1.0 / (n_interp + 1);
        se3_interpolate_pre(a, b, (k + 1) * n_interp_div, slerp[e], st);
      }
      float4 r[4];
      make_pose_rec(f, st, r);
      stage_pose_rec(rw, lane, r);
      edge_of[w] = (uint32_t)e;
    }
    wave_lds_sync();
    flush_pose_recs(rw, lane, total - w0 < 64 ? total - w0 : 64, recs + w0);
    wave_lds_sync();
  }
}

// an invalid interior state clears its edge's label (all writers store 0)
__global__ void __launch_bounds__(256)
reduce_edges_kernel(const uint8_t* __restrict__ state_valid, const uint32_t* __restrict__ edge_of,
                    const uint32_t* __restrict__ offsets, size_t n, uint8_t* __restrict__ edge_valid) {
  const size_t total = offsets[n];
  for (size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x; w < total;
       w += (size_t)gridDim.x * blockDim.x)
    if (!state_valid[w]) edge_valid[edge_of[w]] = 0;
}

// checkMotion(s1, s2, lastValid) (OMPL 1.4.2 DiscreteMotionValidator, second overload): the interior states
// j = 1 .. nd-1 are tested in order, then s2; lastValid.second = (j - 1) / nd at the first failing j, and
// (nd - 1) / nd when only s2 fails.  With "order" = j - 1 for interior task j and nd - 1 for s2 both read
// order / nd, so the first failure is a minimum over the failing tasks of an edge (mode-0 task k = 0 is s2,
// k >= 1 is interior state j = k).
__global__ void __launch_bounds__(256)
reduce_edges_first_bad_kernel(const uint8_t* __restrict__ state_valid, const uint32_t* __restrict__ edge_of,
                              const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ aux, size_t n,
                              uint32_t* __restrict__ first_bad) {
  const size_t total = offsets[n];
  for (size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x; w < total;
       w += (size_t)gridDim.x * blockDim.x) {
    if (state_valid[w]) continue;
    const uint32_t e = edge_of[w];
    const uint32_t k = (uint32_t)(w - offsets[e]);
    const uint32_t nd = aux[e];
    atomicMin(&first_bad[e], k == 0 ? (nd >= 1 ? nd - 1 : 0u) : k - 1);
  }
}

// t_out[e] = lastValid.second, state_out[e] = interpolate(s1, s2, lastValid.second) for failing edges; passing
// edges report t = 1 and s2 (OMPL leaves lastValid untouched for them).
__global__ void __launch_bounds__(256)
last_valid_kernel(const double* __restrict__ s1, const double* __restrict__ s2, size_t n,
                  const uint32_t* __restrict__ aux, const uint32_t* __restrict__ first_bad,
                  double* __restrict__ t_out, double* __restrict__ state_out) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
    const uint32_t fb = first_bad[e];
    const int nd = (int)aux[e];
    double st[7];
    double t = 1.0;
    if (fb == 0xffffffffu) {
#pragma unroll
      for (int i = 0; i < 7; ++i) st[i] = s2[7 * e + i];
    } else {
      // nd == 0 (identical states, invalid s2): OMPL evaluates (double)(nd - 1) / (double)nd = -inf
      t = nd > 0 ? (double)fb / (double)nd : (double)(nd - 1) / (double)nd;
      se3_interpolate(s1 + 7 * e, s2 + 7 * e, t, st);
    }
    t_out[e] = t;
    if (state_out) {
#pragma unroll
      for (int i = 0; i < 7; ++i) state_out[7 * e + i] = st[i];
    }
  }
}

// ---- latency form of checkMotion / the 0.5 m rule: a handful of edges per call ------------------------------------
// The reference calls ob::MotionValidator::checkMotion one edge at a time on the solution path
// (prm_motion_cost.cpp:652, lazy_prm_star_min_update.cpp:725, OMPL's PathSimplifier via planner.cpp:272).  <= 64 edges in
// ONE launch: grid (chunks, n_edges), one workgroup per (edge, chunk), five wavefronts = the five boxes of a state side by
// side (validate_few_kernel's layout).  The workgroup forms its edge's segment count and slerp constants itself, then
// takes the tasks k = chunk, chunk + chunks, ... (mode 0: task 0 = s2, task k = interior state k; mode 1: task k =
// interior state k + 1 of n_interp).  The first failing state in OMPL's order is an atomicMin over the failing tasks
// ("order" as in reduce_edges_first_bad_kernel); workgroups stop as soon as the verdict (first overload) or a smaller
// first failure (second overload) is known.  The LAST workgroup of an edge to arrive writes the verdict -- and the
// lastValid pair -- and re-arms the edge's three words for the next call, so the call is one launch with no memset.
// `status` may be mapped host memory (the host polls for done_tag like validate_few_kernel's caller).
#define ARTP_FEW_EDGES 64
#ifdef ARTP_STAGE_TIMING
__device__ unsigned long long g_few_trace[5][8];  // [box wavefront][phase] wall_clock64 of workgroup (edge 0, chunk 1)
#define ARTP_FEW_MARK(p) do { if (trace_me && lane == 0) g_few_trace[kbox][p] = wall_clock64(); } while (0)
#else
#define ARTP_FEW_MARK(p) do { } while (0)
#endif
struct FewEdgeSync { uint32_t first_bad, arrived, err, pad[61]; };  // device memory, one 256-byte line per edge; armed = {~0, 0, 0}
// two workgroups per CU (LDS: ~63 KB each for the YAML robot): 10 wavefronts = 3 on one SIMD -> at most 168 VGPRs
__global__ void __launch_bounds__(320, 3)  // HIP: second number = wavefronts per SIMD
check_motions_few_kernel(FieldDev fb, FieldDev ff, MapGeom g, RobotDev rb, double z_extent, double r3_extent_override,
                         int mode, const double* __restrict__ s1, const double* __restrict__ s2, uint32_t n,
                         FewEdgeSync* __restrict__ sync, volatile uint8_t* status, uint8_t* __restrict__ valid_dev,
                         uint32_t* __restrict__ aux_out, double* __restrict__ last_t, double* __restrict__ last_state,
                         ScratchCaps caps_torso, ScratchCaps caps_foot, unsigned done_tag) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ double ab[14];
  __shared__ int box_ok[5];
  __shared__ uint32_t known_s;
  const int lane = threadIdx.x & 63;
  const int kbox = threadIdx.x >> 6;  // box index, wave-uniform
  const uint32_t e = blockIdx.y, chunk = blockIdx.x, chunks = gridDim.x;
  if (e >= n) return;
#ifdef ARTP_STAGE_TIMING
  const bool trace_me = e == 0 && chunk == 1;
#endif
  ARTP_FEW_MARK(0);
  if (threadIdx.x < 14) ab[threadIdx.x] = threadIdx.x < 7 ? s1[7 * (size_t)e + threadIdx.x] : s2[7 * (size_t)e + threadIdx.x - 7];
  __syncthreads();
  double a[7], b[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    a[i] = ab[i];
    b[i] = ab[7 + i];
  }
  ARTP_FEW_MARK(1);   // the edge is here (mapped host memory)
  uint32_t aux;
  int overflow = 0;
  uint32_t tasks = edge_task_count(g, z_extent, r3_extent_override, mode, a, b, &aux, &overflow);
  if (overflow) tasks = 0;
  const uint32_t team = tasks < chunks ? (tasks ? tasks : 1u) : chunks;  // workgroups of this edge that take part
  if (chunk >= team) return;
  WaveScratch s;
  if (kbox == 0) {
    s = carve_scratch(smem, 0, caps_torso);
  } else {
    s = carve_scratch(smem + scratch_bytes_per_wave(caps_torso), kbox - 1, caps_foot);
  }
  const SlerpEdge se = slerp_edge(a + 3, b + 3);
  const bool want_last = last_t != nullptr;
  FewEdgeSync* sy = sync + e;
  ARTP_FEW_MARK(2);
  for (uint32_t k = chunk; k < tasks; k += chunks) {
    const uint32_t order = mode == 0 ? (k == 0 ? (aux >= 1 ? aux - 1 : 0u) : k - 1) : k;
    // what the edge's other workgroups have found so far, read ONCE per workgroup (the barriers below need a uniform
    // decision): it only ever skips work
    if (threadIdx.x == 0) known_s = __atomic_load_n(&sy->first_bad, __ATOMIC_RELAXED);
    __syncthreads();
    const uint32_t known = known_s;
    if (want_last ? order >= known : known != 0xffffffffu) {
      if (!want_last) break;
      __syncthreads();   // everybody has read known_s before thread 0 refreshes it
      continue;
    }
    double st[7];
    if (mode == 0) {
      if (k == 0) {
#pragma unroll
        for (int i = 0; i < 7; ++i) st[i] = b[i];
      } else {
        se3_interpolate_pre(a, b, (double)k / (double)aux, se, st);
      }
    } else {
      const double n_interp_div = 1.0 / (aux + 1);
      se3_interpolate_pre(a, b, (k + 1) * n_interp_div, se, st);
    }
    ARTP_FEW_MARK(3);   // [2 -> 3]: segment count, slerp constants, the barrier of known_s, the interpolated state
    const int ok = few_box_ok(fb, ff, g, rb, st, kbox, s, lane);
    ARTP_FEW_MARK(4);   // this wavefront's box
    if (lane == 0) box_ok[kbox] = ok;   // thread 0 read the previous task's values before the barrier above
    __syncthreads();
    if (threadIdx.x == 0) {
      bool err = false, v = true;
#pragma unroll
      for (int q = 0; q < 5; ++q) {
        err = err || box_ok[q] < 0;
        v = v && box_ok[q] > 0;
      }
      if (err) atomicOr(&sy->err, 2u);
      if (!v || err) atomicMin(&sy->first_bad, order);
    }
  }
  ARTP_FEW_MARK(5);     // all boxes met, verdict atomics issued (wavefront 0)
  if (threadIdx.x != 0) return;
  __threadfence();
  const bool last_wg = atomicAdd(&sy->arrived, 1u) + 1u == team;
  ARTP_FEW_MARK(6);
  if (!last_wg) return;
  // last workgroup of the edge: every other one's atomics are visible
  __threadfence();
  const uint32_t fbad = __atomic_load_n(&sy->first_bad, __ATOMIC_RELAXED);
  const uint32_t er = __atomic_load_n(&sy->err, __ATOMIC_RELAXED) | (overflow ? 4u : 0u);
  sy->first_bad = 0xffffffffu;
  sy->arrived = 0u;
  sy->err = 0u;
  const bool v = fbad == 0xffffffffu && !er;
  if (aux_out) aux_out[e] = aux;
  if (want_last) {   // last_valid_kernel's rule
    double st[7];
    double t = 1.0;
    if (fbad == 0xffffffffu) {
#pragma unroll
      for (int i = 0; i < 7; ++i) st[i] = b[i];
    } else {
      const int nd = (int)aux;
      t = nd > 0 ? (double)fbad / (double)nd : (double)(nd - 1) / (double)nd;
      se3_interpolate(a, b, t, st);
    }
    last_t[e] = t;
    if (last_state) {
#pragma unroll
      for (int i = 0; i < 7; ++i) last_state[7 * (size_t)e + i] = st[i];
    }
  }
  if (valid_dev) valid_dev[e] = (uint8_t)v;
  __threadfence_system();   // the lastValid pair lands before the status byte the host polls for
  status[e] = (uint8_t)((v ? 1u : 0u) | (er & 6u) | done_tag);
  __threadfence_system();
}

// ---- the same without a launch: a resident POOL of workgroups (round 6; artp_set_persistent_latency) -----------------------
// One or two edges per host call.  EVERY workgroup polls the request block itself, takes the tasks wg, wg + P, ... of the
// call's edges and writes what IT found -- the smallest failing order, flags, the lastValid state of that order -- to its own
// slot in mapped host memory; the host waits for the slots of the workgroups that had a task and reduces them (min / or).
// No device-side arrival counter, no finalizing workgroup, no hop through a dispatcher: the first form of this kernel had all
// three (one workgroup polling the host, the others spinning on a device word; last arriver writes the verdict) and spent
// 2.4 us in the dispatcher's second PCIe round trip and fence, 1.8 us until a worker held the request, 0.6 + 1.6 us in arrival
// and finalization -- 12.4 us of device time per request against 6.2 us now (profiles/r06_edge_pool.txt).
// The request block is five 64-byte lines, each {tag, word, 7 doubles}.  It lives in DEVICE memory that the host writes through
// the PCIe BAR: polling never leaves the device, so its cost does not depend on P.  (In mapped host memory every poll is a PCIe
// read and the reads of P workgroups queue up: 3.2 - 5.8 us per round trip of a request number with 32 pollers, 6 - 11 us
// with 64, against 2.3 us through the BAR; a pool of 32 polling the host answered in 31 us, slower than one launch per call
// -- so without a large BAR there is no pool and the calls take check_motions_few_kernel.)
// Protocol: the host writes the 35 payload words -- edges, bounds, a header word {request number, mode bits} and a CHECKSUM of
// the other 34 and the number --, a store fence, then the number into line 0 (the doorbell).  A workgroup that sees a new
// number takes the request only if the checksum of what it read verifies; each answer slot carries a check word over its
// fields and the lastValid state it announces, and the host takes a slot only when that verifies.  Neither side depends on
// how stores are combined, split or ordered on the way.  (The first form trusted per-line tags and 16-byte stores: a soak of
// 1.5 million calls found one wrong answer, twice -- scripts/pool_soak.py, profiles/r06_edge_pool.txt.)
// Each workgroup leaves by itself: `leave` bit, 200 us without a request, ARTP_SVC_LIFE_TICKS at the latest.
#define ARTP_POOL_WGS 256
#define ARTP_POOL_MAX_WGS 256
#define ARTP_POOL_MAX_EDGES 2
struct PoolLine { volatile uint32_t tag, word; double v[7]; };   // line 0: tag = the request number (the doorbell), word bit 8 = leave
struct EdgeMailbox {   // DEVICE memory (fine-grained), written by the host through the BAR
  PoolLine line[5];    // 0: s1 of edge 0 | 1: s2 of edge 0 | 2: z_extent, r3_extent_override, header word, .., checksum | 3, 4: edge 1
};
// as 8-byte words of the block: line 2's v[2] = header {request number << 32 | bits 0-7 edges, bit 9 mode, bit 10 lastValid
// wanted}, line 2's v[6] = checksum of the other 34 payload words and the number
#define ARTP_POOL_HDR_LANE 19
#define ARTP_POOL_SUM_LANE 23
__host__ __device__ inline unsigned long long pool_mix(unsigned long long x, unsigned pos) {
  x ^= (unsigned long long)(pos + 1u) * 0x9E3779B97F4A7C15ull;
  x *= 0xFF51AFD7ED558CCDull;
  return x ^ (x >> 29);
}
__host__ __device__ inline uint32_t pool_slot_check(uint32_t tag, uint32_t first_bad, uint32_t flags, uint32_t aux, unsigned long long state_xor) {
  const unsigned long long x = pool_mix(((unsigned long long)tag << 32) | first_bad, 1u) ^ pool_mix(((unsigned long long)flags << 32) | aux, 2u) ^
                               pool_mix(state_xor, 3u);
  return (uint32_t)(x ^ (x >> 32));
}
static_assert(sizeof(PoolLine) == 64 && sizeof(EdgeMailbox) == 320, "a tag per 64-byte line");
// flags: 2 = tile capacity, 4 = segment-count overflow; bits 3..: the edge's task count.  check = pool_slot_check of the other
// four words and of the xor of the 7 doubles of last_state (0 when none was written): the host takes a slot when its check
// verifies, so it cannot act on a slot that reached it in pieces
struct PoolSlot { uint32_t tag, first_bad, flags, aux, check, pad[3]; };
struct PoolResponse {   // mapped host memory, device -> host
  PoolSlot slot[ARTP_POOL_MAX_EDGES][ARTP_POOL_MAX_WGS];
  double last_state[ARTP_POOL_MAX_EDGES][ARTP_POOL_MAX_WGS][7];   // written in front of the slot when first_bad != ~0 and wanted
  volatile uint8_t exited[ARTP_POOL_MAX_WGS];
};
struct PoolCtl { unsigned long long known[ARTP_POOL_MAX_EDGES]; };   // device: (request number << 32) | ~(smallest failing order so far)
#ifdef ARTP_STAGE_TIMING
__device__ unsigned long long g_pool_trace[16];   // wall_clock64 of the phases of one request (scripts/pool_trace.py)
#define ARTP_POOL_MARK(p, cond) do { if ((cond) && lane == 0 && kbox == 0) g_pool_trace[p] = wall_clock64(); } while (0)
#else
#define ARTP_POOL_MARK(p, cond) do { } while (0)
#endif
__global__ void __launch_bounds__(320, 2)   // one workgroup per CU: 256 VGPRs
check_motions_pool_kernel(FieldDev fb, FieldDev ff, MapGeom g, RobotDev rb, const EdgeMailbox* mb, PoolResponse* resp, PoolCtl* ctl,
                          uint32_t last_seq, ScratchCaps caps_torso, ScratchCaps caps_foot) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ double s_edges[2][14], s_zr[2];
  __shared__ int box_ok[5];
  __shared__ uint32_t known_s, s_hdr, s_seq, s_exit;
  const int lane = threadIdx.x & 63;
  const int kbox = threadIdx.x >> 6;
  const uint32_t wg = blockIdx.x, P = gridDim.x;
  WaveScratch s;
  if (kbox == 0) {
    s = carve_scratch(smem, 0, caps_torso);
  } else {
    s = carve_scratch(smem + scratch_bytes_per_wave(caps_torso), kbox - 1, caps_foot);
  }
  const unsigned long long t_start = wall_clock64();
  unsigned long long t_idle = t_start;
  const volatile unsigned long long* words = reinterpret_cast<const volatile unsigned long long*>(mb);
  for (;;) {
    if (kbox == 0) {   // wavefront 0 polls: lanes 0..39 = the five lines (8 x 8 bytes each)
      uint32_t leave = 0, seq = last_seq, hdr = 0;
      for (;;) {
        unsigned long long q = 0;
        if (lane < 40) q = words[lane];
        const unsigned long long h = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(q >> 32)) << 32) |
                                     (uint32_t)__builtin_amdgcn_readfirstlane((int)q);
        seq = (uint32_t)h;
        // `leave` comes FIRST, whatever the number next to it: the host stops the pool with a request pending when some
        // workgroups have already left on their own (it posts that request again to a fresh pool)
        if ((uint32_t)(h >> 32) & 0x100u) {
          leave = 1;
          break;
        }
        if (seq != last_seq) {
          // A new number in line 0.  The request is what the CHECKSUM says it is: the host wrote the 35 payload words (the
          // header word {request number, mode bits} and the checksum among them), a store fence, then this number; the
          // checksum covers every payload word and the number, so a view of the block that mixes two requests -- a line
          // fetched in two halves at different moments, write-combined stores leaving the host in another order than
          // assumed, a number whose word arrives before its payload -- does not verify and is simply read again.
          // (Tags per line and a second read were the first form: one wrong answer in 1.5 million soak calls.)
          const bool payload = lane < 40 && (lane & 7) != 0 && lane != ARTP_POOL_SUM_LANE;
          unsigned long long sum = payload ? pool_mix(q, (unsigned)lane) : 0ull;
#pragma unroll
          for (int off = 32; off >= 1; off >>= 1) sum ^= __shfl_xor(sum, off, 64);
          sum ^= pool_mix((unsigned long long)seq, 64u);
          const unsigned long long want = __shfl(q, ARTP_POOL_SUM_LANE, 64);
          const unsigned long long hw = __shfl(q, ARTP_POOL_HDR_LANE, 64);
          if (sum == want && (uint32_t)(hw >> 32) == seq) {
            hdr = (uint32_t)hw;
            if (lane < 40 && (lane & 7)) {
              const int line = lane >> 3, i = (lane & 7) - 1;
              if (line < 2) {
                reinterpret_cast<unsigned long long*>(s_edges[0])[7 * line + i] = q;
              } else if (line == 2) {
                if (i < 2) reinterpret_cast<unsigned long long*>(s_zr)[i] = q;
              } else {
                reinterpret_cast<unsigned long long*>(s_edges[1])[7 * (line - 3) + i] = q;
              }
            }
            break;
          }
        }
        const unsigned long long now = wall_clock64();
        if (now - t_idle > ARTP_SVC_IDLE_TICKS || now - t_start > ARTP_SVC_LIFE_TICKS) {
          leave = 1;
          break;
        }
        __builtin_amdgcn_s_sleep(4);
      }
      if (lane == 0) {
        s_exit = leave;
        s_hdr = hdr;
        s_seq = seq;
      }
    }
    __syncthreads();
    if (s_exit) {
      if (threadIdx.x == 0) {
        resp->exited[wg] = 1;
        __threadfence_system();
      }
      return;
    }
    ARTP_POOL_MARK(0, wg == 1);   // workgroup 1 holds the request
    const uint32_t hdr = s_hdr, seq = s_seq, n = (hdr & 0xffu) > 1u ? 2u : 1u;
    const int mode = (hdr >> 9) & 1u;
    const bool want_last = mode == 0 && ((hdr >> 10) & 1u);
    const double z_extent = s_zr[0], r3_extent_override = s_zr[1];
    uint32_t base = 0;   // tasks of the edges in front of the current one: the call's tasks are numbered through its edges
    for (uint32_t e = 0; e < n; ++e) {
      double a[7], b[7];
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        a[i] = s_edges[e][i];
        b[i] = s_edges[e][7 + i];
      }
      uint32_t aux;
      int overflow = 0;
      uint32_t tasks = edge_task_count(g, z_extent, r3_extent_override, mode, a, b, &aux, &overflow);
      if (overflow) tasks = 0;
      const SlerpEdge se = slerp_edge(a + 3, b + 3);
      const uint32_t first = (wg + P - base % P) % P;
      base += tasks;
      uint32_t my_bad = 0xffffffffu, my_err = overflow ? 4u : 0u;   // thread 0's
      ARTP_POOL_MARK(1, wg == 1 && e == 0);   // segment count, slerp constants
      for (uint32_t k = first; k < tasks; k += P) {
        const uint32_t order = mode == 0 ? (k == 0 ? (aux >= 1 ? aux - 1 : 0u) : k - 1) : k;
        if (k != first) {
          // a later round of a long edge: what the other workgroups have found so far, read once per workgroup (it only ever
          // skips work; the verdict is the host's minimum over every workgroup's own finding)
          if (threadIdx.x == 0) {
            const unsigned long long w = __hip_atomic_load(&ctl->known[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            known_s = (uint32_t)(w >> 32) == seq ? ~(uint32_t)w : 0xffffffffu;
          }
          __syncthreads();
          const uint32_t known = known_s;
          __syncthreads();   // (read by everybody before thread 0 writes it again)
          if (want_last ? order >= known : known != 0xffffffffu) {
            if (!want_last) break;
            continue;
          }
        }
        double st[7];
        if (mode == 0) {
          if (k == 0) {
#pragma unroll
            for (int i = 0; i < 7; ++i) st[i] = b[i];
          } else {
            se3_interpolate_pre(a, b, (double)k / (double)aux, se, st);
          }
        } else {
          const double n_interp_div = 1.0 / (aux + 1);
          se3_interpolate_pre(a, b, (k + 1) * n_interp_div, se, st);
        }
        ARTP_POOL_MARK(2, wg == 1 && e == 0);   // the interpolated state
        const int ok = few_box_ok(fb, ff, g, rb, st, kbox, s, lane);
        ARTP_POOL_MARK(3, wg == 1 && e == 0);   // its box
        if (lane == 0) box_ok[kbox] = ok;
        __syncthreads();
        if (threadIdx.x == 0) {
          bool err = false, v = true;
#pragma unroll
          for (int q = 0; q < 5; ++q) {
            err = err || box_ok[q] < 0;
            v = v && box_ok[q] > 0;
          }
          if (err) my_err |= 2u;
          if ((!v || err) && order < my_bad) {
            my_bad = order;
            if (tasks > P) atomicMax(&ctl->known[e], ((unsigned long long)seq << 32) | (uint32_t)~order);
          }
        }
        __syncthreads();   // box_ok is the next task's again
      }
      // workgroups without a task of this edge stay silent, except the one task 0 falls to (it reports the task count, from
      // which the host knows whose slots to wait for, and a segment-count overflow)
      if (threadIdx.x == 0 && (first < tasks || first == 0)) {
        unsigned long long state_xor = 0ull;
        if (want_last && my_bad != 0xffffffffu) {   // the lastValid state of THIS workgroup's finding (last_valid_kernel's rule)
          const int nd = (int)aux;
          const double t = nd > 0 ? (double)my_bad / (double)nd : (double)(nd - 1) / (double)nd;
          double st[7];
          se3_interpolate(a, b, t, st);
#pragma unroll
          for (int i = 0; i < 7; ++i) {
            resp->last_state[e][wg][i] = st[i];
            state_xor ^= (unsigned long long)__double_as_longlong(st[i]);
          }
          __threadfence_system();   // ... lands before the slot that announces it
        }
        // two 16-byte stores with system scope (volatile: sc0 sc1, written through -- a plain store may stay in the L2 until a
        // fence or the end of the kernel), no fence on the way
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const uint32_t fl = my_err | (tasks << 3);
        u32x4 rec, rec2;
        rec.x = seq;
        rec.y = my_bad;
        rec.z = fl;
        rec.w = aux;
        rec2.x = pool_slot_check(seq, my_bad, fl, aux, state_xor);
        rec2.y = rec2.z = rec2.w = 0u;
        volatile u32x4* out = reinterpret_cast<volatile u32x4*>(&resp->slot[e][wg]);
        out[0] = rec;
        out[1] = rec2;
        ARTP_POOL_MARK(4, wg == 1 && e == 0);
      }
    }
    last_seq = s_seq;
    t_idle = wall_clock64();
    __syncthreads();
  }
}

// Labels != 0 of a batch.  Sixteen labels per load, one atomic per WORKGROUP of a grid of a few workgroups per CU: with
// an atomic per wavefront of a wavefront-per-256-labels grid (32 768 atomics on one word for 2^22 labels) the counter's
// L2 atomic unit was the whole 0.4 ms of the kernel.
__global__ void __launch_bounds__(256)
count_valid_kernel(const uint8_t* __restrict__ valid, size_t n, unsigned long long* __restrict__ out) {
  __shared__ unsigned wave_sum[4];
  unsigned c = 0;
  const size_t n16 = (reinterpret_cast<uintptr_t>(valid) & 15u) ? 0 : n / 16;  // aligned 16-byte chunks
  const uint4* v16 = reinterpret_cast<const uint4*>(valid);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
    const uint4 q = v16[i];
    const unsigned w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      // a byte is non-zero iff any of its bits is set: fold each byte onto its lowest bit
      unsigned t = w[k] | (w[k] >> 4);
      t |= t >> 2;
      t |= t >> 1;
      c += (unsigned)__popc(t & 0x01010101u);
    }
  }
  for (size_t i = n16 * 16 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    c += valid[i] ? 1u : 0u;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) c += __shfl_xor(c, m, 64);
  if ((threadIdx.x & 63) == 0) wave_sum[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long tot = (unsigned long long)wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
    if (tot) atomicAdd(out, tot);
  }
}


// Measurement helper: algorithmic window vertices of a batch (one lane per state).
__global__ void __launch_bounds__(256)
alg_vertices_kernel(FieldDev fb, FieldDev ff, MapGeom g, RobotDev rb, const double* __restrict__ se3,
                    size_t n, unsigned long long* __restrict__ out) {
  unsigned long long c = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    double st[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) st[k] = se3[7 * i + k];
    float t[3], R[9];
    pose3_from_se3(st, t, R);
    for (int k = 0; k < 5; ++k) {
      const bool body = (k == 0);
      const float ox = body ? rb.torso_off[0] : ((k <= 2) ? rb.feet_off_x : -rb.feet_off_x);
      const float oy = body ? rb.torso_off[1] : ((k & 1) ? rb.feet_off_y : -rb.feet_off_y);
      const float oz = body ? rb.torso_off[2] : 0.0f;
      float pose[16];
      pose[0] = (R[0] * ox + (R[1] * oy + R[2] * oz)) + t[0];
      pose[1] = (R[3] * ox + (R[4] * oy + R[5] * oz)) + t[1];
      pose[2] = (R[6] * ox + (R[7] * oy + R[8] * oz)) + t[2];
      pose[3] = 0.0f;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        pose[4 + 4 * r + 0] = R[3 * r + 0];
        pose[4 + 4 * r + 1] = R[3 * r + 1];
        pose[4 + 4 * r + 2] = R[3 * r + 2];
        pose[4 + 4 * r + 3] = 0.0f;
      }
      if (!map_is_inside(g, (double)pose[0], (double)pose[1])) continue;
      BoxHF b;
      if (body)
        setup_box(fb, pose, rb.torso[0], rb.torso[1], rb.torso[2], b);
      else
        setup_box(ff, pose, rb.foot[0], rb.foot[1], rb.foot[2], b);
      if (b.on_field) c += (unsigned long long)((b.maxX - b.minX + 1) * (b.maxZ - b.minZ + 1));
    }
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) c += __shfl_xor(c, m, 64);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}

}  // namespace artp
