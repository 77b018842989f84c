// cost_kernels.h -- learned motion cost on gfx950.
//
// R8  network.CNNpart (art_planner_motion_cost/.../predictor/network_light.py:78-110): six un-padded
//     convolutions with eval-mode BatchNorm, leaky-ReLU(0.3) and two max-pools, fp16.  BatchNorm is
//     folded into the weights/bias on the host.  TWO launches: conv345_kernel (conv1 o conv2 composed into one 5 x 5
//     layer + pool on the matrix cores inside its patch phase, then conv3 -> conv4 -> pool -> conv5 on LDS-resident halo
//     tiles, MFMA) and conv_ksplit_kernel (the 15 x 15 layer, 85 % of the FLOPs, MFMA).  Activations are NHWC fp16, so for a
//     fixed kernel row the (kw, cin) taps of an output pixel are ONE contiguous run of KW*Cin halfs: the
//     implicit GEMMs walk K in 32-wide steps that are single 16-byte reads per lane and feed
//     v_mfma_f32_16x16x32_f16 (fp32 accumulate).  Weights are pre-packed on the host in fragment order.
//     (The forms that were built, measured and lost -- conv1 o conv2 as its own VALU / MFMA launch, the persistent
//     15 x 15 kernel, the 15 x 15 layer on v_mfma_f32_32x32x16_f16 -- are in cost_kernels_variants.h, variants build only.)
// R9  CostQuery.__call__ + network.FCpart (cost_query.py:39-69, network_light.py:113-165): per edge,
//     gather the 48 features of the start cell, build the 10 geometric inputs, 1x1-conv MLP with three
//     heads -> (energy, time, 1 - prob).  fc_cost_mfma_kernel: the two layers as MFMA tiles over 16 edges, fp32 accuracy
//     from half-float hi / lo operand pairs (fc_cost_kernel / fc_cost_split_kernel: the fp32 VALU forms -- the load-time
//     self-check's reference and its fallback; artp_cost_set_fc_path).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace artp {

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

// Workgroups are dealt to the 8 XCDs round robin (blockIdx % 8), each XCD with its own 4 MB L2.  xcd_contiguous() renumbers the
// workgroups so that those sharing an XCD own CONTIGUOUS tile indices: spatial neighbours' halo rows then meet in that XCD's
// L2 instead of being fetched once per workgroup from the Infinity Cache / HBM (a CU fills from there at ~4 B/clk when every CU
// of the chip does it at once -- round 5, phase counters of conv_kwalk_kernel -- against ~25 B/clk from its L2).
__device__ __forceinline__ int xcd_contiguous(int b, int G) {
  const int x = b & 7, i = b >> 3, q = G >> 3, r = G & 7;
  return x * q + (x < r ? x : r) + i;
}

// ---- the same implicit GEMM for large kernels (the 15 x 15 flatten layer): LDS-staged input patch ---------
// A workgroup of 4 wavefronts computes a TR-row x 16-pixel output tile for all channels from a
// (TR+KH-1) x (16+KW) pixel input patch staged once in LDS (NHWC, 2*CIN bytes per pixel: with CIN = 48 the 16
// lanes of a ds_read_b128 group fall into 16 distinct bank quads, no padding needed; PMC: 0 conflicts).
template <int KH, int KW, int CIN, int COUT, int NT, bool LRELU, int TR_ = 8>
struct ConvLdsCfg {
  static constexpr int KROW = KW * CIN;
  static constexpr int KSTEPS = (KROW + 31) / 32;
  static constexpr int TR = TR_, TP = 16;
  static constexpr int PR = TR + KH - 1;
  static constexpr int XPAD = (KSTEPS * 32 - KROW + CIN - 1) / CIN;  // pixels the zero-padded K tail reaches into
  static constexpr int PPX = TP + KW - 1 + XPAD;
  static constexpr int PIX_B = CIN * 2;
  static constexpr int ROW_B = PPX * PIX_B;
  static constexpr int A_BYTES = PR * ROW_B;
  static_assert(ROW_B % 16 == 0 && A_BYTES % 16 == 0, "16-byte chunks");
};

// ---- K-split variant: the workgroup's four wavefronts split the k-steps, not the pixels -------------------
// Every wavefront accumulates the WHOLE TR-row x 16-pixel x COUT tile (TR x NT accumulators) over the
// k-steps ks = wave, wave+4, ... of every kernel row; the four partial tiles meet in LDS at the end.
//  * B fragments are used by exactly one wavefront of the group: they go global -> VGPR (a ring prefetched two
//    steps ahead that never drains), never through LDS: no staging stores, no barrier inside the main loop.
//  * For a fixed ks the A fragments of kernel row kh are patch rows kh .. kh+TR-1 -- kernel row kh+1 needs
//    ONE new row.  A register ring of TR fragments turns TR LDS reads per step into 1.
//  * Per step: 1 ds_read_b128 + NT global loads feed TR*NT MFMAs (384 cycles for TR = 8, NT = 3).
//  * Two workgroups per CU (LDS <= 80 KB, <= 256 registers): one's patch load / reduction runs under the other's
//    MFMAs.  The partial tiles meet FOUR ROWS at a time (48 KB of LDS instead of 96) and the finished tile is staged
//    right behind them (the patch is dead by then).
//  * TR is chosen per launch (cost_run_cnn): the launch's last, partly filled round of workgroups costs a full
//    round's time, so at 800 x 800 (376 x 376 outputs) 9-row tiles -- 1008 workgroups = 2 rounds of 512 -- beat
//    8-row tiles -- 1128 = 2.2 rounds -- by a tenth although each tile is an eighth bigger.
//  * NWV = wavefronts per workgroup (round 4).  4: two workgroups share a CU (the C4 launch: 1008 tiles).  8: ONE
//    workgroup per CU whose eight wavefronts split the k-steps eight ways -- for launches with no more tiles than CUs
//    (C3: 242 tiles on 256 CUs, where a 4-wavefront workgroup left every SIMD with a single wavefront and nothing to
//    issue MFMAs while it waits for its B fragments or its LDS row: MFMA busy 0.40).  Same tile, same B traffic, two
//    wavefronts per SIMD; the reduction takes eight partial tiles instead of four.
template <int KH, int KW, int CIN, int COUT, int NT, bool LRELU, int TR = 8, int NWV = 4, int MS = 1>
struct ConvKsplitCfg {
  using P = ConvLdsCfg<KH, KW, CIN, COUT, NT, LRELU, TR * MS>;   // MS > 1: the tile is MS x TR rows (conv_ksplit_kernel)
  static constexpr int RED_BYTES = NWV * 4 * NT * 1024;         // the wavefronts' partial rows, four rows at a time
  static constexpr int STAGE_BYTES = TR * MS * P::TP * COUT * 2;  // finished NHWC tile
  static constexpr int STAGE_OFF = RED_BYTES;
  static constexpr int LDS_BYTES = P::A_BYTES > RED_BYTES + STAGE_BYTES ? P::A_BYTES : RED_BYTES + STAGE_BYTES;
  static_assert(NWV == 4 || NWV == 8, "k-steps are dealt out modulo a power of two");
  static_assert(LDS_BYTES <= (NWV == 4 ? 80 : 160) * 1024, "two workgroups per CU (NWV = 4), one (NWV = 8)");
  static_assert((COUT * 2) % 16 == 0, "a pixel is a whole number of 16-byte chunks");
};

#ifdef ARTP_STAGE_TIMING
// conv_kwalk_kernel and (round 5) conv_ksplit_kernel, per workgroup (mod 256) and wavefront: cycles in [0] patch loads, [1] the main loop, [2] from its end to the first
// reduction barrier's release (waiting for the slower wavefronts; the short K slice's row fetch), [3] the reduction.
// Accumulated in registers, stored once at the end of the kernel: the marks cost an s_memtime each and nothing else.
__device__ unsigned long long g_kwalk_cycles[256 * 12 * 4];
// conv_ksplit_kernel, one record per workgroup (wavefront 0): HW_ID, XCC_ID, then s_memrealtime (100 MHz) at the start, at the
// first MFMA step, at the end of the main loop and at the end of the kernel -- the launch as a Gantt chart per CU
__device__ unsigned long long g_ks_trace[1024 * 6];
__device__ __forceinline__ unsigned ks_hw_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v)); return v; }
__device__ __forceinline__ unsigned ks_xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v; }
#define ARTP_KW_MARK(slot) do { const long long n_ = clock64(); kw_c[slot] += (unsigned long long)(n_ - t_prev); t_prev = n_; } while (0)
#else
#define ARTP_KW_MARK(slot) do { } while (0)
#endif
// RD = slots of the B ring: a fragment is requested RD - 1 steps (of TR x NT MFMAs) before its use (3 = two steps ahead, ~860
// cycles; 5 was measured in round 5: no change -- the fragments are there in time).
// MS (round 5) = row groups: the NWV wavefronts are NK = NWV / MS K slices x MS row groups of TR rows each, the tile is MS x TR rows.
// MS = 2, NWV = 8: wavefronts k and k + 4 -- the two that share a SIMD -- take the SAME K slice for the upper and the lower half of
// an 18-row tile: a SIMD always holds two wavefronts in the main loop (one alone issues an MFMA per 26 cycles, two one per 17:
// tests/cpp/mfma_clock_probe.hip; with two independent 4-wavefront workgroups per CU a SIMD held ONE for a third of the 800^2
// launch), the pair's B fragments are the same lines, and only four partial sums meet per output row.  Measured: the same time
// (800^2: 111.6 us against 108-111) -- a tile's main loop is 2 x 2329 MFMAs per SIMD at 17 cycles = 79 k cycles at ~1.65 GHz either
// way; what the lone wavefronts lose in issue rate the part gives back in clock.  Opt-in ($ARTP_KSPLIT_MS=2), tested.
template <int KH, int KW, int CIN, int COUT, int NT, bool LRELU, int TR, int NWV = 4, bool XCD = true, int RD = 3, int MS = 1>
__global__ void __launch_bounds__(64 * NWV, NWV == 4 ? 2 : 1)
conv_ksplit_kernel(const half_t* __restrict__ in, int Hin, int Win, const half8* __restrict__ wp,
                   const float* __restrict__ bias, half_t* __restrict__ out) {
  using Cfg = ConvLdsCfg<KH, KW, CIN, COUT, NT, LRELU, TR * MS>;
  constexpr int KSTEPS = Cfg::KSTEPS;
  constexpr int NTH = 64 * NWV;
  constexpr int NK = NWV / MS;   // K slices
  static_assert(NK * MS == NWV && (NK & (NK - 1)) == 0 && NK >= 4, "NK K slices (a power of two, >= the 4 rows of a reduction pass)");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* As = smem;
  const int Hout = Hin - KH + 1, Wout = Win - KW + 1;
  const int tiles_x = (Wout + Cfg::TP - 1) / Cfg::TP;
  const int tile = XCD ? xcd_contiguous((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
  const int bx = tile % tiles_x, by = tile / tiles_x;
  const int oy0 = by * TR * MS, ox0 = bx * Cfg::TP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, kg = lane >> 4;
  const int kslice = wave & (NK - 1), grp = wave / NK;   // K slice, row group (rows grp * TR .. of the tile)
#ifdef ARTP_STAGE_TIMING
  long long t_prev = clock64();
  unsigned long long kw_c[4] = {0, 0, 0, 0};
  unsigned long long ks_t[4];
  ks_t[0] = wall_clock64();
#endif

  const char* a_lane = As + grp * TR * Cfg::ROW_B + li * Cfg::PIX_B + kg * 16;
  // which k-steps a wavefront takes, and in which order, rotates with the workgroup index (the workgroups of a launch
  // stream the same 1 MB of B fragments; no two neighbours in the same order)
  const int ks_first = (kslice + (int)(blockIdx.x & (unsigned)(NK - 1))) & (NK - 1);
  const int nj = (KSTEPS - ks_first + NK - 1) / NK;
  const int j0 = (int)((blockIdx.x >> 2) % (unsigned)nj);
  static_assert(KH % RD == 0 || KH == 1, "the B ring runs on across k-steps: slot = (step index) % RD");
  static_assert(KSTEPS >= NK, "every wavefront takes at least one k-step");
  auto ks_of = [&](int jj) {
    const int j = jj + j0 < nj ? jj + j0 : jj + j0 - nj;
    return ks_first + NK * j;
  };
  half8 b[RD][NT];  // ring over (k-step, kernel row), RD - 1 ahead -- it never drains: the last RD - 1 kernel rows of a k-step
                    // prefetch the first ones of the wavefront's next k-step
  // input patch -> LDS (rows are contiguous byte runs of the NHWC image; out-of-image chunks are zero).
  // All of a thread's loads are in flight before the first LDS store (one memory round trip, not 16).
  {
    constexpr int CPR = Cfg::ROW_B / 16;  // 16-byte chunks per patch row
    constexpr int NIT = (Cfg::PR * CPR + NTH - 1) / NTH;
    const long row_bytes = (long)Win * Cfg::PIX_B;
    half8 v[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = tid + it * NTH;
      const int r = c / CPR, cc = c - r * CPR;
      const long off = (long)ox0 * Cfg::PIX_B + (long)cc * 16;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[it][j] = (half_t)0;
      if (c < Cfg::PR * CPR && oy0 + r < Hin && off + 16 <= row_bytes)
        v[it] = *reinterpret_cast<const half8*>(reinterpret_cast<const char*>(in) + (long)(oy0 + r) * row_bytes + off);
    }
    // the ring's first two B fragments are requested BEHIND the patch (loads return in order: waiting for the patch does
    // not wait for them) and in front of the barrier, not after it: one L2 round trip per tile less (round 5)
    {
      const half8* b0 = wp + (size_t)ks_of(0) * NT * 64 + lane;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        b[0][n] = b0[n * 64];
        if (KH > 1) {
#pragma unroll
          for (int d = 1; d < RD - 1; ++d) b[d][n] = b0[(size_t)d * KSTEPS * NT * 64 + n * 64];
        }
      }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = tid + it * NTH;
      const int r = c / CPR, cc = c - r * CPR;
      if (c < Cfg::PR * CPR) *reinterpret_cast<half8*>(As + r * Cfg::ROW_B + cc * 16) = v[it];
    }
  }
  __syncthreads();
  ARTP_KW_MARK(0);
#ifdef ARTP_STAGE_TIMING
  ks_t[1] = wall_clock64();
#endif

  floatx4 acc[TR][NT];
#pragma unroll
  for (int m = 0; m < TR; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = floatx4{0.f, 0.f, 0.f, 0.f};

  // (Round 5, measured and left out again -- profiles/r05_cnn_variants.txt: a ring 4 steps deep (RD 5), the step's loads spread
  // between its MFMAs (sched_group_barrier), and s_setprio turns between a CU's two workgroups: no change at 400^2 or 800^2.
  // A wavefront ON ITS OWN issues one of these MFMAs per 26 cycles, two per SIMD one per 17 (tests/cpp/mfma_clock_probe.hip);
  // the SIMD serves oldest first, so the older workgroup's tile is done when the younger one's is half way, and for 36 of the
  // launch's 105 us at 800^2 a SIMD holds ONE wavefront in its main loop (scripts/ksplit_gantt.py).  A third resident workgroup
  // would need <= 168 registers and <= 53 KB of LDS: the 9-row tile has 180 live registers in the loop and a 66 KB patch.)
  for (int jj = 0; jj < nj; ++jj) {
    const int ks = ks_of(jj);
    const char* a_ks = a_lane + ks * 64;
    const half8* b_ks = wp + (size_t)ks * NT * 64 + lane;  // + kh * KSTEPS * NT * 64
    const half8* b_nx = wp + (size_t)ks_of(jj + 1 < nj ? jj + 1 : jj) * NT * 64 + lane;  // the next k-step's (or a harmless re-read)
    half8 a[TR];     // ring: slot (r % TR) holds patch row r
#pragma unroll
    for (int r = 0; r < TR - 1; ++r) a[r] = *reinterpret_cast<const half8*>(a_ks + r * Cfg::ROW_B);
#pragma unroll
    for (int kh = 0; kh < KH; ++kh) {
      a[(kh + TR - 1) % TR] = *reinterpret_cast<const half8*>(a_ks + (kh + TR - 1) * Cfg::ROW_B);
      {
        const half8* src = kh + RD - 1 < KH ? b_ks + (size_t)(kh + RD - 1) * KSTEPS * NT * 64
                                            : b_nx + (size_t)(kh + RD - 1 - KH) * KSTEPS * NT * 64;
#pragma unroll
        for (int n = 0; n < NT; ++n) b[(kh + RD - 1) % RD][n] = src[n * 64];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < TR; ++m)  // m = TR - 1 uses the row requested just above: it goes last
#pragma unroll
        for (int n = 0; n < NT; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[kh % RD][n], a[(kh + m) % TR], acc[m][n], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // The product is computed TRANSPOSED (weights = first MFMA operand): column (lane & 15) = pixel of the 16-pixel
  // row segment, row ((lane >> 4) * 4 + r) = channel, so a lane holds four consecutive channels of one pixel.
  // The four partial tiles meet in LDS four rows at a time (the patch is dead); wavefront w finishes row 4 h + w.
  // The finished halfs are staged as the NHWC tile [TR][16][COUT] behind the partial rows, then leave as whole
  // 16-byte lanes (a tile row is one contiguous 16*COUT*2-byte run of the output image).
  ARTP_KW_MARK(1);
#ifdef ARTP_STAGE_TIMING
  ks_t[2] = wall_clock64();
#endif
  using KCfg = ConvKsplitCfg<KH, KW, CIN, COUT, NT, LRELU, TR, NWV, MS>;
  typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
  floatx4* red = reinterpret_cast<floatx4*>(smem);
  char* stage = smem + KCfg::STAGE_OFF;
  constexpr int NPASS = (TR + 3) / 4;
#pragma unroll
  for (int h = 0; h < NPASS; ++h) {
    __syncthreads();
#pragma unroll
    for (int mm = 0; mm < 4; ++mm)
      if (4 * h + mm < TR) {
#pragma unroll
        for (int n = 0; n < NT; ++n) red[((wave * 4 + mm) * NT + n) * 64 + lane] = acc[4 * h + mm][n];
      }
    __syncthreads();
    const int m = 4 * h + kslice;
    if (kslice < 4 && m < TR) {  // K slice k < 4 of a row group finishes the group's row 4 h + k: the sum over the group's NK slices
      const int w0 = grp * NK;   // (the other slices only contribute their partials)
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        floatx4 v = red[((w0 * 4 + kslice) * NT + n) * 64 + lane];
#pragma unroll
        for (int w = 1; w < NK; ++w) {
          const floatx4 p = red[(((w0 + w) * 4 + kslice) * NT + n) * 64 + lane];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += p[r];
        }
        const int ch = n * 16 + kg * 4;
        if (ch >= COUT) continue;
        const floatx4 bv = *reinterpret_cast<const floatx4*>(bias + ch);
        half4_t y4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float y = v[r] + bv[r];
          if (LRELU) y = fmaxf(y, 0.3f * y);
          y4[r] = (half_t)y;
        }
        *reinterpret_cast<half4_t*>(stage + (((grp * TR + m) * 16 + li) * COUT + ch) * 2) = y4;
      }
    }
  }
  __syncthreads();
  ARTP_KW_MARK(2);
  {
    constexpr int CPR = 16 * COUT * 2 / 16;  // 16-byte chunks per tile row
    for (int c = tid; c < TR * MS * CPR; c += NTH) {
      const int m = c / CPR, cc = c - m * CPR;
      const int px = (cc * 16) / (COUT * 2);
      if (oy0 + m < Hout && ox0 + px < Wout)
        *reinterpret_cast<half8*>(reinterpret_cast<char*>(out) + ((size_t)(oy0 + m) * Wout + ox0) * COUT * 2 + cc * 16) =
            *reinterpret_cast<const half8*>(stage + m * CPR * 16 + cc * 16);
    }
  }
  ARTP_KW_MARK(3);
#ifdef ARTP_STAGE_TIMING
  if (lane == 0)
    for (int k = 0; k < 4; ++k) g_kwalk_cycles[(((int)blockIdx.x & 255) * 12 + wave) * 4 + k] = kw_c[k];
  if (tid == 0 && blockIdx.x < 1024) {
    ks_t[3] = wall_clock64();
    unsigned long long* r = g_ks_trace + (size_t)blockIdx.x * 6;
    r[0] = ks_hw_id();
    r[1] = ks_xcc_id();
    for (int k = 0; k < 4; ++k) r[2 + k] = ks_t[k];
  }
#endif
}

// ======================================================================================================
// Fused front of the feature extractor (round 3): the eight launch-/latency-bound launches in front of the
// 15 x 15 layer become two kernels whose intermediate activations never leave the CU.
//
// (A) conv12_pool_kernel: init_conv1 + BN -> init_conv2 + BN -> leaky-ReLU -> max_pool 2/2
//     (network_light.py:84-89).  There is NO activation between conv1's BatchNorm and conv2, so the two
//     un-padded 3 x 3 convolutions are ONE 5 x 5 convolution 1 -> 24 whose kernel is the full correlation of the two
//     (composed on the host in double, like the BatchNorm folding: artp_cost_load_weights) plus a constant bias
//     b2 + sum W2 * b1 (no padding: every conv2 output sees all nine taps of b1).  600 MACs per conv2 output
//     instead of 216 + 5184, no 24-channel intermediate.  K = 25 with a single input channel is no MFMA shape:
//     plain f32 VALU, weights as scalar (SGPR) operands, one lane per POOLED pixel (2 x 2 conv outputs from a
//     6 x 6 input window in registers), leaky-ReLU after the max (monotone: max(lrelu(x)) = lrelu(max(x))).
//     The f32 map is rounded to fp16 on the way in (predictor.py:33 `.half()`).
// (B) conv345_kernel: init_conv3 -> init_conv4 -> max_pool 3/1 -> init_conv5 (network_light.py:91-102), each
//     + BN + leaky-ReLU, as implicit GEMMs on v_mfma_f32_16x16x32_f16 from LDS-resident NHWC halo tiles: a workgroup
//     owns a T x T tile of conv5 outputs and computes the (T+6)^2 / (T+4)^2 / (T+2)^2 halo regions of the layers
//     in front of it from a (T+8)^2 input patch; nothing but the patch is read and nothing but the conv5 tile is
//     written.  K is walked in 16-byte CHUNKS across the whole 3 x 3 window (kh, kw, cin), 4 chunks per MFMA:
//     27 chunks = 7 steps for Cin 24, 54 = 14 steps for Cin 48 (per-row padding would take 9 / 15).  A wavefront
//     keeps the accumulators of ALL its (<= 8) 16-pixel tiles live, so a B fragment is fetched from L2 once per
//     wavefront and layer (the conv_ksplit scheme) and the k-steps are software-pipelined (next step's 3 B
//     fragments + 8 A fragments in flight under the current 24 MFMAs).
// ======================================================================================================

// (conv12_pool_kernel, the VALU form of (A): cost_kernels_variants.h)

// (A') conv12_mfma_kernel: the same composed 5 x 5 layer on the matrix cores (round 5).  The VALU form above issues 300
//     packed FMAs per lane behind 150 scalar weight loads and runs at a third of the VALU rate (5.3 us at C3, 14.3 us at
//     C4 = 12 % / 9 % of the extractor's time for 0.1 % of its FLOPs).  As a GEMM it is channels x pixels with K = the
//     window: the weights are the first operand (16 channels x 32 k), the pixels the second (32 k x 16 pixels), a lane
//     ends up with 4 consecutive channels of one pixel.  K = 32 slots = 5 window rows x 6 taps (the 6th has weight 0) + 2
//     spare: lane group g supplies slots 8 g .. 8 g + 7 = FOUR dwords of the half-float patch, dword D = 4 g + j being
//     dword D % 3 of window row D / 3 -- and a window row starts on a dword because the patch is kept twice, shifted by
//     one pixel (copy 1 [x] = copy 0 [x + 1]): the 16 pixels of a tile all have the same x parity (below), so a tile reads one
//     copy with 16 consecutive dwords per lane group.  The 6th tap is masked out of the dword rather than left to its zero
//     weight (a non-finite neighbour would turn 0 * x into NaN where the VALU form never looked).
//     The f32 weights keep their precision: w = hi + lo as two half floats (22 bits), two MFMAs per tile (the FC kernel's
//     scheme), fp32 accumulation from the bias -- the same sums as the VALU form up to the order of fp32 additions.
//     2 x 2 pooling without leaving the lane: the four pool partners (2 py + dy, 2 px + dx) are pixel n of FOUR tiles
//     (dy, dx), n = px: the maximum is elementwise over four accumulators.  A workgroup = 16 x 8 pooled pixels (4
//     wavefronts x 2 pooled rows), its 20 x 36 input window in 3.2 KB of LDS.
constexpr int C12M_PX = 16;                  // pooled pixels per tile row = the N of an MFMA tile
constexpr int C12M_PY = 8;                   // pooled rows per tile
constexpr int C12M_IH = 2 * C12M_PY + 4;     // input rows of a tile
constexpr int C12M_RS = 20;                  // LDS row stride in dwords: 40 half floats >= 2 * 16 + 4 (+ the masked 6th tap)
constexpr int C12M_FRAG_HALFS = 2 * 2 * 64 * 8;  // [channel tile][hi, lo][lane][8]

typedef _Float16 half4_t __attribute__((ext_vector_type(4)));

// What a lane keeps for the composed layer: the weight fragments (two channel tiles, hi / lo), its bias, and where its four
// dwords of a window lie relative to the window's first dword (row stride rsd dwords) + the masks of the 6th tap.
struct C12Frag {
  half8 whi0, wlo0, whi1, wlo1;
  floatx4 b0, b1;
  int off[4];
  unsigned mask[4];
};
__device__ __forceinline__ void c12_frag_load(C12Frag& f, const half8* __restrict__ wfrag, const float* __restrict__ bias,
                                              int lane, int rsd) {
  const int g = lane >> 4;
  f.whi0 = wfrag[lane];
  f.wlo0 = wfrag[64 + lane];
  f.whi1 = wfrag[128 + lane];
  f.wlo1 = wfrag[192 + lane];
  f.b0 = *reinterpret_cast<const floatx4*>(bias + 4 * g);
  f.b1 = *reinterpret_cast<const floatx4*>(bias + 16 + 4 * g);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int D = 4 * g + j, row = D == 15 ? 4 : D / 3, d = D == 15 ? 2 : D % 3;
    f.off[j] = row * rsd + d;
    f.mask[j] = D == 15 ? 0u : (d == 2 ? 0xffffu : 0xffffffffu);
  }
}
// 16 pooled pixels (one per lane & 15): the 2 x 2 conv outputs of each as four MFMA tiles, their maximum, leaky-ReLU, half floats.
// patch: the two copies of the half-float window (copy 1 at + COPY dwords), row stride RSD dwords; base = the dword of the
// pooled pixel's window corner in copy 0 ((2 y) * RSD + x).  o0 = channels 4 g .. 4 g + 3, o1 = 16 + 4 g .. (g < 2).
template <int RSD, int COPY>
__device__ __forceinline__ void conv12_pooled16(const unsigned* __restrict__ patch, int base, const C12Frag& f, half4_t& o0,
                                                half4_t& o1) {
  typedef unsigned uint4_t __attribute__((ext_vector_type(4)));
  // the four windows first, then the eight hi MFMAs, then the eight lo MFMAs (no lo step right behind the hi step it accumulates
  // into).  The phase is bound by VALU issue, not by the matrix pipe: ~110 VALU instructions (masks, maxima, lrelu, conversion,
  // addresses) per 16 MFMAs -- 4.8 k cycles per tile of the fused kernel for 1.7 k of MFMA issue, the same in either order.
  half8 xb[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const unsigned* src = patch + (d & 1) * COPY + (d >> 1) * RSD + base;   // d = 2 dy + dx
    uint4_t xw;
#pragma unroll
    for (int j = 0; j < 4; ++j) xw[j] = src[f.off[j]] & f.mask[j];
    xb[d] = __builtin_bit_cast(half8, xw);
  }
  floatx4 a0[4], a1[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    a0[d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.whi0, xb[d], f.b0, 0, 0, 0);
    a1[d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.whi1, xb[d], f.b1, 0, 0, 0);
  }
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    a0[d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.wlo0, xb[d], a0[d], 0, 0, 0);
    a1[d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.wlo1, xb[d], a1[d], 0, 0, 0);
  }
  floatx4 m0 = a0[0], m1 = a1[0];
#pragma unroll
  for (int d = 1; d < 4; ++d)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      m0[i] = fmaxf(m0[i], a0[d][i]);
      m1[i] = fmaxf(m1[i], a1[d][i]);
    }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    o0[i] = (half_t)fmaxf(m0[i], 0.3f * m0[i]);   // lrelu(v) = max(v, 0.3 v): two issue slots instead of three (the phase is VALU-issue bound)
    o1[i] = (half_t)fmaxf(m1[i], 0.3f * m1[i]);
  }
}

// (conv12_mfma_kernel, (A') as a launch of its own: cost_kernels_variants.h)

// ---- (B) conv3 -> conv4 -> pool3 -> conv5 ------------------------------------------------------------------
template <int T>
struct C345Cfg {
  static constexpr int RI = T + 8, R3 = T + 6, R4 = T + 4, RP = T + 2;
  static constexpr int IN_B = RI * RI * 48, C3_B = R3 * R3 * 96, C4_B = R4 * R4 * 96, P_B = RP * RP * 96,
                       OUT_B = T * T * 96;
  static constexpr int X_B = ((IN_B > C4_B ? IN_B : C4_B) + 255) & ~255;   // input patch, then conv4's region, then the out tile
  static constexpr int Y_B = ((C3_B > P_B ? C3_B : P_B) + 255) & ~255;     // conv3's region, then the pooled region
  static constexpr int W_B = 14 * 3 * 1024;   // conv4's, then conv5's B fragments (14 k-steps x 3 x 64 lanes x 16 bytes)
  // conv3's (7 k-steps) get a region of their own when it fits: conv4's weights can then be committed without
  // waiting for every wavefront to leave conv3 (one barrier less); T = 18 shares the region
  static constexpr bool SEP_W3 = X_B + Y_B + W_B + 7 * 3 * 1024 <= 160 * 1024;
  static constexpr int W3_B = SEP_W3 ? 7 * 3 * 1024 : 0;
  static constexpr int LDS_BYTES = X_B + Y_B + W3_B + W_B;
  static_assert(LDS_BYTES <= 160 * 1024, "one workgroup per CU");
  static_assert(OUT_B <= X_B, "the finished tile is staged over conv4's region");
};

constexpr int C345_NW = 8;               // wavefronts per workgroup: two per SIMD, so that one's LDS / VALU / barrier
constexpr int C345_NT = 64 * C345_NW;    // latencies run under the other's MFMAs (one per SIMD: 46 k cycles per tile
                                         // against 11 k of MFMA work, every phase serial)

// A layer's B fragments ([KS][3][64 lanes] x 16 bytes, chunk order) are the same for all the workgroup's
// wavefronts (they split the pixels): they come in ONCE per workgroup -- global -> registers a whole phase ahead of
// their use (w_prefetch), registers -> LDS once the previous layer is done with the region (w_commit) -- and every
// wavefront reads its fragments from LDS.  (Each wavefront fetching them itself moved 420 KB per tile through the
// L1 against 27 KB of input.)
template <int KS>
struct WRegs { half8 v[(KS * 192 + C345_NT - 1) / C345_NT]; };
template <int KS>
__device__ __forceinline__ void w_prefetch(const half8* __restrict__ wp, WRegs<KS>& r, int tid) {
  constexpr int NCH = KS * 192, NIT = (NCH + C345_NT - 1) / C345_NT;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int c = tid + it * C345_NT;
    r.v[it] = wp[c < NCH ? c : NCH - 1];
  }
}
template <int KS>
__device__ __forceinline__ void w_commit(char* __restrict__ W, const WRegs<KS>& r, int tid) {
  constexpr int NCH = KS * 192, NIT = (NCH + C345_NT - 1) / C345_NT;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int c = tid + it * C345_NT;
    if (c < NCH) *reinterpret_cast<half8*>(W + c * 16) = r.v[it];
  }
}

// One un-padded 3 x 3 layer of the fused kernel: region of NPO = RWO x RWO output pixels (row-major, linear pixel
// index p) from an input region of width RWI (pixel stride CIN * 2 bytes) in LDS.  The workgroup's wavefronts
// take the 16-pixel m-tiles round robin (m-tile = NW * m + wave); acc[m][n] holds m-tile m, channels 16 n .. 16 n + 15.
// The WEIGHTS are the MFMA's first operand and the pixels its second, i.e. the product is computed transposed:
// column (lane & 15) = pixel, row ((lane >> 4) * 4 + r) = channel, so a lane ends up with FOUR CONSECUTIVE CHANNELS
// of one pixel -- one 8-byte store into the NHWC region instead of four 2-byte stores that collide 8-fold on the
// LDS banks (first version: the three epilogues cost more cycles than the three MFMA loops).
// W: the layer's B fragments [KS][3][64] in LDS, chunk order (packed by artp_cost_load_weights).
// bv: the layer's bias, bv[n][r] = channel 16 n + 4 (lane >> 4) + r; it rides in the accumulator.
template <int CIN, int RWI, int RWO, int MTW, int KS>
__device__ __forceinline__ void conv3x3_lds_mfma(const char* __restrict__ in, const char* __restrict__ W,
                                                 const floatx4 (&bv)[3], floatx4 (&acc)[MTW][3], int wave, int lane) {
  constexpr int PIXB = CIN * 2;
  constexpr int CPR = 3 * CIN / 8;      // 16-byte chunks per kernel row (kw, cin)
  constexpr int Q = 3 * CPR;            // chunks of the whole window
  static_assert(KS == (Q + 3) / 4, "MFMA k-steps: 4 chunks = 32 k");
  constexpr int NPO = RWO * RWO;
  const int li = lane & 15, kg = lane >> 4;
  int base[MTW];
#pragma unroll
  for (int m = 0; m < MTW; ++m) {
    int p = (m * C345_NW + wave) * 16 + li;
    p = p < NPO ? p : NPO - 1;          // clamped: the lanes of a partial tile read a valid pixel, their results are dropped
    const int y = p / RWO, x = p - y * RWO;
    base[m] = (y * RWI + x) * PIXB;
#pragma unroll
    for (int n = 0; n < 3; ++n) acc[m][n] = bv[n];
  }
  auto chunk_off = [&](int ks) {
    int q = ks * 4 + kg;
    q = q < Q ? q : Q - 1;              // zero weights there, but the operand must be finite data
    const int r = (q >= CPR) + (q >= 2 * CPR);
    return r * (RWI * PIXB - CPR * 16) + q * 16;   // r * RWI * PIXB + (q - r * CPR) * 16
  };
  half8 a[2][MTW], b[2][3];
  const char* wl = W + lane * 16;
  {
    const int off = chunk_off(0);
#pragma unroll
    for (int n = 0; n < 3; ++n) b[0][n] = *reinterpret_cast<const half8*>(wl + n * 1024);
#pragma unroll
    for (int m = 0; m < MTW; ++m) a[0][m] = *reinterpret_cast<const half8*>(in + base[m] + off);
  }
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int cur = ks & 1, nxt = cur ^ 1;
    if (ks + 1 < KS) {
      const int off = chunk_off(ks + 1);
#pragma unroll
      for (int n = 0; n < 3; ++n) b[nxt][n] = *reinterpret_cast<const half8*>(wl + ((ks + 1) * 3 + n) * 1024);
#pragma unroll
      for (int m = 0; m < MTW; ++m) a[nxt][m] = *reinterpret_cast<const half8*>(in + base[m] + off);
    }
#pragma unroll
    for (int m = 0; m < MTW; ++m)
      if (m < MTW - 1 || (m * C345_NW + wave) * 16 < NPO) {  // wave-uniform: a wavefront without a last m-tile skips it
#pragma unroll
        for (int n = 0; n < 3; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[cur][n], a[cur][m], acc[m][n], 0, 0, 0);
      }
  }
}

// leaky-ReLU + fp16 of a layer's accumulators (bias included) into an LDS region (linear pixel index, 96 bytes per
// pixel).  Transposed product (see above): lane = pixel (lane & 15) of the m-tile, acc[m][n][r] = channel 16 n + 4 (lane >> 4) + r.
template <int NPO, int MTW>
__device__ __forceinline__ void store_region_lds(char* __restrict__ out, const floatx4 (&acc)[MTW][3], int wave,
                                                 int lane) {
  typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
  const int li = lane & 15, kg = lane >> 4;
#pragma unroll
  for (int m = 0; m < MTW; ++m) {
    const int p = (m * C345_NW + wave) * 16 + li;
    if (p < NPO) {
#pragma unroll
      for (int n = 0; n < 3; ++n) {
        half4_t h;
#pragma unroll
        for (int r = 0; r < 4; ++r) h[r] = (half_t)fmaxf(acc[m][n][r], 0.3f * acc[m][n][r]);  // lrelu(v) = max(v, 0.3 v)
        *reinterpret_cast<half4_t*>(out + p * 96 + (n * 16 + kg * 4) * 2) = h;
      }
    }
  }
}

#ifdef ARTP_STAGE_TIMING
__device__ unsigned long long g_cnn_cycles[16];  // conv345 phases (cycles of wavefront 0, summed over workgroups), [15] = workgroups
// The stamps stay in registers until the end of the kernel (round 5): an atomic per mark sat in front of the weight commits'
// s_waitcnt vmcnt -- returns are counted in order -- and 256 workgroups' atomics on one address made those waits look like
// 24 k cycles of "waiting for conv5's weights".
#define ARTP_CNN_MARK(slot) do { const long long n_ = clock64(); t_mark[slot] = (unsigned long long)(n_ - t_prev); t_prev = n_; } while (0)
#define ARTP_CNN_FLUSH() do { if (tid == 0) { for (int s_ = 0; s_ < 13; ++s_) atomicAdd(&g_cnn_cycles[s_], t_mark[s_]); } } while (0)
#else
#define ARTP_CNN_MARK(slot) do { } while (0)
#define ARTP_CNN_FLUSH() do { } while (0)
#endif

// F12 (round 5): the composed conv1 o conv2 + pool layer is computed HERE, into the patch, from the tile's window of the f32 map
// (conv12_pooled16, the arithmetic of conv12_mfma_kernel: the same bits): `in` is not read, no 24-channel image exists, and
// the launch in front is gone.  The window of a (T+8)^2 patch is (2 T + 20)^2 floats -- fewer bytes than the patch itself --,
// kept as half floats (twice, see conv12_mfma_kernel) in conv3's still unused output region.
template <int T, bool XCD = true, bool F12 = false>
__global__ void __launch_bounds__(C345_NT)
conv345_kernel(const half_t* __restrict__ in /*[Hin][Win][24]*/, int Hin, int Win,
               const half8* __restrict__ w3, const float* __restrict__ b3, const half8* __restrict__ w4,
               const float* __restrict__ b4, const half8* __restrict__ w5, const float* __restrict__ b5,
               half_t* __restrict__ out /*[Hin-8][Win-8][48]*/, const float* __restrict__ raw = nullptr /*[H][W]*/, int H = 0,
               int W = 0, const half8* __restrict__ w12 = nullptr, const float* __restrict__ b12 = nullptr) {
  using Cfg = C345Cfg<T>;
  constexpr int NW = C345_NW, NT_ = C345_NT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* X = smem;
  char* Y = smem + Cfg::X_B;
  char* W3 = smem + Cfg::X_B + Cfg::Y_B;
  char* Wl = W3 + Cfg::W3_B;
  const int Hout = Hin - 8, Wout = Win - 8;
  const int tiles_x = (Wout + T - 1) / T;
  const int tile = XCD ? xcd_contiguous((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
  const int bx = tile % tiles_x, by = tile / tiles_x;
  const int oy0 = by * T, ox0 = bx * T;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef ARTP_STAGE_TIMING
  long long t_prev = clock64();
  unsigned long long t_mark[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
  // the three layers' biases (transposed product: four consecutive channels per lane) up front: a global load in
  // front of a layer's first MFMA would sit in its critical path
  floatx4 bv3[3], bv4[3], bv5[3];
#pragma unroll
  for (int n = 0; n < 3; ++n) {
    bv3[n] = *reinterpret_cast<const floatx4*>(b3 + n * 16 + (lane >> 4) * 4);
    bv4[n] = *reinterpret_cast<const floatx4*>(b4 + n * 16 + (lane >> 4) * 4);
    bv5[n] = *reinterpret_cast<const floatx4*>(b5 + n * 16 + (lane >> 4) * 4);
  }
  // ALL three layers' weight fragments are requested up front (round 5): loads retire in order, so waiting for the patch
  // and conv3's weights (requested first) does not wait for the others -- conv4's have the whole of conv3 to arrive and
  // conv5's the whole of conv3 and conv4 (requested a phase ahead, 256 workgroups fetching the same 43 KB at the same
  // moment waited 4-6 k cycles at each commit).  60 registers until the commits.
  WRegs<7> rw3;
  WRegs<14> rw, rw5;
  w_prefetch<7>(w3, rw3, tid);  // conv3's weights travel with the patch
  if constexpr (F12) {
    // the tile's window of the map -> half floats in Y, twice (copy 1 shifted by one pixel); zeros outside the map
    constexpr int RI = Cfg::RI, WIN = 2 * RI + 4, RSD = RI + 3, COPY = WIN * RSD;
    static_assert(2 * COPY * 4 <= Cfg::Y_B, "the window fits conv3's region");
    constexpr int NEL = WIN * WIN, NIT = (NEL + NT_ - 1) / NT_;
    half_t* const p0 = reinterpret_cast<half_t*>(Y);
    half_t* const p1 = reinterpret_cast<half_t*>(Y + COPY * 4);
    float v[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = tid + it * NT_;
      const int r = i / WIN, c = i - r * WIN;
      const int y = 2 * oy0 + r, x = 2 * ox0 + c;
      // an UNCONDITIONAL load from a clamped address: a load under a condition becomes a branch, and hipcc waits for each
      // (s_waitcnt vmcnt(0)) before the next block -- seven serial trips to memory (10.4 k cycles per tile) instead of one
      const bool ok = i < NEL && y < H && x < W;
      const float t = raw[ok ? (size_t)y * W + x : (size_t)0];
      v[it] = ok ? t : 0.0f;
    }
    C12Frag f;
    c12_frag_load(f, w12, b12, lane, RSD);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = tid + it * NT_;
      const int r = i / WIN, c = i - r * WIN;
      if (i < NEL) {
        const half_t h = (half_t)v[it];
        p0[r * 2 * RSD + c] = h;
        if (c > 0) p1[r * 2 * RSD + c - 1] = h;
      }
    }
    w_prefetch<14>(w4, rw, tid);
    w_prefetch<14>(w5, rw5, tid);
    w_commit<7>(W3, rw3, tid);
    __syncthreads();
    ARTP_CNN_MARK(11);
    // the patch: 16 pooled pixels per step and wavefront, linear pixel index over the RI x RI patch
    constexpr int NPX = RI * RI, NQ = (NPX + 15) / 16;
    for (int q = wave; q < NQ; q += NW) {
      int pp = q * 16 + (lane & 15);
      const bool live = pp < NPX;
      pp = live ? pp : NPX - 1;
      const int y = pp / RI, x = pp - y * RI;
      half4_t o0, o1;
      conv12_pooled16<RSD, COPY>(reinterpret_cast<const unsigned*>(Y), 2 * y * RSD + x, f, o0, o1);
      if (!(oy0 + y < Hin && ox0 + x < Win)) {   // outside the pooled image: zeros, as the patch load below
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          o0[i] = (half_t)0;
          o1[i] = (half_t)0;
        }
      }
      if (live) {
        const int g = lane >> 4;
        *reinterpret_cast<half4_t*>(X + pp * 48 + 8 * g) = o0;
        if (g < 2) *reinterpret_cast<half4_t*>(X + pp * 48 + 32 + 8 * g) = o1;
      }
    }
    ARTP_CNN_MARK(12);
  } else {
  // input patch (T+8)^2 x 24 channels -> X; rows are contiguous byte runs of the NHWC image, zeros outside it
  {
    constexpr int CPR = Cfg::RI * 48 / 16;  // 16-byte chunks per patch row
    constexpr int NCH = Cfg::RI * CPR;
    constexpr int NIT = (NCH + NT_ - 1) / NT_;
    const long row_bytes = (long)Win * 48;
    half8 v[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = tid + it * NT_;
      const int r = c / CPR, cc = c - r * CPR;
      const long off = (long)ox0 * 48 + (long)cc * 16;
      // unconditional load from a clamped address (see the fused branch above: conditional loads are serialised)
      const bool ok = c < NCH && oy0 + r < Hin && off + 16 <= row_bytes;
      const half8 t = *reinterpret_cast<const half8*>(reinterpret_cast<const char*>(in) + (ok ? (long)(oy0 + r) * row_bytes + off : 0l));
#pragma unroll
      for (int j = 0; j < 8; ++j) v[it][j] = ok ? t[j] : (half_t)0;
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = tid + it * NT_;
      if (c < NCH) *reinterpret_cast<half8*>(X + c * 16) = v[it];
    }
  }
  w_prefetch<14>(w4, rw, tid);
  w_prefetch<14>(w5, rw5, tid);
  w_commit<7>(W3, rw3, tid);
  }
  __syncthreads();
  ARTP_CNN_MARK(0);
  {  // conv3: X (24 ch) -> Y
    constexpr int MTW = (Cfg::R3 * Cfg::R3 + 16 * NW - 1) / (16 * NW);
    floatx4 acc[MTW][3];
    conv3x3_lds_mfma<24, Cfg::RI, Cfg::R3, MTW, 7>(X, W3, bv3, acc, wave, lane);
    ARTP_CNN_MARK(1);
    store_region_lds<Cfg::R3 * Cfg::R3, MTW>(Y, acc, wave, lane);
    ARTP_CNN_MARK(2);
  }
  if (!Cfg::SEP_W3) __syncthreads();  // shared region: every wavefront must be done with conv3's weights
  w_commit<14>(Wl, rw, tid);          // (own region: no wavefront reads it before the barrier)
  __syncthreads();
  ARTP_CNN_MARK(3);
  {  // conv4: Y -> X (the patch is dead)
    constexpr int MTW = (Cfg::R4 * Cfg::R4 + 16 * NW - 1) / (16 * NW);
    floatx4 acc[MTW][3];
    conv3x3_lds_mfma<48, Cfg::R3, Cfg::R4, MTW, 14>(Y, Wl, bv4, acc, wave, lane);
    ARTP_CNN_MARK(4);
    store_region_lds<Cfg::R4 * Cfg::R4, MTW>(X, acc, wave, lane);
    ARTP_CNN_MARK(5);
  }
  __syncthreads();               // conv4's weights are dead, X is complete
  w_commit<14>(Wl, rw5, tid);
  ARTP_CNN_MARK(6);
  {  // max_pool 3 / 1: X -> Y.  An item = (pooled pixel, 8-channel chunk): nine independent 16-byte reads, a max tree, one
     // store; the items are dealt out thread by thread, fully unrolled (round 5: the column walk with three rows in
     // registers read a third as much but as a chain of dependent LDS round trips -- 6 k cycles for 1.2 k of LDS time).
    constexpr int RP = Cfg::RP, R4 = Cfg::R4, NITEM = RP * RP * 6, NIT = (NITEM + NT_ - 1) / NT_;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int item = tid + it * NT_;
      if (item < NITEM) {
        const int p = item / 6, c = item - p * 6;
        const int y = p / RP, x = p - y * RP;
        const char* src = X + ((y * R4 + x) * 96 + c * 16);
        half8 m = *reinterpret_cast<const half8*>(src);
#pragma unroll
        for (int k = 1; k < 9; ++k)
          m = __builtin_elementwise_max(m, *reinterpret_cast<const half8*>(src + ((k / 3) * R4 + (k % 3)) * 96));
        *reinterpret_cast<half8*>(Y + p * 96 + c * 16) = m;
      }
    }
  }
  __syncthreads();
  ARTP_CNN_MARK(7);
  {  // conv5: Y -> the finished tile, staged in X, then whole 16-byte lanes to the NHWC image
    constexpr int MTW = (T * T + 16 * NW - 1) / (16 * NW);
    floatx4 acc[MTW][3];
    conv3x3_lds_mfma<48, Cfg::RP, T, MTW, 14>(Y, Wl, bv5, acc, wave, lane);
    ARTP_CNN_MARK(8);
    store_region_lds<T * T, MTW>(X, acc, wave, lane);
  }
  __syncthreads();
  ARTP_CNN_MARK(9);
  {
    constexpr int CPR = T * 96 / 16;  // chunks per tile row
    for (int c = tid; c < T * CPR; c += NT_) {
      const int r = c / CPR, cc = c - r * CPR;
      const int px = (cc * 16) / 96;
      if (oy0 + r < Hout && ox0 + px < Wout)
        *reinterpret_cast<half8*>(reinterpret_cast<char*>(out) + ((size_t)(oy0 + r) * Wout + ox0) * 96 + cc * 16) =
            *reinterpret_cast<const half8*>(X + c * 16);
    }
  }
  ARTP_CNN_MARK(10);
  ARTP_CNN_FLUSH();
#ifdef ARTP_STAGE_TIMING
  if (tid == 0) atomicAdd(&g_cnn_cycles[15], 1ull);
#endif
}

// ---- R9: per-edge cost ---------------------------------------------------------------------------------
struct FcWeights {  // BN folded; fp32; offsets into one LDS/global array
  // tar0: [16][10] + [16]; out0: [48][64] + [48]; h1: [24][48]+[24]; h2: [24][48]+[24]; h3: [36][48]+[36];
  // o1: [24]+1; o2: [24]+1; o3: [36]+1
  enum {
    TAR0_W = 0, TAR0_B = TAR0_W + 160, OUT0_W = TAR0_B + 16, OUT0_B = OUT0_W + 48 * 64,
    H1_W = OUT0_B + 48, H1_B = H1_W + 24 * 48, H2_W = H1_B + 24, H2_B = H2_W + 24 * 48,
    H3_W = H2_B + 24, H3_B = H3_W + 36 * 48, O1_W = H3_B + 36, O1_B = O1_W + 24, O2_W = O1_B + 1,
    O2_B = O2_W + 24, O3_W = O2_B + 1, O3_B = O3_W + 36, TOTAL = O3_B + 1
  };
};

struct CostMapGeom {
  int Fh, Fw;         // feature map is Fh x Fw = predictor.features.shape[2], shape[3] (cost_query.py:54-55)
  double feat_res;    // res * featureResDownsampleFactor (cost_query.py:33)
  int row_bias, col_bias;  // cost_query.py:34-35
  double cx, cy;      // map centre subtracted by the server (cost_query_server.py:134-135,160-161)
};

// CostQuery.__call__'s gather index (cost_query.py:54-55) for a start position: float64 arithmetic (the server hands
// float64 numpy to torch), clamp to [1, shape - 2], .long() truncation.  The ONE place the product computes it (both
// fc kernels and artp_cost_debug_query_cells, which the tests compare with the reference's own CostQuery).
__device__ __forceinline__ void cost_query_cell(const CostMapGeom& g, double sx, double sy, int* row, int* col) {
  double pr = (sx - g.cx) / g.feat_res + (double)g.row_bias;
  double pc = (sy - g.cy) / g.feat_res + (double)g.col_bias;
  pr = pr < 1.0 ? 1.0 : (pr > (double)(g.Fh - 2) ? (double)(g.Fh - 2) : pr);
  pc = pc < 1.0 ? 1.0 : (pc > (double)(g.Fw - 2) ? (double)(g.Fw - 2) : pc);
  *row = (int)pr;
  *col = (int)pc;
}

__global__ void __launch_bounds__(256)
cost_query_cells_kernel(const float* __restrict__ edges, size_t B, CostMapGeom g, int* __restrict__ rows, int* __restrict__ cols) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= B) return;
  int r, c;
  cost_query_cell(g, (double)edges[6 * e + 3], (double)edges[6 * e + 4], &r, &c);
  rows[e] = r;
  cols[e] = c;
}

// edges: [B][6] = tx ty tyaw sx sy syaw (motion_cost_objective.h:22, prm_motion_cost.cpp:41-52)
// feat : NHWC fp16 [F][F][48], index [row][col] with row growing along world x (cost_query_server.py:74)
// cost : [B][3] = energy, time, 1 - prob
__global__ void __launch_bounds__(256)
fc_cost_kernel(const float* __restrict__ edges, size_t B, const half_t* __restrict__ feat, CostMapGeom g,
               const float* __restrict__ wts, float* __restrict__ cost) {
  // the weights are wave-uniform operands: read straight from the (scalar-cached) blob, they arrive in SGPRs
  const float* __restrict__ sw = wts;
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= B) return;
  const float* ed = edges + 6 * e;
  const double tx = ed[0], ty = ed[1], tyaw = ed[2], sx = ed[3], sy = ed[4], syaw = ed[5];
  int row, col;
  cost_query_cell(g, sx, sy, &row, &col);  // cost_query.py:51-55
  const half_t* fp = feat + ((size_t)row * g.Fw + col) * 48;
  float x[64];
#pragma unroll
  for (int v = 0; v < 6; ++v) {
    const half8 t = reinterpret_cast<const half8*>(fp)[v];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[v * 8 + j] = (float)t[j];
  }
  // network_light.py:118-132
  const float dx = (float)(tx - sx), dy = (float)(ty - sy);
  float dyaw = (float)(tyaw - syaw);
  const float PI = 3.14159265358979323846f;
  if (dyaw > PI) dyaw -= 2.0f * PI;
  if (dyaw < -PI) dyaw += 2.0f * PI;
  const float sya = (float)syaw;
  float t[10];
  t[0] = dx;
  t[1] = dy;
  t[2] = sqrtf(dx * dx + dy * dy);
  t[3] = atan2f(dy, dx);
  t[4] = dyaw;
  t[5] = cosf(dyaw);
  t[6] = sinf(dyaw);
  t[7] = sya;
  t[8] = cosf(sya);
  t[9] = sinf(sya);
  // tar0 (1x1 conv + BN), concatenated after the 48 map features (network_light.py:135-137)
#pragma unroll
  for (int o = 0; o < 16; ++o) {
    float a = sw[FcWeights::TAR0_B + o];
#pragma unroll
    for (int k = 0; k < 10; ++k) a = fmaf(t[k], sw[FcWeights::TAR0_W + o * 10 + k], a);
    x[48 + o] = a;
  }
  float h[48];
#pragma unroll
  for (int o = 0; o < 48; ++o) {
    float a = sw[FcWeights::OUT0_B + o];
#pragma unroll
    for (int k = 0; k < 64; ++k) a = fmaf(x[k], sw[FcWeights::OUT0_W + o * 64 + k], a);
    h[o] = a > 0.f ? a : 0.3f * a;
  }
  float power = sw[FcWeights::O1_B], tim = sw[FcWeights::O2_B], prob = sw[FcWeights::O3_B];
#pragma unroll 4
  for (int o = 0; o < 24; ++o) {
    float a = sw[FcWeights::H1_B + o], c = sw[FcWeights::H2_B + o];
#pragma unroll
    for (int k = 0; k < 48; ++k) {
      a = fmaf(h[k], sw[FcWeights::H1_W + o * 48 + k], a);
      c = fmaf(h[k], sw[FcWeights::H2_W + o * 48 + k], c);
    }
    a = a > 0.f ? a : 0.3f * a;
    c = c > 0.f ? c : 0.3f * c;
    power = fmaf(a, sw[FcWeights::O1_W + o], power);
    tim = fmaf(c, sw[FcWeights::O2_W + o], tim);
  }
#pragma unroll 4
  for (int o = 0; o < 36; ++o) {
    float a = sw[FcWeights::H3_B + o];
#pragma unroll
    for (int k = 0; k < 48; ++k) a = fmaf(h[k], sw[FcWeights::H3_W + o * 48 + k], a);
    a = a > 0.f ? a : 0.3f * a;
    prob = fmaf(a, sw[FcWeights::O3_W + o], prob);
  }
  power = power > 0.f ? power : 0.f;
  tim = tim > 0.f ? tim : 0.f;
  prob = 1.0f / (1.0f + expf(-prob));
  cost[3 * e + 0] = power;
  cost[3 * e + 1] = tim;
  cost[3 * e + 2] = 1.0f - prob;  // cost_query.py:65-69 returns cost[3] = 1 - prob
}

// ---- the same MLP on the matrix cores -----------------------------------------------------------------------------------
// Per edge the network is two small GEMMs: 64 inputs -> 48 hidden units -> 84 head units (7 300 MACs), then three dot
// products.  Batched over 16 edges they are v_mfma_f32_16x16x32_f16 tiles (weights = first operand: the transposed product,
// a lane ends up with four consecutive units of ONE edge), and fp32 accuracy is kept by splitting every fp32 operand into a
// half-float "hi" and the half-float "lo" of what hi missed (x = hi + lo to 22 bits; hi*hi + lo*hi + hi*lo in the fp32
// accumulator; lo*lo is below fp32's last bit): three MFMAs per product instead of one, still ~40x fewer issue slots than
// the 7 300 FMAs of a lane.  The map features are half floats already (lo = 0).
//  * tar0 (10 geometric inputs -> 16, NO activation, network_light.py:135-137) is composed into out0 on the host, in double,
//    like conv1 o conv2: K = 48 features + 10 inputs + 1 (the composed bias, input 1.0) = 59 of 64.
//  * The hidden units are numbered so that the accumulator layout of the first GEMM IS the second GEMM's operand layout:
//    a lane holds rows 16 t + 4 (lane >> 4) + i of tile t; the second GEMM wants k = 8 (lane >> 4) + j of a 32-wide step --
//    hidden unit 8 g + 4 t + i sits in row 16 t + 4 g + i (t < 2), units 32 .. 47 stay where they are and go through a
//    16-wide step (v_mfma_f32_16x16x16_f16: k = 4 (lane >> 4) + j).  No shuffles, no LDS between the layers.
//  * A wavefront takes 64 edges: one lane per edge forms the gather offset and the geometric inputs (sqrt, atan2, two
//    sin / cos pairs: the VALU part) and parks them in LDS; then four 16-edge tiles run through the GEMMs.
// Blob (fc_mfma_pack on the host): [s][t][hi|lo] 1 KB fragments of the first GEMM (12), [t][hi|lo] of the second's 32-wide
// step (12), [t][hi|lo] 512-byte fragments of its 16-wide step (12), head biases [96], output weights [3][96] zero-padded
// per head, output biases [3].
struct FcMfma {
  enum {
    G1 = 0, G2A = G1 + 12 * 1024, G2B = G2A + 12 * 1024, BIAS2 = G2B + 12 * 512, OUT = BIAS2 + 96 * 4,
    OB = OUT + 3 * 96 * 4, TOTAL = OB + 16,
    STAGE = 256 + 2 * 64 * 32  // per wavefront: 64 gather offsets, 64 x 16 halfs hi, 64 x 16 halfs lo
  };
  static_assert(TOTAL % 16 == 0, "16-byte copies");
};
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));

// MIXED-SHAPE accumulation needs software wait states on gfx950, and ROCm 7.2's compiler does not insert them: a
// v_mfma_f32_16x16x16_f16 that accumulates into the registers a v_mfma_f32_16x16x32_f16 is still writing (or the other way
// round) reads a stale srcC unless >= 5 wait states lie between them (tests/cpp/mfma_hazard_probe.hip measures it on the
// device: probes B / B3; same-shape chains are interlocked, probe B2).  hipcc puts the two back to back.  Round 4 saw the
// wrong sums, mis-attributed them to "VALU overwrites srcC" (probe A: no such hazard) and fenced EVERY MFMA with ten wait
// states; round 5 orders each accumulator's MFMAs by shape -- all 32-wide steps, FCM_SHAPE_CHANGE(), all 16-wide steps --
// so a tile pays one wait.  scripts/mfma_hazard_check.py checks the distance in the ISA of every kernel of this file.
#ifndef FCM_NOP
#define FCM_NOP 7   // 8 wait states: >= 5 needed, measured
#endif
#define FCM_SHAPE_CHANGE()                          \
  do {                                              \
    __builtin_amdgcn_sched_barrier(0);              \
    asm volatile("s_nop %0" ::"n"(FCM_NOP));        \
    __builtin_amdgcn_sched_barrier(0);              \
  } while (0)
#define FCM_MFMA32X2(c0, c1, a, b0, b1)                                  \
  do {                                                                   \
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b0, c0, 0, 0, 0);     \
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b1, c1, 0, 0, 0);     \
  } while (0)
#define FCM_MFMA16X2(c0, c1, a, b0, b1)                                  \
  do {                                                                   \
    c0 = __builtin_amdgcn_mfma_f32_16x16x16f16(a, b0, c0, 0, 0, 0);      \
    c1 = __builtin_amdgcn_mfma_f32_16x16x16f16(a, b1, c1, 0, 0, 0);      \
  } while (0)
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3)))
fc_cost_mfma_kernel(const float* __restrict__ edges, size_t B, const half_t* __restrict__ feat, CostMapGeom g,
                    const char* __restrict__ blob, float* __restrict__ cost) {
  __shared__ __attribute__((aligned(16))) char sw[FcMfma::TOTAL];
  __shared__ __attribute__((aligned(16))) char stage_all[4][FcMfma::STAGE];
  for (int i = threadIdx.x; i < FcMfma::TOTAL / 16; i += blockDim.x)
    reinterpret_cast<uint4*>(sw)[i] = reinterpret_cast<const uint4*>(blob)[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, kg = lane >> 4;
  char* st = stage_all[wave];
  int* s_off = reinterpret_cast<int*>(st);
  half_t* s_hi = reinterpret_cast<half_t*>(st + 256);
  half_t* s_lo = reinterpret_cast<half_t*>(st + 256 + 64 * 32);
  const size_t n_chunks = (B + 63) / 64;
  for (size_t chunk = (size_t)blockIdx.x * 4 + wave; chunk < n_chunks; chunk += (size_t)gridDim.x * 4) {
    {  // one lane per edge: gather offset and the ten geometric inputs (network_light.py:118-132), hi / lo halves
      const size_t e_raw = chunk * 64 + lane;
      const size_t e = e_raw < B ? e_raw : B - 1;
      const float* ed = edges + 6 * e;
      const double tx = ed[0], ty = ed[1], tyaw = ed[2], sx = ed[3], sy = ed[4], syaw = ed[5];
      int row, col;
      cost_query_cell(g, sx, sy, &row, &col);  // cost_query.py:51-55
      s_off[lane] = (row * g.Fw + col) * 48;
      const float dx = (float)(tx - sx), dy = (float)(ty - sy);
      float dyaw = (float)(tyaw - syaw);
      const float PI = 3.14159265358979323846f;
      if (dyaw > PI) dyaw -= 2.0f * PI;
      if (dyaw < -PI) dyaw += 2.0f * PI;
      const float sya = (float)syaw;
      float t[16];
      t[0] = dx;
      t[1] = dy;
      t[2] = sqrtf(dx * dx + dy * dy);
      t[3] = atan2f(dy, dx);
      t[4] = dyaw;
      t[5] = cosf(dyaw);
      t[6] = sinf(dyaw);
      t[7] = sya;
      t[8] = cosf(sya);
      t[9] = sinf(sya);
      t[10] = 1.0f;  // the composed bias rides on this input
      t[11] = t[12] = t[13] = t[14] = t[15] = 0.0f;
      half8 h0, h1, l0, l1;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        h0[k] = (half_t)t[k];
        l0[k] = (half_t)(t[k] - (float)h0[k]);
        h1[k] = (half_t)t[8 + k];
        l1[k] = (half_t)(t[8 + k] - (float)h1[k]);
      }
      reinterpret_cast<half8*>(s_hi + lane * 16)[0] = h0;
      reinterpret_cast<half8*>(s_hi + lane * 16)[1] = h1;
      reinterpret_cast<half8*>(s_lo + lane * 16)[0] = l0;
      reinterpret_cast<half8*>(s_lo + lane * 16)[1] = l1;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll 1
    for (int pair = 0; pair < 2; ++pair) {
      // TWO 16-edge tiles per pass: a weight fragment read from LDS feeds two MFMAs.
      // The fragments are READ FROM LDS IN EVERY PASS (36 reads, 30 KB per wavefront): held in registers across the loop they
      // are 150 VGPRs and one wavefront per SIMD -- nothing to cover the feature gather with.  The opaque offset keeps the
      // compiler from hoisting them
      int woff = lane * 16;
      asm volatile("" : "+v"(woff));
      const char* wl = sw + woff;              // this lane's 16 bytes of a 1 KB fragment
      const char* wl8 = sw + (woff >> 1);      // ... 8 bytes of a 512-byte fragment
      int el[2];
      half8 x0[2], x1[2], x1lo[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        el[u] = (pair * 2 + u) * 16 + li;
        const half_t* fp = feat + s_off[el[u]];
        // operands of the first GEMM: lane = edge li, k = 32 s + 8 kg + j
        x0[u] = *reinterpret_cast<const half8*>(fp + 8 * kg);
        if (kg < 2) {
          x1[u] = *reinterpret_cast<const half8*>(fp + 32 + 8 * kg);
#pragma unroll
          for (int j = 0; j < 8; ++j) x1lo[u][j] = (half_t)0.0f;
        } else {
          x1[u] = *reinterpret_cast<const half8*>(s_hi + el[u] * 16 + 8 * (kg - 2));
          x1lo[u] = *reinterpret_cast<const half8*>(s_lo + el[u] * 16 + 8 * (kg - 2));
        }
      }
      floatx4 a1[2][3];
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        floatx4 c0 = {0.0f, 0.0f, 0.0f, 0.0f}, c1 = {0.0f, 0.0f, 0.0f, 0.0f};
        const half8 w0h = *reinterpret_cast<const half8*>(wl + FcMfma::G1 + ((0 * 3 + t) * 2 + 0) * 1024);
        const half8 w0l = *reinterpret_cast<const half8*>(wl + FcMfma::G1 + ((0 * 3 + t) * 2 + 1) * 1024);
        const half8 w1h = *reinterpret_cast<const half8*>(wl + FcMfma::G1 + ((1 * 3 + t) * 2 + 0) * 1024);
        const half8 w1l = *reinterpret_cast<const half8*>(wl + FcMfma::G1 + ((1 * 3 + t) * 2 + 1) * 1024);
        FCM_MFMA32X2(c0, c1, w0l, x0[0], x0[1]);
        FCM_MFMA32X2(c0, c1, w1l, x1[0], x1[1]);
        FCM_MFMA32X2(c0, c1, w1h, x1lo[0], x1lo[1]);
        FCM_MFMA32X2(c0, c1, w0h, x0[0], x0[1]);
        FCM_MFMA32X2(c0, c1, w1h, x1[0], x1[1]);
        a1[0][t] = c0;
        a1[1][t] = c1;
      }
      // leaky-ReLU (0.3), hi / lo halves: already in the second GEMM's operand layout (see above)
      half8 hA[2], hAlo[2];
      half4_t hB[2], hBlo[2];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float v0 = a1[u][0][j], v1 = a1[u][1][j], v2 = a1[u][2][j];
          v0 = v0 > 0.f ? v0 : 0.3f * v0;
          v1 = v1 > 0.f ? v1 : 0.3f * v1;
          v2 = v2 > 0.f ? v2 : 0.3f * v2;
          hA[u][j] = (half_t)v0;
          hAlo[u][j] = (half_t)(v0 - (float)hA[u][j]);
          hA[u][4 + j] = (half_t)v1;
          hAlo[u][4 + j] = (half_t)(v1 - (float)hA[u][4 + j]);
          hB[u][j] = (half_t)v2;
          hBlo[u][j] = (half_t)(v2 - (float)hB[u][j]);
        }
      float p[2] = {0.f, 0.f}, q[2] = {0.f, 0.f}, r[2] = {0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 6; ++t) {
        const floatx4 bias = *reinterpret_cast<const floatx4*>(sw + FcMfma::BIAS2 + (16 * t + 4 * kg) * 4);
        floatx4 c0 = bias, c1 = bias;
        const half8 wah = *reinterpret_cast<const half8*>(wl + FcMfma::G2A + (t * 2 + 0) * 1024);
        const half8 wal = *reinterpret_cast<const half8*>(wl + FcMfma::G2A + (t * 2 + 1) * 1024);
        const half4_t wbh = *reinterpret_cast<const half4_t*>(wl8 + FcMfma::G2B + (t * 2 + 0) * 512);
        const half4_t wbl = *reinterpret_cast<const half4_t*>(wl8 + FcMfma::G2B + (t * 2 + 1) * 512);
        // small terms first within a shape (lo x hi, hi x lo, then hi x hi), the 32-wide steps before the 16-wide ones
        FCM_MFMA32X2(c0, c1, wal, hA[0], hA[1]);
        FCM_MFMA32X2(c0, c1, wah, hAlo[0], hAlo[1]);
        FCM_MFMA32X2(c0, c1, wah, hA[0], hA[1]);
        FCM_SHAPE_CHANGE();
        FCM_MFMA16X2(c0, c1, wbl, hB[0], hB[1]);
        FCM_MFMA16X2(c0, c1, wbh, hBlo[0], hBlo[1]);
        FCM_MFMA16X2(c0, c1, wbh, hB[0], hB[1]);
        // head units 16 t + 4 kg + i: leaky-ReLU, then their share of the three output dot products (the output weight
        // vectors are zero outside their head: tile 0 is all energy, tile 1 half energy half time, ...)
        const int u0 = (16 * t + 4 * kg) * 4;
        floatx4 cc[2] = {c0, c1};
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
          for (int i = 0; i < 4; ++i) cc[u][i] = cc[u][i] > 0.f ? cc[u][i] : 0.3f * cc[u][i];
          if (t <= 1) {
            const floatx4 o = *reinterpret_cast<const floatx4*>(sw + FcMfma::OUT + 0 * 384 + u0);
            p[u] += cc[u][0] * o[0] + cc[u][1] * o[1] + cc[u][2] * o[2] + cc[u][3] * o[3];
          }
          if (t >= 1 && t <= 2) {
            const floatx4 o = *reinterpret_cast<const floatx4*>(sw + FcMfma::OUT + 1 * 384 + u0);
            q[u] += cc[u][0] * o[0] + cc[u][1] * o[1] + cc[u][2] * o[2] + cc[u][3] * o[3];
          }
          if (t >= 3) {
            const floatx4 o = *reinterpret_cast<const floatx4*>(sw + FcMfma::OUT + 2 * 384 + u0);
            r[u] += cc[u][0] * o[0] + cc[u][1] * o[1] + cc[u][2] * o[2] + cc[u][3] * o[3];
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        // the four lane groups of an edge hold a quarter of the units each
        float pp = p[u], qq = q[u], rr = r[u];
        pp += __shfl_xor(pp, 16, 64);
        qq += __shfl_xor(qq, 16, 64);
        rr += __shfl_xor(rr, 16, 64);
        pp += __shfl_xor(pp, 32, 64);
        qq += __shfl_xor(qq, 32, 64);
        rr += __shfl_xor(rr, 32, 64);
        const size_t e = chunk * 64 + el[u];
        if (kg == 0 && e < B) {
          const float* ob = reinterpret_cast<const float*>(sw + FcMfma::OB);
          float power = pp + ob[0], tim = qq + ob[1], prob = rr + ob[2];
          power = power > 0.f ? power : 0.f;
          tim = tim > 0.f ? tim : 0.f;
          prob = 1.0f / (1.0f + expf(-prob));
          cost[3 * e + 0] = power;
          cost[3 * e + 1] = tim;
          cost[3 * e + 2] = 1.0f - prob;  // cost_query.py:65-69 returns cost[3] = 1 - prob
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();   // the staging area is rewritten by the next chunk
  }
}

// Small batches (a roadmap update of the reference queries ~50 000 edges): one lane per edge leaves most of the GPU idle
// behind ~7 300 dependent FMAs per lane (76 us for 50 000 edges).  Here FOUR lanes share an edge: every lane forms the
// edge's 64 inputs itself (the loads hit the L1, tar0 is 160 FMAs), then computes a quarter of the 48 hidden units, and
// -- after the hidden layer went through LDS -- a quarter of the 84 head units; one lane adds the heads up in the same
// order as fc_cost_kernel.  Every unit is still accumulated by one lane over k ascending: the results are bit-identical
// to fc_cost_kernel's.  The weights are lane-varying operands now (the part index), so they come from LDS (16-byte
// reads, four distinct addresses per wavefront).  A wavefront = 16 edges x 4 parts, part = lane / 16.
#define FC_SPLIT_EDGES 64
#define FC_SPLIT_HSTRIDE 84
__global__ void __launch_bounds__(256)
fc_cost_split_kernel(const float* __restrict__ edges, size_t B, const half_t* __restrict__ feat, CostMapGeom g,
                     const float* __restrict__ wts, float* __restrict__ cost) {
  __shared__ __attribute__((aligned(16))) float w[(FcWeights::TOTAL + 3) & ~3];
  // per edge: the 48 hidden units, then -- once the four lanes hold them in registers -- the 84 head units over them
  __shared__ __attribute__((aligned(16))) float hs[FC_SPLIT_EDGES][FC_SPLIT_HSTRIDE];
  float (*as_)[FC_SPLIT_HSTRIDE] = hs;
  {
    // the blob is 16-byte aligned (hipMalloc) and the LDS copy padded to a multiple of four floats
    const float4* src = reinterpret_cast<const float4*>(wts);
    float4* dst = reinterpret_cast<float4*>(w);
    for (int i = threadIdx.x; i < FcWeights::TOTAL / 4; i += blockDim.x) dst[i] = src[i];
    for (int i = (FcWeights::TOTAL / 4) * 4 + threadIdx.x; i < FcWeights::TOTAL; i += blockDim.x) w[i] = wts[i];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int part = lane >> 4, el = wave * 16 + (lane & 15);
  const size_t e_raw = (size_t)blockIdx.x * FC_SPLIT_EDGES + el;
  const bool live = e_raw < B;
  const size_t e = live ? e_raw : B - 1;
  const float* ed = edges + 6 * e;
  const double tx = ed[0], ty = ed[1], tyaw = ed[2], sx = ed[3], sy = ed[4], syaw = ed[5];
  int row, col;
  cost_query_cell(g, sx, sy, &row, &col);
  const half_t* fp = feat + ((size_t)row * g.Fw + col) * 48;
  float x[64];
#pragma unroll
  for (int v = 0; v < 6; ++v) {
    const half8 t8 = reinterpret_cast<const half8*>(fp)[v];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[v * 8 + j] = (float)t8[j];
  }
  const float dx = (float)(tx - sx), dy = (float)(ty - sy);
  float dyaw = (float)(tyaw - syaw);
  const float PI = 3.14159265358979323846f;
  if (dyaw > PI) dyaw -= 2.0f * PI;
  if (dyaw < -PI) dyaw += 2.0f * PI;
  const float sya = (float)syaw;
  float t[10];
  t[0] = dx;
  t[1] = dy;
  t[2] = sqrtf(dx * dx + dy * dy);
  t[3] = atan2f(dy, dx);
  t[4] = dyaw;
  t[5] = cosf(dyaw);
  t[6] = sinf(dyaw);
  t[7] = sya;
  t[8] = cosf(sya);
  t[9] = sinf(sya);
#pragma unroll
  for (int o = 0; o < 16; ++o) {
    float a = w[FcWeights::TAR0_B + o];
#pragma unroll
    for (int k = 0; k < 10; ++k) a = fmaf(t[k], w[FcWeights::TAR0_W + o * 10 + k], a);
    x[48 + o] = a;
  }
  // hidden units 12 part .. 12 part + 11
  for (int oo = 0; oo < 12; ++oo) {
    const int o = part * 12 + oo;
    const float4* wr = reinterpret_cast<const float4*>(&w[FcWeights::OUT0_W + o * 64]);
    float a = w[FcWeights::OUT0_B + o];
#pragma unroll
    for (int k4 = 0; k4 < 16; ++k4) {
      const float4 q = wr[k4];
      a = fmaf(x[4 * k4 + 0], q.x, a);
      a = fmaf(x[4 * k4 + 1], q.y, a);
      a = fmaf(x[4 * k4 + 2], q.z, a);
      a = fmaf(x[4 * k4 + 3], q.w, a);
    }
    hs[el][o] = a > 0.f ? a : 0.3f * a;
  }
  wave_lds_sync();
  float h[48];
#pragma unroll
  for (int k4 = 0; k4 < 12; ++k4) {
    const float4 q = reinterpret_cast<const float4*>(&hs[el][0])[k4];
    h[4 * k4 + 0] = q.x;
    h[4 * k4 + 1] = q.y;
    h[4 * k4 + 2] = q.z;
    h[4 * k4 + 3] = q.w;
  }
  wave_lds_sync();  // every lane of the wavefront has its edge's hidden units: the buffer is free for the head units
  // head units: part p takes 6 of h1, 6 of h2, 9 of h3
  for (int u = 0; u < 21; ++u) {
    int wo, bo, slot;
    if (u < 6) {
      wo = FcWeights::H1_W + (part * 6 + u) * 48; bo = FcWeights::H1_B + part * 6 + u; slot = part * 6 + u;
    } else if (u < 12) {
      wo = FcWeights::H2_W + (part * 6 + u - 6) * 48; bo = FcWeights::H2_B + part * 6 + u - 6; slot = 24 + part * 6 + u - 6;
    } else {
      wo = FcWeights::H3_W + (part * 9 + u - 12) * 48; bo = FcWeights::H3_B + part * 9 + u - 12; slot = 48 + part * 9 + u - 12;
    }
    const float4* wr = reinterpret_cast<const float4*>(&w[wo]);
    float a = w[bo];
#pragma unroll
    for (int k4 = 0; k4 < 12; ++k4) {
      const float4 q = wr[k4];
      a = fmaf(h[4 * k4 + 0], q.x, a);
      a = fmaf(h[4 * k4 + 1], q.y, a);
      a = fmaf(h[4 * k4 + 2], q.z, a);
      a = fmaf(h[4 * k4 + 3], q.w, a);
    }
    as_[el][slot] = a > 0.f ? a : 0.3f * a;
  }
  wave_lds_sync();
  if (part != 0 || !live) return;
  float power = w[FcWeights::O1_B], tim = w[FcWeights::O2_B], prob = w[FcWeights::O3_B];
  for (int o = 0; o < 24; ++o) {
    power = fmaf(as_[el][o], w[FcWeights::O1_W + o], power);
    tim = fmaf(as_[el][24 + o], w[FcWeights::O2_W + o], tim);
  }
  for (int o = 0; o < 36; ++o) prob = fmaf(as_[el][48 + o], w[FcWeights::O3_W + o], prob);
  power = power > 0.f ? power : 0.f;
  tim = tim > 0.f ? tim : 0.f;
  prob = 1.0f / (1.0f + expf(-prob));
  cost[3 * e + 0] = power;
  cost[3 * e + 1] = tim;
  cost[3 * e + 2] = 1.0f - prob;
}

}  // namespace artp

// Built-and-measured forms that did not become the default (profiles/r05_cnn_variants.txt, r06_cnn_variants.txt): only in
// libartp_variants.so (make -C art_planner_amd/csrc variants), where the tests that pin their parity run them.
#ifdef ARTP_VARIANTS
#include "cost_kernels_variants.h"
#endif
