// cost_kernels.h -- learned motion cost on gfx950.
//
// R8  network.CNNpart (art_planner_motion_cost/.../predictor/network_light.py:78-110): six un-padded
//     convolutions with eval-mode BatchNorm, leaky-ReLU(0.3) and two max-pools, fp16.  BatchNorm is
//     folded into the weights/bias on the host.  Activations live in HBM as NHWC fp16, so for a fixed
//     kernel row the (kw, cin) taps of an output pixel are ONE contiguous run of KW*Cin halfs: the
//     implicit GEMM walks K = (kh, [kw, cin]) in 32-wide steps that are single 16-byte loads per lane
//     and feeds v_mfma_f32_16x16x32_f16 (fp32 accumulate).  Weights are pre-packed on the host in
//     exactly the B-fragment order.  The first layer (Cin = 1, K = 9) is plain VALU.
// R9  CostQuery.__call__ + network.FCpart (cost_query.py:39-69, network_light.py:113-165): per edge,
//     gather the 48 features of the start cell, build the 10 geometric inputs, 1x1-conv MLP with three
//     heads -> (energy, time, 1 - prob).  One lane per edge, fp32 math, weights broadcast from LDS.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace artp {

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

// ---- layer 1: 1 -> 24 channels, 3x3, BN folded, no activation (network_light.py:84-85) --------------
// in: [H][W] fp16; out: NHWC [H-2][W-2][24] fp16.  One lane per output pixel.
__global__ void __launch_bounds__(256)
conv1_kernel(const half_t* __restrict__ in, int H, int W, const float* __restrict__ w /*[24][9]*/,
             const float* __restrict__ bias /*[24]*/, half_t* __restrict__ out) {
  __shared__ float sw[24 * 9 + 24];
  for (int i = threadIdx.x; i < 24 * 9 + 24; i += blockDim.x) sw[i] = i < 216 ? w[i] : bias[i - 216];
  __syncthreads();
  const int Ho = H - 2, Wo = W - 2;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= Ho * Wo) return;
  const int oy = p / Wo, ox = p - oy * Wo;
  float x[9];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) x[ky * 3 + kx] = (float)in[(oy + ky) * W + ox + kx];
  half_t o[24];
#pragma unroll
  for (int c = 0; c < 24; ++c) {
    float a = sw[216 + c];
#pragma unroll
    for (int k = 0; k < 9; ++k) a = fmaf(x[k], sw[c * 9 + k], a);
    o[c] = (half_t)a;
  }
  half8* dst = reinterpret_cast<half8*>(out + (size_t)p * 24);
#pragma unroll
  for (int v = 0; v < 3; ++v) {
    half8 t;
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = o[v * 8 + j];
    dst[v] = t;
  }
}

// ---- generic un-padded convolution as implicit GEMM on the matrix cores -------------------------------
// in : NHWC fp16 [Hin][Win][CIN]           (CIN = 24 or 48; rows of KW*CIN halfs are contiguous)
// wp : packed fp16 B fragments [KH][KSTEPS][NT][64 lanes][8]   (K padded with zeros to 32*KSTEPS)
// out: NHWC fp16 [Hout][Wout][COUT], y = lrelu?(acc + bias)
// One wavefront computes MT x 16 consecutive output pixels of one row for all NT*16 (>= COUT) channels.
template <int KH, int KW, int CIN, int COUT, int NT, int MT, bool LRELU>
__global__ void __launch_bounds__(256)
conv_mfma_kernel(const half_t* __restrict__ in, int Hin, int Win, const half8* __restrict__ wp,
                 const float* __restrict__ bias, half_t* __restrict__ out) {
  constexpr int KROW = KW * CIN;             // contiguous K run per kernel row
  constexpr int KSTEPS = (KROW + 31) / 32;   // 32-wide MFMA steps per kernel row
  const int Hout = Hin - KH + 1, Wout = Win - KW + 1;
  const int tiles_x = (Wout + 16 * MT - 1) / (16 * MT);
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (wave >= tiles_x * Hout) return;
  const int oy = wave / tiles_x;
  const int ox0 = (wave - oy * tiles_x) * 16 * MT;
  const int li = lane & 15, kg = lane >> 4;

  floatx4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = floatx4{0.f, 0.f, 0.f, 0.f};

  for (int kh = 0; kh < KH; ++kh) {
    const half_t* row = in + ((size_t)(oy + kh) * Win + ox0) * CIN + kg * 8;
    const half8* wrow = wp + (size_t)kh * KSTEPS * NT * 64 + lane;
#pragma unroll 2
    for (int ks = 0; ks < KSTEPS; ++ks) {
      half8 a[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m)
        a[m] = *reinterpret_cast<const half8*>(row + (size_t)(m * 16 + li) * CIN + ks * 32);
      half8 b[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) b[n] = wrow[(size_t)(ks * NT + n) * 64];
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[m], b[n], acc[m][n], 0, 0, 0);
    }
  }
  // C/D layout of 16x16 MFMA: column (channel in tile) = lane & 15, row (pixel) = (lane >> 4) * 4 + r
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int ch = n * 16 + li;
    if (ch >= COUT) continue;
    const float bv = bias[ch];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ox = ox0 + m * 16 + kg * 4 + r;
        if (ox < Wout) {
          float v = acc[m][n][r] + bv;
          if (LRELU) v = v > 0.f ? v : 0.3f * v;
          out[((size_t)oy * Wout + ox) * COUT + ch] = (half_t)v;
        }
      }
  }
}

// ---- the same implicit GEMM for large kernels (the 15 x 15 flatten layer): LDS-staged input patch ---------
// A workgroup of 4 wavefronts computes an 8-row x 16-pixel output tile for all channels from a
// (8+KH-1) x (16+KW) pixel input patch staged once in LDS (NHWC, 2*CIN bytes per pixel: with CIN = 48 the 16
// lanes of a ds_read_b128 group fall into 16 distinct bank quads, no padding needed; PMC: 0 conflicts).
template <int KH, int KW, int CIN, int COUT, int NT, bool LRELU>
struct ConvLdsCfg {
  static constexpr int KROW = KW * CIN;
  static constexpr int KSTEPS = (KROW + 31) / 32;
  static constexpr int TR = 8, TP = 16;
  static constexpr int PR = TR + KH - 1;
  static constexpr int XPAD = (KSTEPS * 32 - KROW + CIN - 1) / CIN;  // pixels the zero-padded K tail reaches into
  static constexpr int PPX = TP + KW - 1 + XPAD;
  static constexpr int PIX_B = CIN * 2;
  static constexpr int ROW_B = PPX * PIX_B;
  static constexpr int A_BYTES = PR * ROW_B;
  static_assert(ROW_B % 16 == 0 && A_BYTES % 16 == 0, "16-byte chunks");
};

// ---- K-split variant: the workgroup's four wavefronts split the k-steps, not the pixels -------------------
// Every wavefront accumulates the WHOLE 8-row x 16-pixel x COUT tile (8 x NT accumulators) over the
// k-steps ks = wave, wave+4, ... of every kernel row; the four partial tiles meet in LDS at the end.
//  * B fragments are used by exactly one wavefront of the group: they go global -> VGPR (prefetched two
//    steps ahead), never through LDS: no staging stores, no barrier inside the main loop.
//  * For a fixed ks the A fragments of kernel row kh are patch rows kh .. kh+7 -- kernel row kh+1 needs
//    ONE new row.  A register ring of 8 fragments turns 8 LDS reads per step into 1.
//  * Per step: 1 ds_read_b128 + NT global loads feed 8*NT MFMAs (384 cycles for NT = 3).
template <int KH, int KW, int CIN, int COUT, int NT, bool LRELU>
struct ConvKsplitCfg {
  using P = ConvLdsCfg<KH, KW, CIN, COUT, NT, LRELU>;
  static constexpr int RED_BYTES = 4 * P::TR * NT * 1024;  // four partial tiles of floatx4 per lane
  static constexpr int STAGE_BYTES = P::TR * P::TP * COUT * 2;  // finished NHWC tile
  static constexpr int LDS_BYTES = (P::A_BYTES > RED_BYTES ? P::A_BYTES : RED_BYTES) + STAGE_BYTES;
  static_assert((COUT * 2) % 16 == 0, "a pixel is a whole number of 16-byte chunks");
};

template <int KH, int KW, int CIN, int COUT, int NT, bool LRELU>
__global__ void __launch_bounds__(256)
conv_ksplit_kernel(const half_t* __restrict__ in, int Hin, int Win, const half8* __restrict__ wp,
                   const float* __restrict__ bias, half_t* __restrict__ out) {
  using Cfg = ConvLdsCfg<KH, KW, CIN, COUT, NT, LRELU>;
  static_assert(Cfg::TR == 8, "the fragment ring below holds 8 rows");
  constexpr int KSTEPS = Cfg::KSTEPS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* As = smem;
  const int Hout = Hin - KH + 1, Wout = Win - KW + 1;
  const int tiles_x = (Wout + Cfg::TP - 1) / Cfg::TP;
  const int bx = blockIdx.x % tiles_x, by = blockIdx.x / tiles_x;
  const int oy0 = by * Cfg::TR, ox0 = bx * Cfg::TP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, kg = lane >> 4;

  // input patch -> LDS (rows are contiguous byte runs of the NHWC image; out-of-image chunks are zero).
  // All of a thread's loads are in flight before the first LDS store (one memory round trip, not 16).
  {
    constexpr int CPR = Cfg::ROW_B / 16;  // 16-byte chunks per patch row
    constexpr int NIT = (Cfg::PR * CPR + 255) / 256;
    const long row_bytes = (long)Win * Cfg::PIX_B;
    half8 v[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = tid + it * 256;
      const int r = c / CPR, cc = c - r * CPR;
      const long off = (long)ox0 * Cfg::PIX_B + (long)cc * 16;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[it][j] = (half_t)0;
      if (c < Cfg::PR * CPR && oy0 + r < Hin && off + 16 <= row_bytes)
        v[it] = *reinterpret_cast<const half8*>(reinterpret_cast<const char*>(in) + (long)(oy0 + r) * row_bytes + off);
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = tid + it * 256;
      const int r = c / CPR, cc = c - r * CPR;
      if (c < Cfg::PR * CPR) *reinterpret_cast<half8*>(As + r * Cfg::ROW_B + cc * 16) = v[it];
    }
  }
  __syncthreads();

  floatx4 acc[8][NT];
#pragma unroll
  for (int m = 0; m < 8; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = floatx4{0.f, 0.f, 0.f, 0.f};

  const char* a_lane = As + li * Cfg::PIX_B + kg * 16;
  for (int ks = wave; ks < KSTEPS; ks += 4) {
    const char* a_ks = a_lane + ks * 64;
    const half8* b_ks = wp + (size_t)ks * NT * 64 + lane;  // + kh * KSTEPS * NT * 64
    half8 a[8];      // ring: slot (r & 7) holds patch row r
    half8 b[3][NT];  // ring over kernel rows, two ahead
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      b[0][n] = b_ks[n * 64];
      if (KH > 1) b[1][n] = b_ks[(size_t)KSTEPS * NT * 64 + n * 64];
    }
#pragma unroll
    for (int r = 0; r < 7; ++r) a[r] = *reinterpret_cast<const half8*>(a_ks + r * Cfg::ROW_B);
#pragma unroll
    for (int kh = 0; kh < KH; ++kh) {
      a[(kh + 7) & 7] = *reinterpret_cast<const half8*>(a_ks + (kh + 7) * Cfg::ROW_B);
      if (kh + 2 < KH) {
#pragma unroll
        for (int n = 0; n < NT; ++n) b[(kh + 2) % 3][n] = b_ks[(size_t)(kh + 2) * KSTEPS * NT * 64 + n * 64];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < 8; ++m)  // m = 7 uses the row requested just above: it goes last
#pragma unroll
        for (int n = 0; n < NT; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(kh + m) & 7], b[kh % 3][n], acc[m][n], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // the four partial tiles -> LDS (the patch is dead), wavefront w finishes rows 2w, 2w+1
  __syncthreads();
  floatx4* red = reinterpret_cast<floatx4*>(smem);
#pragma unroll
  for (int m = 0; m < 8; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) red[((wave * 8 + m) * NT + n) * 64 + lane] = acc[m][n];
  __syncthreads();
  // C/D layout of 16x16 MFMA: column (channel in tile) = lane & 15, row (pixel) = (lane >> 4) * 4 + r.
  // The finished halfs are staged as the NHWC tile [8][16][COUT] behind the partial tiles, then leave as
  // whole 16-byte lanes (a tile row is one contiguous 16*COUT*2-byte run of the output image).
  half_t* stage = reinterpret_cast<half_t*>(smem + ConvKsplitCfg<KH, KW, CIN, COUT, NT, LRELU>::RED_BYTES);
#pragma unroll
  for (int mm = 0; mm < 2; ++mm) {
    const int m = 2 * wave + mm;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      floatx4 v = red[((0 * 8 + m) * NT + n) * 64 + lane];
#pragma unroll
      for (int w = 1; w < 4; ++w) {
        const floatx4 p = red[((w * 8 + m) * NT + n) * 64 + lane];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += p[r];
      }
      const int ch = n * 16 + li;
      if (ch >= COUT) continue;
      const float bv = bias[ch];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float y = v[r] + bv;
        if (LRELU) y = y > 0.f ? y : 0.3f * y;
        stage[(m * 16 + kg * 4 + r) * COUT + ch] = (half_t)y;
      }
    }
  }
  __syncthreads();
  {
    constexpr int CPR = 16 * COUT * 2 / 16;  // 16-byte chunks per tile row
    for (int c = tid; c < 8 * CPR; c += 256) {
      const int m = c / CPR, cc = c - m * CPR;
      const int px = (cc * 16) / (COUT * 2);
      if (oy0 + m < Hout && ox0 + px < Wout)
        *reinterpret_cast<half8*>(reinterpret_cast<char*>(out) + ((size_t)(oy0 + m) * Wout + ox0) * COUT * 2 + cc * 16) =
            *reinterpret_cast<const half8*>(reinterpret_cast<const char*>(stage) + m * CPR * 16 + cc * 16);
    }
  }
}

// max_pool2d NHWC, window P, stride S (network_light.py:89,97)
template <int P, int S>
__global__ void __launch_bounds__(256)
maxpool_kernel(const half_t* __restrict__ in, int Hin, int Win, int C, half_t* __restrict__ out) {
  const int Ho = (Hin - P) / S + 1, Wo = (Win - P) / S + 1;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)Ho * Wo * C) return;
  const int c = (int)(i % C);
  const size_t p = i / C;
  const int ox = (int)(p % Wo), oy = (int)(p / Wo);
  float m = -INFINITY;
#pragma unroll
  for (int dy = 0; dy < P; ++dy)
#pragma unroll
    for (int dx = 0; dx < P; ++dx) {
      const float v = (float)in[((size_t)(oy * S + dy) * Win + ox * S + dx) * C + c];
      m = v > m ? v : m;
    }
  out[i] = (half_t)m;
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
  int F;              // feature map is F x F
  double feat_res;    // res * featureResDownsampleFactor (cost_query.py:33)
  int row_bias, col_bias;  // cost_query.py:34-35
  double cx, cy;      // map centre subtracted by the server (cost_query_server.py:134-135,160-161)
};

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
  // cost_query.py:51-55 (double arithmetic: the server hands float64 numpy to torch)
  double pr = (sx - g.cx) / g.feat_res + (double)g.row_bias;
  double pc = (sy - g.cy) / g.feat_res + (double)g.col_bias;
  pr = pr < 1.0 ? 1.0 : (pr > (double)(g.F - 2) ? (double)(g.F - 2) : pr);
  pc = pc < 1.0 ? 1.0 : (pc > (double)(g.F - 2) ? (double)(g.F - 2) : pc);
  const int row = (int)pr, col = (int)pc;
  const half_t* fp = feat + ((size_t)row * g.F + col) * 48;
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

__global__ void __launch_bounds__(256)
f32_to_f16_kernel(const float* __restrict__ in, size_t n, half_t* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (half_t)in[i];
}

}  // namespace artp
