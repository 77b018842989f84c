// cost_kernels_variants.h -- forms of the motion-cost kernels that were built, pinned for parity and measured, and did NOT
// become the default.  Not part of libartp.so: `make -C art_planner_amd/csrc variants` builds libartp_variants.so with
// -DARTP_VARIANTS, which compiles these and the environment switches that select them (artp_capi.hip).  The numbers are in
// profiles/r05_cnn_variants.txt, r05_conv12_ab.txt and r06_cnn_variants.txt.
#pragma once

namespace artp {

constexpr int C12_PT = 8;              // pooled pixels per tile edge: one wavefront's 64 lanes
constexpr int C12_IN = 2 * C12_PT + 4; // input window of a tile
constexpr int C12_RS = 24;             // LDS row stride in floats: the four pooled rows of a 32-lane group land 16 banks apart
constexpr int C12_CG = 6;              // channels per wavefront (4 wavefronts x 6 = 24)

// Workgroup = an 8 x 8 tile of pooled pixels; wavefront w computes channels 6 w .. 6 w + 5 of all 64 (weights are
// wave-uniform), lane = pooled pixel.  625 workgroups at C3, 2500 at C4: several wavefronts per SIMD hide the scalar
// weight loads that one big tile per CU (first version: 12.8 us at C3) left exposed.
__global__ void __launch_bounds__(256)
conv12_pool_kernel(const float* __restrict__ in, int H, int W, const float* __restrict__ w /*[24][25]*/,
                   const float* __restrict__ bias /*[24]*/, half_t* __restrict__ out /*[hp][wp][24]*/) {
  __shared__ float tile[C12_IN * C12_RS];
  const int hp = (H - 4) / 2, wp = (W - 4) / 2;
  const int tiles_x = (wp + C12_PT - 1) / C12_PT;
  const int bx = blockIdx.x % tiles_x, by = blockIdx.x / tiles_x;
  const int iy0 = by * 2 * C12_PT, ix0 = bx * 2 * C12_PT;
  const int tid = threadIdx.x;
  for (int i = tid; i < C12_IN * C12_IN; i += 256) {
    const int r = i / C12_IN, c = i - r * C12_IN;
    const int y = iy0 + r, x = ix0 + c;
    tile[r * C12_RS + c] = (y < H && x < W) ? (float)(half_t)in[(size_t)y * W + x] : 0.0f;
  }
  __syncthreads();
  const int lane = tid & 63;
  const int cg = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform channel group: weight addresses stay scalar
  const int py = lane >> 3, px = lane & 7;
  // The two conv outputs of a row pair share a weight: packed f32 FMAs (v_pk_fma_f32, two lanes of math per issue
  // slot; the kernel is bound by VALU issue) on (x[kx], x[kx+1]) pairs.  xe[r][i] = (x[r][2i], x[r][2i+1]) come straight
  // from the 8-byte LDS reads, xo[r][i] = (x[r][2i+1], x[r][2i+2]) are the odd-aligned pairs.
  typedef float float2_t __attribute__((ext_vector_type(2)));
  float2_t xe[6][3], xo[6][2];
#pragma unroll
  for (int r = 0; r < 6; ++r) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
      xe[r][i] = *reinterpret_cast<const float2_t*>(&tile[(2 * py + r) * C12_RS + 2 * px + 2 * i]);
#pragma unroll
    for (int i = 0; i < 2; ++i) xo[r][i] = float2_t{xe[r][i][1], xe[r][i + 1][0]};
  }
  const float* __restrict__ wg = w + cg * C12_CG * 25;
  const float* __restrict__ bg = bias + cg * C12_CG;
  half_t o[C12_CG];
#pragma unroll
  for (int co = 0; co < C12_CG; ++co) {
    const float b = bg[co];
    float2_t a0 = float2_t{b, b}, a1 = float2_t{b, b};   // (a00, a01), (a10, a11)
#pragma unroll
    for (int ky = 0; ky < 5; ++ky)
#pragma unroll
      for (int kx = 0; kx < 5; ++kx) {
        const float wv = wg[co * 25 + ky * 5 + kx];  // wave-uniform: a scalar load, an SGPR operand
        const float2_t w2 = float2_t{wv, wv};
        const float2_t p0 = (kx & 1) ? xo[ky][kx >> 1] : xe[ky][kx >> 1];
        const float2_t p1 = (kx & 1) ? xo[ky + 1][kx >> 1] : xe[ky + 1][kx >> 1];
        a0 = __builtin_elementwise_fma(p0, w2, a0);
        a1 = __builtin_elementwise_fma(p1, w2, a1);
      }
    float m = fmaxf(fmaxf(a0[0], a0[1]), fmaxf(a1[0], a1[1]));
    m = m > 0.f ? m : 0.3f * m;
    o[co] = (half_t)m;
  }
  const int gy = by * C12_PT + py, gx = bx * C12_PT + px;
  if (gy < hp && gx < wp) {
    // 6 halfs = 12 bytes at byte offset 12 cg of the pixel's 48: three dword stores
    unsigned* dst = reinterpret_cast<unsigned*>(out + ((size_t)gy * wp + gx) * 24 + cg * C12_CG);
#pragma unroll
    for (int v = 0; v < 3; ++v) {
      typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
      half2_t t;
      t[0] = o[2 * v];
      t[1] = o[2 * v + 1];
      dst[v] = __builtin_bit_cast(unsigned, t);
    }
  }
}

__global__ void __launch_bounds__(256)
conv12_mfma_kernel(const float* __restrict__ in, int H, int W, const half8* __restrict__ wfrag /*[2][2][64]*/,
                   const float* __restrict__ bias /*[32]*/, half_t* __restrict__ out /*[hp][wp][24]*/) {
  __shared__ unsigned patch[2 * C12M_IH * C12M_RS];
  half_t* const p0 = reinterpret_cast<half_t*>(patch);
  half_t* const p1 = reinterpret_cast<half_t*>(patch + C12M_IH * C12M_RS);
  const int hp = (H - 4) / 2, wp = (W - 4) / 2;
  const int tiles_x = (wp + C12M_PX - 1) / C12M_PX;
  const int bx = blockIdx.x % tiles_x, by = blockIdx.x / tiles_x;
  const int iy0 = by * 2 * C12M_PY, ix0 = bx * 2 * C12M_PX;
  const int tid = threadIdx.x;
  {
    constexpr int NEL = C12M_IH * 2 * C12M_RS, NIT = (NEL + 255) / 256;
    float v[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {  // unconditional loads from clamped addresses: conditional ones are serialised (conv345_kernel)
      const int i = tid + it * 256;
      const int r = i / (2 * C12M_RS), c = i - r * (2 * C12M_RS);
      const int y = iy0 + r, x = ix0 + c;
      const bool ok = i < NEL && y < H && x < W;
      const float t = in[ok ? (size_t)y * W + x : (size_t)0];
      v[it] = ok ? t : 0.0f;
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = tid + it * 256;
      const int r = i / (2 * C12M_RS), c = i - r * (2 * C12M_RS);
      if (i < NEL) {
        const half_t h = (half_t)v[it];
        p0[r * 2 * C12M_RS + c] = h;
        if (c > 0) p1[r * 2 * C12M_RS + c - 1] = h;
      }
    }
  }
  const int lane = tid & 63, wv = tid >> 6;
  C12Frag f;
  c12_frag_load(f, wfrag, bias, lane, C12M_RS);
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int pyl = 2 * wv + q;
    half4_t o0, o1;
    conv12_pooled16<C12M_RS, C12M_IH * C12M_RS>(patch, 2 * pyl * C12M_RS + (lane & 15), f, o0, o1);
    const int gy = by * C12M_PY + pyl, gx = bx * C12M_PX + (lane & 15), g = lane >> 4;
    if (gy < hp && gx < wp) {
      half_t* dst = out + ((size_t)gy * wp + gx) * 24;
      *reinterpret_cast<half4_t*>(dst + 4 * g) = o0;
      if (g < 2) *reinterpret_cast<half4_t*>(dst + 16 + 4 * g) = o1;
    }
  }
}

// ---- round 5: the 15 x 15 layer as a PERSISTENT kernel that walks down 16-pixel column strips -------------------------
// One workgroup per CU for the whole launch; it owns a run of vertically adjacent TR-row x 16-pixel tiles.
//  * The patch rows live in an LDS RING of PR + TR rows: the tile below needs only TR new rows (24-27 KB instead of the
//    65 KB patch), and they are fetched while the current tile's MFMAs run -- by the wavefronts that own the SHORT K
//    slice (23 k-steps over 4 slices = 6, 6, 6, 5: those three wavefronts are done a sixth early and would only wait at
//    the barrier).  Their loads are not in the main loop's vmcnt stream (loads retire in order: a slow row fetch
//    in front of the B ring would stall every MFMA step behind it).
//  * 12 wavefronts = 3 (channel tiles, N) x 4 (K slices): a wavefront accumulates TR x ONE channel tile, so per k-step it
//    needs one B fragment (global -> VGPR ring, a whole 15-row k-step column ahead, never drains -- not even across tiles:
//    the weights are the same; the order of the k-steps rotates with the workgroup) and one new A row (LDS, register ring over the kernel rows as in conv_ksplit_kernel) for TR MFMAs.  Only the
//    4 K slices meet in LDS at the end of a tile: 96-108 KB of partial sums per tile against 192 KB for an 8-way K split
//    (LDS stores run at ~79 B/clk/CU: the 8-way reduction cost a seventh of a tile's MFMA time).
//  * Three wavefronts per SIMD (<= 168 registers): one's LDS / global latency and the reduction's barriers run under the
//    others' MFMAs.
//  * The finished sums leave as 8-byte stores (four consecutive channels of one pixel: the transposed product).
template <int TR, int BD_ = 5>
struct KwalkCfg {
  using P = ConvLdsCfg<15, 15, 48, 48, 3, true, TR>;
  static constexpr int NK = 4, NN = 3, NWV = NK * NN, NTH = 64 * NWV;
  static constexpr int KSTEPS = P::KSTEPS, PR = P::PR, RING = PR + TR, ROW_B = P::ROW_B, CPR = ROW_B / 16;
  static constexpr int RING_B = RING * ROW_B;
  static constexpr int RPP_MAX = (160 * 1024 - RING_B) / (NWV * 1024);   // rows of partial sums that fit beside the ring
  static constexpr int NPASS = (TR + RPP_MAX - 1) / RPP_MAX;
  static constexpr int RPP = (TR + NPASS - 1) / NPASS;
  static constexpr int BIAS_OFF = RING_B + NWV * RPP * 1024;
  static constexpr int LDS_BYTES = BIAS_OFF + 256;   // + the layer's 48 biases (the reduction reads them per element)
  static constexpr int BD = BD_;      // B ring depth: BD - 1 steps ahead; 15 kernel rows = a whole number of turns, so slot = kh % BD
  static_assert(15 % BD == 0, "slot = kh % BD across k-steps");
  static_assert(KSTEPS == 23 && KSTEPS % NK == 3, "the last K slice is the short one (it fetches the next tile's rows)");
  static_assert(LDS_BYTES <= 160 * 1024 && RPP >= 1, "one workgroup per CU");
  static_assert(RING_B % 16 == 0, "16-byte chunks");
};


template <int TR, int BD_ = 5, bool EARLY = true>
__global__ void __launch_bounds__(768, 3)
conv_kwalk_kernel(const half_t* __restrict__ in, int Hin, int Win, const half8* __restrict__ wp,
                  const float* __restrict__ bias, half_t* __restrict__ out, int tiles_per_strip, int n_tiles) {
  using Cfg = KwalkCfg<TR, BD_>;
  constexpr int KSTEPS = Cfg::KSTEPS, PR = Cfg::PR, RING = Cfg::RING, ROW_B = Cfg::ROW_B, CPR = Cfg::CPR, NTH = Cfg::NTH;
  constexpr int BD = Cfg::BD, KH = 15;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ring = smem;
  floatx4* red = reinterpret_cast<floatx4*>(smem + Cfg::RING_B);
  const int Hout = Hin - 14, Wout = Win - 14;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kg = lane >> 4;
  const int nt = wave % 3, kq = wave / 3;     // waves {s, s+4, s+8} share a SIMD: every SIMD gets all three channel tiles'
                                              // worth of different K slices (18 / 17 / 17 / 17 k-step units)
  // this workgroup's run of tiles (strip-major: consecutive tiles are vertically adjacent).  Workgroups that share an XCD
  // (blockIdx % 8) take neighbouring runs: their halo rows meet in that XCD's L2.
  const int G = (int)gridDim.x;
  const int b = xcd_contiguous((int)blockIdx.x, G);
  const int t0 = (int)(((long)b * n_tiles) / G), t1 = (int)(((long)(b + 1) * n_tiles) / G);
  const long row_bytes = (long)Win * 96;
#ifdef ARTP_STAGE_TIMING
  long long t_prev = clock64();
  unsigned long long kw_c[4] = {0, 0, 0, 0};
#endif

  // B ring (this wavefront's channel tile): step = (jj, kh) with ks = kq + 4 ((jj + j0) mod nj); slot = kh.  The k-step a
  // workgroup starts with rotates with its index: the workgroups of a launch stream the same 1 MB of fragments, no two
  // neighbours in the same order.
  const int nj = (KSTEPS - kq + 3) / 4;
  const int j0 = (int)(blockIdx.x % (unsigned)nj);
  auto ks_of = [&](int jj) {
    const int j = jj + j0 < nj ? jj + j0 : jj + j0 - nj;
    return kq + 4 * j;
  };
  // the fragment's address = a wave-uniform pointer (scalar registers) + the lane's 16 bytes: no 64-bit pointer per slot
  auto b_ptr = [&](int ks, int kh) { return wp + ((size_t)(kh * KSTEPS + ks) * 3 + nt) * 64 + lane; };
  half8 bq[BD];
  {
    const int ks0 = ks_of(0);
#pragma unroll
    for (int s = 0; s < BD - 1; ++s) bq[s] = *b_ptr(ks0, s);
  }
  if (tid < 48) reinterpret_cast<float*>(smem + Cfg::BIAS_OFF)[tid] = bias[tid];

  floatx4 acc[TR];
  const int a_lane = li * 96 + kg * 16;
  int base = 0;   // ring slot of the current tile's patch row 0
  for (int t = t0; t < t1; ++t) {
    const int strip = t / tiles_per_strip, rt = t - strip * tiles_per_strip;
    const int oy0 = rt * TR, ox0 = strip * 16;
    if (t == t0 || rt == 0) {
      // first tile of the run / of a strip: the whole patch (rows are contiguous byte runs of the NHWC image; zeros
      // outside it).  Every wavefront is past the previous tile's reduction barriers: the ring is free.
      constexpr int NIT = (PR * CPR + NTH - 1) / NTH;
      half8 v[NIT];
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int c = tid + it * NTH;
        const int r = c / CPR, cc = c - r * CPR;
        const long off = (long)ox0 * 96 + (long)cc * 16;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[it][j] = (half_t)0;
        if (c < PR * CPR && oy0 + r < Hin && off + 16 <= row_bytes)
          v[it] = *reinterpret_cast<const half8*>(reinterpret_cast<const char*>(in) + (long)(oy0 + r) * row_bytes + off);
      }
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int c = tid + it * NTH;
        if (c < PR * CPR) *reinterpret_cast<half8*>(ring + c * 16) = v[it];
      }
      base = 0;
      __syncthreads();
      ARTP_KW_MARK(0);
    }
    const bool has_next = t + 1 < t1 && rt + 1 < tiles_per_strip;
#pragma unroll
    for (int m = 0; m < TR; ++m) acc[m] = floatx4{0.f, 0.f, 0.f, 0.f};
    auto row_ptr = [&](int r) {   // patch row r of the current tile (r is a compile-time constant at every use)
      const int s = base + r;
      return ring + (s >= RING ? s - RING : s) * ROW_B + a_lane;
    };
    // The TR rows the tile below adds (patch rows PR .. PR + TR - 1 -> the ring slots the current tile does not use) are
    // fetched by the short K slice's three wavefronts.  EARLY: as LDS-DMA (global_load_lds_dwordx4: no registers, 1 KB per
    // wave-instruction, three per row) issued at the START of the tile -- a CU fills from the Infinity Cache at only
    // ~4 B/clk when every CU does (27 KB = 16 k cycles, phase counters), and loads retire in order, so the wavefront parks
    // at its first B fragment younger than the DMAs while the SIMD's other two wavefronts keep the matrix pipe busy; it
    // has a column less to do.  Chunks outside the image are zeroed by ordinary stores.  !EARLY: through registers, after
    // the wavefront's last column.
    constexpr int NLR = 192, NITR = (TR * CPR + NLR - 1) / NLR;
    const int t3 = nt * 64 + lane;
    auto rows_dma = [&]() {
#pragma unroll
      for (int i0 = 0; i0 < TR * 3; i0 += 3) {
        const int i = i0 + nt;               // (row, 1 KB segment) pairs dealt to the three wavefronts
        const int r = i / 3, seg = i - r * 3;
        const int c = seg * 64 + lane;
        const int sl = base + PR + r;
        char* dst = ring + (sl >= RING ? sl - RING : sl) * ROW_B + seg * 1024;
        const long off = (long)ox0 * 96 + (long)c * 16;
        const bool in_img = oy0 + PR + r < Hin && off + 16 <= row_bytes;
        if (c < CPR) {
          if (in_img)
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(in) + (long)(oy0 + PR + r) * row_bytes + off),
                (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
          else
            *reinterpret_cast<half8*>(dst + lane * 16) = half8{0, 0, 0, 0, 0, 0, 0, 0};
        }
      }
    };
    auto rows_through_registers = [&]() {
      half8 vr[NITR];
#pragma unroll
      for (int it = 0; it < NITR; ++it) {
        const int c = t3 + it * NLR;
        const int r = c / CPR, cc = c - r * CPR;
        const long off = (long)ox0 * 96 + (long)cc * 16;
#pragma unroll
        for (int j = 0; j < 8; ++j) vr[it][j] = (half_t)0;
        if (c < TR * CPR && oy0 + PR + r < Hin && off + 16 <= row_bytes)
          vr[it] = *reinterpret_cast<const half8*>(reinterpret_cast<const char*>(in) + (long)(oy0 + PR + r) * row_bytes + off);
      }
#pragma unroll
      for (int it = 0; it < NITR; ++it) {
        const int c = t3 + it * NLR;
        const int r = c / CPR, cc = c - r * CPR;
        const int s = base + PR + r;
        if (c < TR * CPR) *reinterpret_cast<half8*>(ring + (s >= RING ? s - RING : s) * ROW_B + cc * 16) = vr[it];
      }
    };
    const bool fetch = kq == 3 && has_next;
    if (EARLY && fetch) rows_dma();
    for (int jj = 0; jj < nj; ++jj) {
      const int ks = ks_of(jj);
      const int ks_nx = ks_of(jj + 1 < nj ? jj + 1 : 0);   // wraps into the next tile's first k-step: the same weights
      // A ring of TR + 1 fragments: slot (r % (TR + 1)) holds patch row r; a row is requested a whole step before the MFMA
      // that needs it (with TR = 9 MFMAs per step the row asked for at the head of a step came back ~130 cycles later, right
      // when the step's last MFMA wanted it: LDS latency with 12 wavefronts reading was the stall)
      constexpr int AR = TR + 1;
      half8 a[AR];
#pragma unroll
      for (int r = 0; r < TR; ++r) a[r] = *reinterpret_cast<const half8*>(row_ptr(r) + ks * 64);
#pragma unroll
      for (int kh = 0; kh < KH; ++kh) {
        if (kh + TR < PR) a[(kh + TR) % AR] = *reinterpret_cast<const half8*>(row_ptr(kh + TR) + ks * 64);
        bq[(kh + BD - 1) % BD] = kh + BD - 1 < KH ? *b_ptr(ks, kh + BD - 1) : *b_ptr(ks_nx, kh + BD - 1 - KH);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < TR; ++m)
          acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bq[kh % BD], a[(kh + m) % AR], acc[m], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    ARTP_KW_MARK(1);
    if (!EARLY && fetch) rows_through_registers();
    // The four K slices of a channel tile meet in LDS, RPP rows at a time; element e of a pass = (row j, channel tile n,
    // lane l) is summed over the slices in a fixed order by thread e (mod NTH), gets bias + leaky-ReLU and leaves as four
    // consecutive channels of one pixel (transposed product: column l & 15 = pixel, row 4 (l >> 4) + r = channel).
    typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int h = 0; h < Cfg::NPASS; ++h) {
      constexpr int RPP = Cfg::RPP;
      const int rows = (h + 1) * RPP <= TR ? RPP : TR - h * RPP;
      __syncthreads();   // h = 0: every wavefront has left the main loop and the new rows are in the ring
      if (h == 0) ARTP_KW_MARK(2);
#pragma unroll
      for (int j = 0; j < RPP; ++j)
        if (h * RPP + j < TR) red[(wave * RPP + j) * 64 + lane] = acc[h * RPP + j];
      __syncthreads();
      for (int e = tid; e < rows * 192; e += NTH) {
        const int j = e / 192, rem = e - j * 192, n = rem >> 6, l = rem & 63;
        floatx4 v = red[((n + 0) * RPP + j) * 64 + l];
#pragma unroll
        for (int q = 1; q < 4; ++q) {
          const floatx4 p = red[((n + 3 * q) * RPP + j) * 64 + l];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += p[r];
        }
        const int ch = n * 16 + (l >> 4) * 4, px = ox0 + (l & 15), oy = oy0 + h * RPP + j;
        const floatx4 bv = *reinterpret_cast<const floatx4*>(smem + Cfg::BIAS_OFF + ch * 4);
        half4_t y4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float y = v[r] + bv[r];
          y = fmaxf(y, 0.3f * y);
          y4[r] = (half_t)y;
        }
        if (oy < Hout && px < Wout) *reinterpret_cast<half4_t*>(out + ((size_t)oy * Wout + px) * 48 + ch) = y4;
      }
    }
    base += TR;
    if (base >= RING) base -= RING;
    ARTP_KW_MARK(3);
  }
#ifdef ARTP_STAGE_TIMING
  if (lane == 0)
    for (int k = 0; k < 4; ++k) g_kwalk_cycles[(((int)blockIdx.x & 255) * 12 + wave) * 4 + k] = kw_c[k];
#endif
}

// ---- round 6: the 15 x 15 layer on v_mfma_f32_32x32x16_f16, two output rows per accumulator set ---------------------
// What the matrix cores sustain (tests/cpp/mfma_clock_probe.hip, profiles/r06_mfma_clock_probe.txt): 32x32x16 issues back
// to back at 32.0 cycles from ONE wavefront per SIMD (0.84 of the nominal 2.5 PFLOP/s, 0.89 with two) where 16x16x32 needs
// two wavefronts per SIMD for 17 cycles per 16 (0.74-0.78) and a lone wavefront gets 26 -- and it reads half the operand
// registers per FLOP.  48 output channels are not a multiple of 32, so the M dimension is the channels of TWO output rows:
// for the output row pair (2j, 2j + 1) and the patch row rho = 2j + t, kernel row t feeds row 2j and kernel row t - 1 feeds
// row 2j + 1 FROM THE SAME ACTIVATIONS.  Stacked, that is a 96-row weight matrix per step t = 0 .. 15
//     rows  0 .. 47 = W[kh = t][channel 0 .. 47]      (-> output row 2j)
//     rows 48 .. 95 = W[kh = t - 1][channel 0 .. 47]  (-> output row 2j + 1)
// = three 32-row blocks that share one activation fragment (32 pixels x 16 k).  W[-1] = W[15] = 0: step 0 skips block 2,
// step 15 block 0, and half of block 1 is zero there -- 46 MFMAs per pair and k-half where 45 are needed.
// Otherwise the K-split kernel above: the four wavefronts split the k-steps, the weight fragments go global -> registers
// (ring, PD steps ahead), the activation rows come from the LDS patch through a register ring (ONE new row per step), the
// partial tiles meet in LDS a row pair at a time.  One 4-wavefront workgroup per CU (a wavefront per SIMD, ~300 registers).
template <int TR>
struct Conv15P32Cfg {
  static constexpr int KH = 15, KW = 15, CIN = 48, COUT = 48;
  static constexpr int KROW = KW * CIN;                    // 720 halfs per kernel row
  static constexpr int KSTEPS = (KROW + 31) / 32;          // 23 k-steps of 32 (two MFMA k-halves each)
  static constexpr int TP = 32, NP = TR / 2, NSTEP = KH + 1;
  static constexpr int PR = TR + KH - 1;
  static constexpr int XPAD = (KSTEPS * 32 - KROW + CIN - 1) / CIN;
  static constexpr int PPX = TP + KW - 1 + XPAD;
  static constexpr int PIX_B = CIN * 2;
  static constexpr int ROW_B = PPX * PIX_B;
  static constexpr int A_BYTES = PR * ROW_B;
  static constexpr int RED_BYTES = 4 * 12 * 64 * 16;       // one row pair: 4 wavefronts x (3 blocks x 4 quads) x 64 lanes x float4
  static constexpr int STAGE_BYTES = TR * TP * COUT * 2;
  static constexpr int LDS_BYTES = A_BYTES > RED_BYTES + STAGE_BYTES ? A_BYTES : RED_BYTES + STAGE_BYTES;
  // packed weights: [kernel row -1 .. 15 (both ends zero)][k-step][k-half][channel 0 .. 47][k-quad 0 .. 1] half8
  static constexpr int SH_FR = COUT * 2, KS_FR = 2 * SH_FR, KH_FR = KSTEPS * KS_FR;
  static constexpr size_t W_FRAGS = (size_t)(KH + 2) * KH_FR;
  static_assert(TR % 2 == 0 && TR >= 4, "row pairs");
  static_assert(ROW_B % 16 == 0 && LDS_BYTES <= 160 * 1024, "16-byte chunks; one workgroup per CU");
};

typedef float floatx16 __attribute__((ext_vector_type(16)));

// wp: Conv15P32Cfg's packed weights.  The three 32-row blocks of a step are not stored: a lane picks its stacked row's
// (kernel row, channel) itself -- block 0: (t, m); block 1: (t, 32 + m) for m < 16, (t - 1, m - 16) above; block 2:
// (t - 1, 16 + m) -- so the weights keep their size (1.2 MB in the L2s, not 2.3) and the kernel row a step read as "t" is
// the one the next step reads as "t - 1", one step later in the same wavefront: an L1 hit.
template <int TR, bool XCD = true, int RD = 4, int PD = 3>
__global__ void __launch_bounds__(256, 1)
conv15_pair32_kernel(const half_t* __restrict__ in, int Hin, int Win, const half8* __restrict__ wp,
                     const float* __restrict__ bias, half_t* __restrict__ out) {
  using Cfg = Conv15P32Cfg<TR>;
  constexpr int KSTEPS = Cfg::KSTEPS, NP = Cfg::NP, NSTEP = Cfg::NSTEP, NTH = 256, NK = 4;
  static_assert(NSTEP % RD == 0 && PD < RD, "the weight ring runs on across k-steps: slot = step % RD");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int Hout = Hin - Cfg::KH + 1, Wout = Win - Cfg::KW + 1;
  const int tiles_x = (Wout + Cfg::TP - 1) / Cfg::TP;
  const int tile = XCD ? xcd_contiguous((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
  const int bx = tile % tiles_x, by = tile / tiles_x;
  const int oy0 = by * TR, ox0 = bx * Cfg::TP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int px = lane & 31, kq = lane >> 5;
  const char* a_lane = smem + px * Cfg::PIX_B + kq * 16;
  // which k-steps a wavefront takes, and in which order, rotates with the workgroup index (conv_ksplit_kernel)
  const int ks_first = (wave + (int)(blockIdx.x & 3u)) & 3;
  const int nj = (KSTEPS - ks_first + NK - 1) / NK;
  const int j0 = (int)((blockIdx.x >> 2) % (unsigned)nj);
  auto ks_of = [&](int jj) {
    const int j = jj + j0 < nj ? jj + j0 : jj + j0 - nj;
    return ks_first + NK * j;
  };
  constexpr int SH_FR = Cfg::SH_FR, KS_FR = Cfg::KS_FR, KH_FR = Cfg::KH_FR;
  // this lane's row in the three blocks, as an offset (in half8 units) from the (kernel row t, k-step, k-half 0) slab
  const int off_b0 = px * 2 + kq;
  const int off_b1 = px < 16 ? (32 + px) * 2 + kq : (px - 16) * 2 + kq - KH_FR;
  const int off_b2 = (16 + px) * 2 + kq - KH_FR;
  half8 w[RD][6];   // [slot][block * 2 + k-half]
  // input patch -> LDS (conv_ksplit_kernel's copy: all loads of a thread in flight before its first LDS store)
  {
    constexpr int CPR = Cfg::ROW_B / 16;
    constexpr int NIT = (Cfg::PR * CPR + NTH - 1) / NTH;
    const long row_bytes = (long)Win * Cfg::PIX_B;
    half8 v[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = tid + it * NTH;
      const int r = c / CPR, cc = c - r * CPR;
      const long off = (long)ox0 * Cfg::PIX_B + (long)cc * 16;
      const bool ok = c < Cfg::PR * CPR && oy0 + r < Hin && off + 16 <= row_bytes;
      // unconditional load from a clamped address + select (hipcc serialises conditional loads: a trip to memory each)
      const half8 ld = *reinterpret_cast<const half8*>(reinterpret_cast<const char*>(in) + (ok ? (long)(oy0 + r) * row_bytes + off : 0));
#pragma unroll
      for (int j = 0; j < 8; ++j) v[it][j] = ok ? ld[j] : (half_t)0;
    }
    {  // the ring's first PD steps, requested behind the patch
      const half8* w0 = wp + (size_t)ks_of(0) * KS_FR + KH_FR;   // kernel row 0 = slab 1
#pragma unroll
      for (int d = 0; d < PD; ++d)
#pragma unroll
        for (int sh = 0; sh < 2; ++sh) {
          w[d][0 + sh] = w0[(size_t)d * KH_FR + sh * SH_FR + off_b0];
          w[d][2 + sh] = w0[(size_t)d * KH_FR + sh * SH_FR + off_b1];
          if (d > 0) w[d][4 + sh] = w0[(size_t)d * KH_FR + sh * SH_FR + off_b2];   // step 0 has no block 2
        }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = tid + it * NTH;
      const int r = c / CPR, cc = c - r * CPR;
      if (c < Cfg::PR * CPR) *reinterpret_cast<half8*>(smem + r * Cfg::ROW_B + cc * 16) = v[it];
    }
  }
  __syncthreads();

  floatx16 acc[NP][3];
#pragma unroll
  for (int j = 0; j < NP; ++j)
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[j][b][q] = 0.f;

  for (int jj = 0; jj < nj; ++jj) {
    const int ks = ks_of(jj);
    const char* a_ks = a_lane + ks * 64;   // + row * ROW_B + k-half * 32
    const half8* w_ks = wp + (size_t)ks * KS_FR + KH_FR;
    const half8* w_nx = wp + (size_t)ks_of(jj + 1 < nj ? jj + 1 : jj) * KS_FR + KH_FR;  // the next k-step's (or a harmless re-read)
    half8 a[TR][2];   // ring: slot rho % TR holds patch row rho (its two k-halves); row rho is used at steps t = rho - 2j
#pragma unroll
    for (int r = 0; r < TR - 1; ++r)
#pragma unroll
      for (int sh = 0; sh < 2; ++sh) a[r][sh] = *reinterpret_cast<const half8*>(a_ks + r * Cfg::ROW_B + sh * 32);
#pragma unroll
    for (int t = 0; t < NSTEP; ++t) {
      // the row step t + 1 needs new (rho = t + TR - 1; its slot's last occupant, row t - 1, was last used at step t - 1)
      if (t + TR - 1 < Cfg::PR) {
#pragma unroll
        for (int sh = 0; sh < 2; ++sh)
          a[(t + TR - 1) % TR][sh] = *reinterpret_cast<const half8*>(a_ks + (t + TR - 1) * Cfg::ROW_B + sh * 32);
      }
      {  // weights of step t + PD
        const int tn = t + PD;
#ifdef ARTP_P32_SAMEW   // bound experiment (wrong sums): every step reads the same 6 KB -- no weight stream from L2
        const half8* src = wp + KH_FR + (size_t)(tn & 1) * KH_FR;
#else
        const half8* src = tn < NSTEP ? w_ks + (size_t)tn * KH_FR : w_nx + (size_t)(tn - NSTEP) * KH_FR;
#endif
        const int tt = tn < NSTEP ? tn : tn - NSTEP;
#pragma unroll
        for (int sh = 0; sh < 2; ++sh) {
          if (tt < NSTEP - 1) w[(t + PD) % RD][0 + sh] = src[sh * SH_FR + off_b0];
          w[(t + PD) % RD][2 + sh] = src[sh * SH_FR + off_b1];
          if (tt > 0) w[(t + PD) % RD][4 + sh] = src[sh * SH_FR + off_b2];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NP; ++j) {
#pragma unroll
        for (int sh = 0; sh < 2; ++sh) {
          const half8 x = a[(2 * j + t) % TR][sh];
          if (t < NSTEP - 1) acc[j][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[t % RD][0 + sh], x, acc[j][0], 0, 0, 0);
          acc[j][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[t % RD][2 + sh], x, acc[j][1], 0, 0, 0);
          if (t > 0) acc[j][2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[t % RD][4 + sh], x, acc[j][2], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // The product is TRANSPOSED (weights = first operand): column (lane & 31) = pixel, register v of a block = stacked row
  // 8 (v / 4) + 4 (lane >> 5) + v % 4 -- four consecutive channels of one output row per register quad.  The four partial
  // tiles meet in LDS one row pair at a time (the patch is dead); wavefront w finishes the (block, quad) items w, w + 4, w + 8.
  typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
  floatx4* red = reinterpret_cast<floatx4*>(smem);
  char* stage = smem + Cfg::RED_BYTES;
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    __syncthreads();
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        red[((wave * 3 + b) * 4 + q) * 64 + lane] = floatx4{acc[j][b][4 * q], acc[j][b][4 * q + 1], acc[j][b][4 * q + 2], acc[j][b][4 * q + 3]};
    __syncthreads();
#pragma unroll
    for (int i3 = 0; i3 < 3; ++i3) {
      const int item = wave + 4 * i3, b = item >> 2, q = item & 3;
      floatx4 v = red[((0 * 3 + b) * 4 + q) * 64 + lane];
#pragma unroll
      for (int ww = 1; ww < 4; ++ww) {
        const floatx4 pv = red[((ww * 3 + b) * 4 + q) * 64 + lane];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += pv[r];
      }
      const int R = 32 * b + 8 * q + 4 * kq;          // stacked row of v[0]
      const int orow = 2 * j + (R >= 48 ? 1 : 0), ch = R >= 48 ? R - 48 : R;
      const floatx4 bv = *reinterpret_cast<const floatx4*>(bias + ch);
      half4_t y4;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float y = v[r] + bv[r];
        y = fmaxf(y, 0.3f * y);
        y4[r] = (half_t)y;
      }
      *reinterpret_cast<half4_t*>(stage + ((orow * Cfg::TP + px) * Cfg::COUT + ch) * 2) = y4;
    }
  }
  __syncthreads();
  {
    constexpr int CPR = Cfg::TP * Cfg::COUT * 2 / 16;  // 16-byte chunks per tile row
    for (int c = tid; c < TR * CPR; c += NTH) {
      const int m = c / CPR, cc = c - m * CPR;
      const int opx = (cc * 16) / (Cfg::COUT * 2);
      if (oy0 + m < Hout && ox0 + opx < Wout)
        *reinterpret_cast<half8*>(reinterpret_cast<char*>(out) + ((size_t)(oy0 + m) * Wout + ox0) * Cfg::COUT * 2 + cc * 16) =
            *reinterpret_cast<const half8*>(stage + m * CPR * 16 + cc * 16);
    }
  }
}

}  // namespace artp
