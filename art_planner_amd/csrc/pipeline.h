// pipeline.h -- the throughput path of StateValidityChecker::isValid on gfx950.
//
// The reference decides ~3/4 of all box checks from three window statistics alone (max height, min
// finite height, all-finite): exits (b)(c)(d)(e) of dCollideHeightfieldZone
// (ode/ode/src/heightfield.cpp:1027-1064,1139-1160) -- but only after scanning the whole window.
// max / min are idempotent, so exact 2-D range-max / range-min-of-finite tables over power-of-two
// blocks (built once per map upload, resident in HBM/L2) give bit-identical statistics from a handful
// of loads; "holds a non-finite / a NaN sample" comes from per-block flag bytes.  That turns the common
// case into lane-parallel work and leaves the cooperative window work to the boxes that really need it
// (DESIGN.md 4.1 has the exactness argument of every shortcut):
//
//   classify_states_kernel      1 lane / (state, box): poses, frame change, AABB, window, table statistics,
//                               exits (b)-(e), 2 x 2 vertex probe of (f) for feet; undecided -> queue 1
//   feet_stream_kernel          16 lanes / foot box  : (f) streamed off the map, list-free corner stage
//   resolve_boxes_kernel<.,64,0>  1 wave / torso box : the same for the ~900-sample torso windows
//   resolve_boxes_kernel<.,16,2> 16 lanes / box      : candidates with possible partners: LDS tile,
//                               kept-triangle list, partner search (queue 5)
//   plane_stage_kernel          1 wave / box         : exact greedy plane grouping (queue 2)
//   feet_lane_kernel + resolve_boxes_kernel<.,16,1>  : boxes the tables cannot answer (a NaN in the window,
//                               windows thinner than 4 samples): ordered scan with ODE's running-dMAX quirk
//
// A state's label is the AND over its boxes of "torso does not touch" / "foot touches", so boxes can
// be decided in any order and in different kernels; a failing box stores 0 into the state's label.
// Evaluating boxes the reference would have short-circuited cannot change the label.
#pragma once

#include <hip/hip_fp16.h>

#include "kernels.h"

namespace artp {

#define ARTP_TABLE_LEVELS 4  // block sizes 4, 8, 16, 32 samples

struct TablesDev {
  // {max, min-of-finite} interleaved so one 8-byte gather answers both (the lookups are random
  // accesses into tables larger than one XCD's L2: cache lines touched, not bytes, are the cost).
  // The levels lie back to back, `stride` entries apart: level l (block 4 << l) of entry i is mm[l * stride + i].
  // One base pointer + arithmetic instead of an array of pointers: a per-lane level index into a pointer array in
  // the kernel arguments costs a dependent memory round trip (pointer, then gather) in front of every lookup.
  const float2* mm;         // .x = max over [x, x+B) x [z, z+B), NaN samples count as -inf
                            // .y = min over the FINITE samples of the block, +inf if none
  const unsigned char* fl;  // bit 0: the block holds a non-finite sample, bit 1: a NaN
  unsigned stride;          // entries per level (= nW * nD)
  int has_nan;              // the layer holds a NaN somewhere
  int has_nonfinite;        // the layer holds a NaN or an infinity somewhere (else fl is all zero)
  int valid;
  // Stride tables: blocks of 8 / 16 / 32 samples anchored every 2 / 4 / 8 samples only (level l: block 8 << l,
  // anchors at multiples of 2 << l), back to back in `st`, one packed pair of HALF floats per entry.  210 KB per
  // 400 x 400 layer instead of 5 MB of exact tables: they stay in every XCD's L2 (the feet's 16-sample table, 40 KB,
  // largely in the L1s), and they are CONSERVATIVE bounds (see stride_entry) that decide most boxes before an exact
  // table is touched.
  const unsigned* st;
  unsigned st_off1, st_off2;  // first entry of levels 1 and 2 (level 0 starts at 0)
};

#define ARTP_STRIDE_LEVELS 3

// ---- table construction (map upload) ---------------------------------------------------------------
__global__ void __launch_bounds__(256)
table_level0_kernel(const float* __restrict__ data, int n, float2* __restrict__ mm, unsigned char* __restrict__ fl) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const float h = data[i];
    mm[i] = make_float2(is_nan(h) ? -INFINITY : h, is_finite(h) ? h : INFINITY);
    fl[i] = (unsigned char)((is_finite(h) ? 0 : 1) | (is_nan(h) ? 2 : 0));
  }
}

// out = combine of the four half-size blocks at offsets (0,0), (h,0), (0,h), (h,h); indices clamp at
// the border (blocks hanging over the edge are never queried).
__global__ void __launch_bounds__(256)
table_level_up_kernel(const float2* __restrict__ in, const unsigned char* __restrict__ fin, int nW, int nD, int half,
                      float2* __restrict__ out, unsigned char* __restrict__ fout) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nW * nD) return;
  const int x = i % nW, z = i / nW;
  const int x1 = min(x + half, nW - 1), z1 = min(z + half, nD - 1);
  const float2 a = in[x + z * nW], b = in[x1 + z * nW], c = in[x + z1 * nW], d = in[x1 + z1 * nW];
  const float ab = (b.x > a.x) ? b.x : a.x, cd = (d.x > c.x) ? d.x : c.x;
  const float ef = (b.y < a.y) ? b.y : a.y, gh = (d.y < c.y) ? d.y : c.y;
  out[i] = make_float2((cd > ab) ? cd : ab, (gh < ef) ? gh : ef);
  fout[i] = fin[x + z * nW] | fin[x1 + z * nW] | fin[x + z1 * nW] | fin[x1 + z1 * nW];
}

// Stride-table entry: {max', min'} of a block with exact statistics {mx, mn} and flags fl (bit 0 non-finite sample,
// bit 1 NaN), as two half floats (max' in the low 16 bits).  The tiers that read it only need max' >= max and
// min' <= min-of-finite, so the values are rounded OUTWARD to half precision (a millimetre at the heights of a map:
// far below what decides an exit) and the flags ride in them: a NaN block becomes {+inf, -inf} (no exit can fire on
// it), and "holds a non-finite sample" is the lowest mantissa bit of max', moved UP to the next half of that parity
// (+inf cannot carry the bit and does not need it: exit (d) never fires on max' = +inf).
__device__ __forceinline__ unsigned stride_entry(float mx, float mn, unsigned fl) {
  if (fl & 2u) return 0xFC007C00u;
  unsigned u = (unsigned)__half_as_ushort(__float2half_ru(mx));
  const unsigned bit = fl & 1u;
  if ((u & 1u) != bit && u != 0x7C00u) {
    if ((u & 0x7FFFu) == 0u) u = 1u;        // +-0 -> smallest positive subnormal
    else if (u & 0x8000u) u -= 1u;          // negative: toward zero
    else u += 1u;
  }
  return u | ((unsigned)__half_as_ushort(__float2half_rd(mn)) << 16);
}
__device__ __forceinline__ float stride_max(unsigned e) { return __half2float(__ushort_as_half((unsigned short)(e & 0xFFFFu))); }
__device__ __forceinline__ float stride_min(unsigned e) { return __half2float(__ushort_as_half((unsigned short)(e >> 16))); }

// One thread per stride-table entry, all levels in one launch.  mm / fl = the exact tables of blocks 4, 8, 16, 32
// (TablesDev::mm / fl); level l of the stride tables copies the exact level l + 1 at its anchors.
__global__ void __launch_bounds__(256)
stride_tables_kernel(const float2* __restrict__ mm, const unsigned char* __restrict__ fl, unsigned stride, int nW,
                     int nD, unsigned off1, unsigned off2, unsigned total, unsigned* __restrict__ out) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int l = i >= off2 ? 2 : (i >= off1 ? 1 : 0);
  const unsigned j = i - (l == 2 ? off2 : (l == 1 ? off1 : 0u));
  const int sh = l + 1, nxs = (nW + (1 << sh) - 1) >> sh;
  const int x = (int)(j % (unsigned)nxs) << sh, z = (int)(j / (unsigned)nxs) << sh;
  const size_t at = (size_t)(l + 1) * stride + (size_t)x + (size_t)z * nW;
  const float2 e = mm[at];
  out[i] = stride_entry(e.x, e.y, fl[at]);
}

// Partner table (FieldDev::partner_flags): both triangles of a cell against every triangle of the
// (2R+1)^2 cell neighbourhood, with the very arithmetic of the plane stage (triangle_plane on absolute
// sample coordinates, the raw-cross pre-filter, the four epsilon compares).
// Pass 1 (tri_raw_rects_kernel): per cell {raw0, raw2} of the ABC and the DBC triangle (+inf for a triangle
// with a non-finite vertex: it can never be kept, and +inf fails the pre-filter).
// Pass 2 (partner_count_kernel): one lane per (own cell, neighbour row); the inner loop is one 16-byte load and four
// pre-filter compares per neighbour cell, the exact planes are only formed on a pre-filter hit.  Partners are COUNTED
// per own triangle; the byte table the box stages read is derived from the counts (see below).
__device__ __forceinline__ void cell_triangle(const FieldDev& f, int cx, int cz, bool up, bool& finite,
                                              float pl[4], float raw[3]) {
  const int i = cx + cz * f.nW;
  const float hA = f.data[i], hB = f.data[i + 1], hC = f.data[i + f.nW], hD = f.data[i + f.nW + 1];
  const float xA = (float)cx * f.sample_w, xB = (float)(cx + 1) * f.sample_w;
  const float zA = (float)cz * f.sample_d, zC = (float)(cz + 1) * f.sample_d;
  finite = is_finite(hB) && is_finite(hC) && is_finite(up ? hA : hD);
  if (up)
    triangle_plane(xA, hA, zA, xB, hB, zA, xA, hC, zC, true, pl, raw);
  else
    triangle_plane(xB, hD, zC, xB, hB, zA, xA, hC, zC, false, pl, raw);
}

// The table is kept as COUNTS (cnt32[cell]: low half = partner triangles of the cell's ABC, high half = of its DBC;
// at most 2 (2R+1)^2 < 2^16) and the byte flags the box stages read are derived from them.  Counts are what makes a
// rectangle update local: a pair (own cell c, neighbour cell e) can only change when one of its ends changed, so
//   c inside a changed rectangle:   recount over the full neighbourhood (RESTRICT = false)
//   c outside, within R of one:     count -= its partners among the rectangle's OLD triangles (before the samples are
//                                   overwritten), count += its partners among the NEW ones (RESTRICT = true, sign -+1)
// -- (53 + 2 R)^2 cells x 53^2 neighbours twice + 53^2 cells x (2R+1)^2 instead of (53 + 2 R + 2)^2 x (2R+1)^2 pairs
// for a 52 x 52 sample patch and the torso's R = 41: 2.8 x fewer of the ~30-instruction pair tests this kernel is
// bound by.  A pair is always evaluated from the own cell's side with the own cell's tolerance, exactly as the full
// build evaluates it, so the counts of an updated table equal those of a fresh build.
struct PartnerRects {  // changed-cell rectangles (inclusive cell ranges), pairwise disjoint
  int n;
  int x0[8], z0[8], x1[8], z1[8];
};

// One launch for all rectangles (blockIdx.z = rectangle k).  RESTRICT: own cells = rectangle k widened by R, neighbours =
// rectangle k, grid.y = its rows.  Otherwise: own cells = rectangle k, neighbours = everything within R, grid.y = 2 R + 1
// (the whole layer is one rectangle).
template <bool RESTRICT>
__global__ void __launch_bounds__(256)
partner_count_kernel(FieldDev f, int R, const float4* __restrict__ raw4, unsigned* __restrict__ cnt32, PartnerRects pr,
                     int sign) {
  const int k = blockIdx.z;
  const int cx0 = RESTRICT ? max(pr.x0[k] - R, 0) : pr.x0[k], cz0 = RESTRICT ? max(pr.z0[k] - R, 0) : pr.z0[k];
  const int cx1 = RESTRICT ? min(pr.x1[k] + R, f.nW - 2) : pr.x1[k], cz1 = RESTRICT ? min(pr.z1[k] + R, f.nD - 2) : pr.z1[k];
  const int ncx = cx1 - cx0 + 1, ncz = cz1 - cz0 + 1;
  const int li = blockIdx.x * blockDim.x + threadIdx.x;
  if (li >= ncx * ncz) return;
  const int cx = cx0 + li % ncx, cz = cz0 + li / ncx;
  if (cx >= f.nW - 1 || cz >= f.nD - 1) return;
  int z, xa = cx - R, xb = cx + R;
  if (RESTRICT) {
    // own cells inside ANY changed rectangle are recounted in full elsewhere
    for (int j = 0; j < pr.n; ++j)
      if (cx >= pr.x0[j] && cx <= pr.x1[j] && cz >= pr.z0[j] && cz <= pr.z1[j]) return;
    z = pr.z0[k] + (int)blockIdx.y;
    if (z > pr.z1[k] || z < cz - R || z > cz + R) return;
    xa = max(xa, pr.x0[k]);
    xb = min(xb, pr.x1[k]);
  } else {
    if ((int)blockIdx.y > 2 * R) return;
    z = cz + (int)blockIdx.y - R;
  }
  if (z < 0 || z > f.nD - 2) return;
  const int i = cx + cz * f.nW;
  float pl[2][4], raw[2][3], tol[2];
  bool own[2];
  cell_triangle(f, cx, cz, true, own[0], pl[0], raw[0]);
  cell_triangle(f, cx, cz, false, own[1], pl[1], raw[1]);
  if (!(own[0] || own[1])) return;
#pragma unroll
  for (int o = 0; o < 2; ++o) {
    const float len = sqrtf(raw[o][0] * raw[o][0] + raw[o][1] * raw[o][1] + raw[o][2] * raw[o][2]);
    tol[o] = !own[o] ? -1.0f : ((fabsf(pl[o][1]) >= 0.05f) ? f.partner_tol * len : INFINITY);
  }
  unsigned n0 = 0, n1 = 0;
  const int x0 = max(xa, 0), x1 = min(xb, f.nW - 2);
  for (int x = x0; x <= x1; ++x) {
    const int e = x + z * f.nW;
    const float4 r = raw4[e];
    // pre-filter of own triangle o against the neighbour cell's ABC (r.x, r.y) and DBC (r.z, r.w);
    // tol = -1 (own triangle not finite) never passes
    const bool c00 = !(fabsf(r.x - raw[0][0]) > tol[0]) && !(fabsf(r.y - raw[0][2]) > tol[0]);
    const bool c01 = !(fabsf(r.z - raw[0][0]) > tol[0]) && !(fabsf(r.w - raw[0][2]) > tol[0]);
    const bool c10 = !(fabsf(r.x - raw[1][0]) > tol[1]) && !(fabsf(r.y - raw[1][2]) > tol[1]);
    const bool c11 = !(fabsf(r.z - raw[1][0]) > tol[1]) && !(fabsf(r.w - raw[1][2]) > tol[1]);
    if (!(c00 || c01 || c10 || c11)) continue;
    const bool same = (e == i);
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      if (!(p == 0 ? (c00 || c10) : (c01 || c11))) continue;
      float p2[4], r2[3];
      bool fin;
      cell_triangle(f, x, z, p == 0, fin, p2, r2);
      if (!fin) continue;
      if (own[0] && !(same && p == 0) && planes_eps_equal(pl[0], p2)) ++n0;
      if (own[1] && !(same && p == 1) && planes_eps_equal(pl[1], p2)) ++n1;
    }
  }
  const unsigned delta = n0 | (n1 << 16);
  if (delta) {
    if (sign > 0) atomicAdd(&cnt32[i], delta); else atomicSub(&cnt32[i], delta);
  }
}

// raw cross products of the rectangles' cells (the changed cells of an update), their counts to zero for the recount
__global__ void __launch_bounds__(256)
tri_raw_rects_kernel(FieldDev f, PartnerRects pr, float4* __restrict__ raw4, unsigned* __restrict__ cnt32) {
  const int k = blockIdx.y;
  const int ncx = pr.x1[k] - pr.x0[k] + 1, ncz = pr.z1[k] - pr.z0[k] + 1;
  const int li = blockIdx.x * blockDim.x + threadIdx.x;
  if (li >= ncx * ncz) return;
  const int cx = pr.x0[k] + li % ncx, cz = pr.z0[k] + li / ncx;
  float4 o = make_float4(INFINITY, INFINITY, INFINITY, INFINITY);
  if (cx < f.nW - 1 && cz < f.nD - 1) {
    float pl[4], raw[3];
    bool fin;
    cell_triangle(f, cx, cz, true, fin, pl, raw);
    if (fin) { o.x = raw[0]; o.y = raw[2]; }
    cell_triangle(f, cx, cz, false, fin, pl, raw);
    if (fin) { o.z = raw[0]; o.w = raw[2]; }
  }
  raw4[cx + cz * f.nW] = o;
  cnt32[cx + (size_t)cz * f.nW] = 0u;
}

// byte flags from the counts (bit 0: the ABC triangle has a partner, bit 1: the DBC triangle) for the rectangles
// widened by `margin` cells
__global__ void __launch_bounds__(256)
partner_flags_from_counts_kernel(int nW, int nD, PartnerRects pr, int margin, const unsigned* __restrict__ cnt32,
                                 unsigned char* __restrict__ flags) {
  const int k = blockIdx.y;
  const int cx0 = max(pr.x0[k] - margin, 0), cz0 = max(pr.z0[k] - margin, 0);
  const int cx1 = min(pr.x1[k] + margin, nW - 1), cz1 = min(pr.z1[k] + margin, nD - 1);
  const int ncx = cx1 - cx0 + 1, ncz = cz1 - cz0 + 1;
  const int li = blockIdx.x * blockDim.x + threadIdx.x;
  if (li >= ncx * ncz) return;
  const size_t i = (cx0 + li % ncx) + (size_t)(cz0 + li / ncx) * nW;
  const unsigned c = cnt32[i];
  flags[i] = (unsigned char)(((c & 0xffffu) ? 1u : 0u) | ((c >> 16) ? 2u : 0u));
}

// Conservative exits from the stride tables.  A set of blocks whose union CONTAINS the window bounds the window's
// statistics from outside: max_c >= maxY, min_c <= minY, and "blocks all finite" implies "window all finite".
// That decides most boxes exactly:
//   (b) above : minO2 - max_c > -eps  =>  minO2 - maxY > -eps                       -> no contact
//   (c) under : min_c - maxO2 > -eps  =>  minY - maxO2 > -eps (whether (b) fired first or not: both say 0)
//   (d) spans : blocks finite, min_c - minO2 >= eps and maxO2 - max_c >= eps  =>  (b) and (c) cannot fire
//               (maxY >= min_c >= minO2 + eps, minY <= max_c <= maxO2 - eps) and (d)'s own tests hold
// Returns 0 / 1 = decided result, -1 = undecided.
__device__ __forceinline__ int conservative_exits(const BoxHF& b, float max_c, float min_c, bool nonfinite) {
  const float minO2 = b.aabb[2], maxO2 = b.aabb[3];
  if (minO2 - max_c > -ARTP_EPS) return 0;
  if (min_c - maxO2 > -ARTP_EPS) return 0;
  if (!nonfinite && min_c - minO2 >= ARTP_EPS && maxO2 - max_c >= ARTP_EPS) return 1;
  return -1;
}

__device__ __forceinline__ unsigned stride_level_offset(const TablesDev& t, int l) {
  return l == 2 ? t.st_off2 : (l == 1 ? t.st_off1 : 0u);  // selects, not an indexed load from the kernel arguments
}

// Tier 1: ONE block that contains the window.  A block of B samples anchored at the multiple of s = B / 4 below
// the window's corner reaches at least B - s + 1 samples past it: windows up to 7 / 13 / 25 samples use the
// 8 / 16 / 32 blocks.
// cover_finite (both tiers): set when the blocks looked at prove that the WINDOW holds no non-finite sample (no block
// carries the flag bit and none has max' = +inf, which is what a NaN block or a block with a +inf sample stores): the
// tail then needs neither the flag gathers of the exact statistics nor the per-vertex flags of the probe.
__device__ __forceinline__ int coarse_block_exit(const FieldDev& f, const TablesDev& t, const BoxHF& b, bool& cover_finite) {
  const int wX = b.maxX - b.minX + 1, wZ = b.maxZ - b.minZ + 1;
  const int wmax = wX > wZ ? wX : wZ;
  if (wmax > 25) return -1;
  const int l = wmax <= 7 ? 0 : (wmax <= 13 ? 1 : 2), sh = l + 1;
  const int nxs = (f.nW + (1 << sh) - 1) >> sh;
  const unsigned e = gather32(t.st, stride_level_offset(t, l) + (unsigned)((b.minX >> sh) + (b.minZ >> sh) * nxs));
  const float mx = stride_max(e);
  cover_finite = !(e & 1u) && mx < INFINITY;
  return conservative_exits(b, mx, stride_min(e), e & 1u);
}

// Tier 2: a tight cover.  Blocks of B <= min(wX, wZ) (8 at least) per axis: one at the anchor below the window's low
// end, one at the anchor above (high end - B + 1), a third between them when the window is longer than 2 B - s: the
// union reaches at most s - 1 = B / 4 - 1 samples past the window on each side, so it decides nearly everything the
// exact statistics would -- out of tables that stay in L2.  Up to 3 x 3 predicated gathers in one round trip.
// N = the most blocks per axis the caller is willing to gather (N x N predicated loads are issued whether a lane
// needs them or not: a foot window takes 2 x 2 of the 8-sample blocks, a torso window up to 3 x 3 of the 16s); a
// window that needs more stays undecided here.
template <int N>
__device__ __forceinline__ int tight_cover_exit(const FieldDev& f, const TablesDev& t, const BoxHF& b, bool& cover_finite) {
  const int wX = b.maxX - b.minX + 1, wZ = b.maxZ - b.minZ + 1;
  const int m = wX < wZ ? wX : wZ, wl = wX > wZ ? wX : wZ;
  // block size by the short side, but large enough that N blocks span the long side (N B - s samples at least): a
  // torso window of 33 x 15 samples takes three 16-sample blocks by one, not five 8-sample blocks it may not gather
  int l = m >= 32 ? 2 : (m >= 16 ? 1 : 0);
  if (l < 2 && wl > N * (8 << l) - (2 << l)) ++l;
  if (l < 2 && wl > N * (8 << l) - (2 << l)) ++l;
  const int sh = l + 1, B = 8 << l, s = 1 << sh;
  const int nxs = (f.nW + s - 1) >> sh;
  int px[3], pz[3], nx, nz;
  {
    const int a0 = (b.minX >> sh) << sh;
    const int aR = a0 + B > b.maxX ? a0 : ((b.maxX - B + s) >> sh) << sh;  // ceil to the stride of (maxX - B + 1)
    nx = aR == a0 ? 1 : (aR <= a0 + B ? 2 : (aR <= a0 + 2 * B ? 3 : 4));
    px[0] = a0; px[1] = nx == 3 ? a0 + B : aR; px[2] = aR;
  }
  {
    const int a0 = (b.minZ >> sh) << sh;
    const int aR = a0 + B > b.maxZ ? a0 : ((b.maxZ - B + s) >> sh) << sh;
    nz = aR == a0 ? 1 : (aR <= a0 + B ? 2 : (aR <= a0 + 2 * B ? 3 : 4));
    pz[0] = a0; pz[1] = nz == 3 ? a0 + B : aR; pz[2] = aR;
  }
  if (nx > N || nz > N) return -1;
  const unsigned lo = stride_level_offset(t, l);
  unsigned v[N * N];
#pragma unroll
  for (int u = 0; u < N * N; ++u) {
    const int i = u % N, j = u / N;
    v[u] = 0x7C00FC00u;  // masked-off slot: {max' = -inf (lowest bit 0), min' = +inf}
    // block i of an axis with n <= N blocks: the first, [the middle one,] the last
    const int xi = (i == 0) ? px[0] : ((i == nx - 1) ? px[2] : px[1]);
    const int zj = (j == 0) ? pz[0] : ((j == nz - 1) ? pz[2] : pz[1]);
    if (i < nx && j < nz) v[u] = gather32(t.st, lo + (unsigned)((xi >> sh) + (zj >> sh) * nxs));
  }
  float vmax = -INFINITY, vmin = INFINITY;
  unsigned nf = 0u;
#pragma unroll
  for (int u = 0; u < N * N; ++u) {
    nf |= v[u];
    const float mx = stride_max(v[u]), mn = stride_min(v[u]);
    vmax = (mx > vmax) ? mx : vmax;
    vmin = (mn < vmin) ? mn : vmin;
  }
  cover_finite = cover_finite || (!(nf & 1u) && vmax < INFINITY);
  return conservative_exits(b, vmax, vmin, nf & 1u);
}

// Exact window statistics from the tables.  Returns false when the tables cannot answer (window
// thinner than the smallest block, or a NaN in the window -> the running-dMAX quirk needs the scan).
__device__ __forceinline__ bool table_window_stats(const FieldDev& f, const TablesDev& t, const BoxHF& b,
                                                   WindowStats& w, bool window_known_finite) {
  const int wX = b.maxX - b.minX + 1, wZ = b.maxZ - b.minZ + 1;
  const int m = wX < wZ ? wX : wZ;
  if (m < 4) return false;
  const int lvl = m >= 32 ? 3 : (m >= 16 ? 2 : (m >= 8 ? 1 : 0));
  const int B = 4 << lvl;
  const unsigned lo = (unsigned)lvl * t.stride;
  float vmax = -INFINITY, vmin = INFINITY;
  unsigned flags = 0;
  const int lastX = b.maxX - B + 1, lastZ = b.maxZ - B + 1;
  // blocks anchored at min + i * B, the last one clamped to last: ceil(w / B) per axis.  The short axis needs at
  // most 2 (B <= m < 2 B below the top level), the long one 2 - 3 for the boxes of a legged robot.  A 3 x 2
  // batch of gathers is issued at once (lanes that need fewer are masked off: no transaction), so the statistics
  // cost ONE memory round trip instead of up to nine dependent ones; larger windows loop.
  const int nx = (wX + B - 1) >> (lvl + 2), nz = (wZ + B - 1) >> (lvl + 2);
  for (int j0 = 0; j0 < nz; j0 += 2) {
    for (int i0 = 0; i0 < nx; i0 += 3) {
      float2 v[6];
      unsigned fb[6];
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        const int i = i0 + (u % 3), j = j0 + (u / 3);
        v[u] = make_float2(-INFINITY, INFINITY);
        fb[u] = 0u;
        if (i < nx && j < nz) {
          const int xx = b.minX + i * B, zz = b.minZ + j * B;
          const unsigned at = lo + (unsigned)((xx < lastX ? xx : lastX) + (zz < lastZ ? zz : lastZ) * f.nW);
          v[u] = gather32(t.mm, at);
          // a fully finite layer (uniform) and a window the stride tables proved finite skip the flag lookup
          if (t.has_nonfinite && !window_known_finite) fb[u] = gather32(t.fl, at);
        }
      }
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        vmax = (v[u].x > vmax) ? v[u].x : vmax;
        vmin = (v[u].y < vmin) ? v[u].y : vmin;
        flags |= fb[u];
      }
    }
  }
  if (flags & 2u) {
    // A NaN in the window: ODE's running dMAX (maxY = (maxY > h) ? maxY : h over the scan, x outer, z inner) makes the
    // maximum depend on the ORDER -- a NaN replaces it, what follows starts over -- and the ordered scan of the fallback
    // stages is needed.  Except where NOTHING in the window is finite or +inf (min over the finite samples = +inf; max with
    // NaN counted as -inf = -inf): every sample is then a NaN or -inf, every step of the fold takes h, and the result is the
    // LAST sample scanned, (maxX, maxZ): NaN or -inf.  That is the window of a state out in unknown / untraversable
    // terrain -- on a map with an unknown margin half of all boxes -- and with (maxY, minY = +inf, not all finite) the
    // exits decide it right here like the scan would.
    if (vmin == INFINITY && vmax == -INFINITY) {
      const unsigned last = gather32(t.fl, (unsigned)(b.maxX + b.maxZ * f.nW));  // level 0 = the samples themselves
      w.maxY = (last & 2u) ? __builtin_nanf("") : -INFINITY;
      w.minY = INFINITY;
      w.allFinite = false;
      return true;
    }
    return false;
  }
  w.allFinite = !(flags & 1u);
  w.maxY = vmax;
  w.minY = vmin;
  return true;
}

// ---- queue record ------------------------------------------------------------------------------------
struct __attribute__((aligned(16))) PendingBox {  // 96 bytes
  float pos[3];
  float R[9];
  float aabb[6];
  short minX, maxX, minZ, maxZ;
  unsigned state;
  unsigned kind;  // 0 = torso vs body layer (ok = no contact), 1 = foot vs masked layer (ok = contact)
  unsigned pad[2];
};

// The two big queues (torso, feet) are split into ARTP_NSUB sub-queues, each with its own slot counter on its own
// 128-byte line: a workgroup of the classify stage takes its slots from sub-queue blockIdx % ARTP_NSUB.  One
// counter word for all 32 768 workgroups of a 2^22-state batch serialised them on one L2 atomic unit (0.14 ms of
// the stage); the consumers walk sub-queue blockIdx % ARTP_NSUB, so their launch grids are multiples of ARTP_NSUB.
#ifndef ARTP_NSUB
#define ARTP_NSUB 16
#endif

struct PipelineQueues {
  PendingBox* q1;                // undecided boxes: torso sub-queue s at [s * seg_t, ...), foot sub-queue s at
                                 // [feet_base + s * 4 * seg_t, ...)
  unsigned long long* sub;       // slot counters: sub[(kind * ARTP_NSUB + s) * 16], kind 0 = torso, 1 = feet
  unsigned long long seg_t;      // capacity of one torso sub-queue; a foot sub-queue holds 4 * seg_t
  unsigned* q2;                  // indices into q1 of boxes that need the exact-grouping stage
  unsigned* q3;                  // indices into q1 of foot boxes that survive the lane scan stage
  unsigned* q4;                  // foot boxes whose exits the tables could not evaluate (lane-scan path)
  unsigned* q5;                  // foot boxes whose corner candidates may have partners (list pass)
  unsigned* q6;                  // torso boxes the streaming pass cannot finish (staged pass)
  unsigned long long* counters;  // [1] q2, [5] q3, [6] q5 counts (queues 4 and 6 are segmented: sub_fwd)
  unsigned long long feet_base;  // = ARTP_NSUB * seg_t
};

__device__ __forceinline__ unsigned long long* sub_counter(const PipelineQueues& q, int kind, int s) {
  return q.sub + (size_t)(kind * ARTP_NSUB + s) * 16;
}
// Consumer cursor of a sub-queue (same 128-byte line as its slot counter, which is final by then): the streaming
// kernels take their boxes in chunks through it instead of a static stride.  A group's 55 boxes cost anywhere
// between an early vertex hit and a full corner stage, and with a static stride the kernel lasts as long as its
// unluckiest group.
__device__ __forceinline__ unsigned long long* sub_cursor(const PipelineQueues& q, int kind, int s) {
  return sub_counter(q, kind, s) + 1;
}
// first record of sub-queue s of the torso (kind 0) / foot (kind 1) queue
__device__ __forceinline__ unsigned long long sub_base(const PipelineQueues& q, int kind, int s) {
  return kind ? q.feet_base + (unsigned long long)s * 4ull * q.seg_t : (unsigned long long)s * q.seg_t;
}

// Boxes a streaming kernel hands on to a fallback stage (foot boxes without a table verdict -> queue 4, torso boxes the
// streaming pass cannot finish -> queue 6).  On a map without unknown cells that is a handful; on a map WITH them (a band
// of NaN across the C2 map) it is millions per batch, and one atomicAdd per box on ONE counter word made the two streaming
// kernels 12 - 15 ms each (the L2 atomic unit does ~3 ns per operation).  So: the fallback queues are split like the big
// ones -- segment s of queue 4 / 6 belongs to sub-queue s, its counter shares the sub-queue's line (slot [2]) --, and a
// wavefront collects what it forwards per chunk (fwd_flush_slots) and it leaves with ONE atomic.
__device__ __forceinline__ unsigned long long* sub_fwd(const PipelineQueues& q, int kind, int s) {
  return sub_counter(q, kind, s) + 2;
}
__device__ __forceinline__ unsigned long long fwd_base(const PipelineQueues& q, int kind, int s) {  // segment start in q4 / q6
  return kind ? (unsigned long long)s * 4ull * q.seg_t : (unsigned long long)s * q.seg_t;
}
// A wavefront's forwards of one chunk: box k of the chunk puts (item + 1) into slot k of a few bytes of LDS -- written inside
// the (rare) branch that decides it, so the hot path carries no extra registers --, and at the end of the chunk the filled
// slots leave together.  CHUNK <= 64; slots are zero when the chunk starts (flush leaves them so).
template <int CHUNK>
__device__ __forceinline__ void fwd_flush_slots(unsigned* slots, unsigned long long* counter, unsigned* segment, int lane) {
  const unsigned v = lane < CHUNK ? slots[lane] : 0u;
  const unsigned long long bal = __ballot(v != 0u);
  if (bal == 0) return;
  unsigned long long base = 0;
  if (lane == 0) base = atomicAdd(counter, (unsigned long long)__popcll(bal));
  base = __shfl(base, 0, 64);
  if (v != 0u) {
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    segment[base + (unsigned long long)__popcll(bal & lt)] = v - 1u;
    slots[lane] = 0u;
  }
}
// item `it` of a segmented fallback queue (kind 1: queue 4, kind 0: queue 6); total = the sum of the segment counts
__device__ __forceinline__ unsigned long long fwd_total(const PipelineQueues& q, int kind) {
  unsigned long long t = 0;
  for (int s = 0; s < ARTP_NSUB; ++s) t += *sub_fwd(q, kind, s);
  return t;
}
__device__ __forceinline__ unsigned fwd_item(const PipelineQueues& q, int kind, const unsigned* queue, unsigned long long it) {
  int s = 0;
  for (; s < ARTP_NSUB - 1; ++s) {
    const unsigned long long c = *sub_fwd(q, kind, s);
    if (it < c) break;
    it -= c;
  }
  return queue[fwd_base(q, kind, s) + it];
}

__device__ __forceinline__ void box_from_record(const PendingBox& r, const RobotDev& rb, BoxHF& b) {
#pragma unroll
  for (int i = 0; i < 3; ++i) b.pos[i] = r.pos[i];
#pragma unroll
  for (int i = 0; i < 9; ++i) b.R[i] = r.R[i];
#pragma unroll
  for (int i = 0; i < 6; ++i) b.aabb[i] = r.aabb[i];
  const bool foot = (r.kind & 1u) != 0;
  b.side[0] = foot ? rb.foot[0] : rb.torso[0];
  b.side[1] = foot ? rb.foot[1] : rb.torso[1];
  b.side[2] = foot ? rb.foot[2] : rb.torso[2];
  b.minX = r.minX;
  b.maxX = r.maxX;
  b.minZ = r.minZ;
  b.maxZ = r.maxZ;
  b.on_field = 1;
}

// dPose of box k of a state (validity_checker.cpp:40-43, validity_checker_feet.cpp:64-68).
__device__ __forceinline__ void state_box_pose(const RobotDev& rb, const float t[3], const float R[9], int k,
                                               float pose[16]) {
  const bool body = (k == 0);
  const float ox = body ? rb.torso_off[0] : ((k <= 2) ? rb.feet_off_x : -rb.feet_off_x);
  const float oy = body ? rb.torso_off[1] : ((k & 1) ? rb.feet_off_y : -rb.feet_off_y);
  const float oz = body ? rb.torso_off[2] : 0.0f;
  // pose * Pose3FromXYZ(o): Eigen affine product, translation = R*o + t with the 3-term dot summed as
  // x0 + (x1 + x2) (Eigen's unrolled redux).
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
}

// record flags (PendingBox::kind, bit 0 = foot)
#define ARTP_REC_EXITS_NEGATIVE 0x400u  // exits (b)-(e) already evaluated (from the tables): none fired
#define ARTP_REC_ALL_FINITE 0x800u      // ... and the window holds no non-finite sample

// Probe of exit (f) "a colliding terrain vertex lies inside the box" on an N x N lattice of samples under
// the box (in an all-finite window every window vertex is a member of an all-finite triangle; otherwise the
// membership is looked up).  (f) is an existence test, hence any vertex found inside decides it; finding none
// decides nothing.  Only valid once exits (b)-(e) are ruled out (no NaN in the window).
template <int N>
__device__ __forceinline__ bool probe_vertices_inside(const FieldDev& f, const TablesDev& t, const BoxHF& b, float frac,
                                                      bool window_all_finite) {
  if (b.maxX - b.minX < 1 || b.maxZ - b.minZ < 1) return false;
  float h[N * N];
  unsigned nf[N * N];
  int ix[N * N], iz[N * N];
#pragma unroll
  for (int j = 0; j < N; ++j) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const float a = (N == 1) ? 0.0f : frac * ((float)(2 * i) / (float)(N - 1) - 1.0f) * b.side[0];
      const float c = (N == 1) ? 0.0f : frac * ((float)(2 * j) / (float)(N - 1) - 1.0f) * b.side[1];
      const float wx = b.pos[0] + a * b.R[0] + c * b.R[1];
      const float wz = b.pos[2] + a * b.R[6] + c * b.R[7];
      int x = (int)rintf(wx * f.inv_w), z = (int)rintf(wz * f.inv_d);
      x = x < b.minX ? b.minX : (x > b.maxX ? b.maxX : x);
      z = z < b.minZ ? b.minZ : (z > b.maxZ ? b.maxZ : z);
      ix[j * N + i] = x;
      iz[j * N + i] = z;
      h[j * N + i] = gather32(f.data, (unsigned)(x + z * f.nW));
      // A window with masked cells: the vertex only counts as a corner of an all-finite triangle OF THE WINDOW
      // (heightfield.cpp:1306-1441).  The 4 x 4 block anchored one sample before the vertex being all finite
      // and the vertex's neighbours lying in the window is sufficient (its ABC triangle qualifies); the flag
      // is fetched together with the height so the probe stays one round trip.
      nf[j * N + i] = 0u;
      if (!window_all_finite) {
        const bool inner = x > b.minX && x < b.maxX && z > b.minZ && z < b.maxZ;
        nf[j * N + i] = inner ? (unsigned)gather32(t.fl, (unsigned)((x - 1) + (z - 1) * f.nW)) : 1u;
      }
    }
  }
  bool hit = false;
#pragma unroll
  for (int k = 0; k < N * N; ++k)
    hit = hit || (nf[k] == 0u && is_finite(h[k]) && h[k] > b.aabb[2] &&
                  point_in_box(b, (float)ix[k] * f.sample_w, h[k], (float)iz[k] * f.sample_d));
  return hit;
}

// Box k of a state against ITS layer, first half: pose, AABB, index window and the conservative exits of the stride
// tables.  0 = decided ok, 1 = decided failing, ARTP_CODE_OPEN = not decided yet (`b` is complete).
// Called with the body layer for k = 0 and the feet layer for k = 1..4 from separate call sites:
// selecting the FieldDev / TablesDev kernel arguments by a per-lane index would force both structs into
// per-lane scratch memory (240 B/lane of HBM traffic).
#define ARTP_CODE_OPEN 4
__device__ __forceinline__ int classify_head(const FieldDev& f, const TablesDev& tab, const MapGeom& g,
                                             const RobotDev& rb, const float t[3], const float R[9],
                                             const float bR[9], int k, BoxHF& b, bool& cover_finite) {
  const bool body = (k == 0);
  cover_finite = false;
  float pose[16];
  state_box_pose(rb, t, R, k, pose);
  if (!map_is_inside(g, (double)pose[0], (double)pose[1])) {
    // body outside -> valid (validity_checker_body.cpp:29-32); foot outside ->
    // !unknown_space_untraversable (validity_checker_feet.cpp:34-37)
    return (!body && rb.unknown_space_untraversable) ? 1 : 0;
  }
  setup_box_rotated(f, pose, bR, body ? rb.torso[0] : rb.foot[0], body ? rb.torso[1] : rb.foot[1],
                    body ? rb.torso[2] : rb.foot[2], b);
  if (!b.on_field) return body ? 0 : 1;  // AABB off the field: no contact (heightfield.cpp:1868-1877)
  if (tab.valid) {
    int coarse = coarse_block_exit(f, tab, b, cover_finite);
    if (coarse < 0) coarse = body ? tight_cover_exit<3>(f, tab, b, cover_finite) : tight_cover_exit<2>(f, tab, b, cover_finite);
    if (coarse >= 0) return (body ? coarse : !coarse) ? 1 : 0;
  }
  return ARTP_CODE_OPEN;
}

// Second half, for the boxes the stride tables left open: exact window statistics, exits (b)-(e), and for feet
// the vertex probe.  0 = decided ok, 1 = decided failing, 2 = undecided (exits known not to fire), 3 = undecided
// (tables could not answer).
__device__ __forceinline__ int classify_tail(const FieldDev& f, const TablesDev& tab, const BoxHF& b, bool body,
                                             bool window_known_finite, bool& all_finite, long long* t_stats = nullptr) {
  WindowStats w;
  int hit = 0, ec;
  const bool have_stats = tab.valid && table_window_stats(f, tab, b, w, window_known_finite);
  all_finite = have_stats && w.allFinite;
#ifdef ARTP_STAGE_TIMING
  if (t_stats) *t_stats = (w.maxY > 1e30f) ? 0 : clock64();  // depends on the statistics: taken after they arrive
#endif
  if (!(have_stats && decide_exits(b, w, hit, ec))) {
    // feet only: 3/5 of the undecided foot boxes hold a vertex and the 2 x 2 probe finds most of them;
    // torso hits sit at the rim of the box (a 3 x 3 probe caught 1 in 4) and do not pay for the probe
    if (have_stats && !body && probe_vertices_inside<2>(f, tab, b, 0.33f, all_finite)) return 0;  // exit (f): the foot touches
    return have_stats ? 2 : 3;
  }
  return (body ? hit : !hit) ? 1 : 0;
}

// Queue record of a box into the LDS slot dst[0..5].  dst[5].z carries the classify stage's own bookkeeping (the
// lane the box came from); consumers ignore it.
__device__ __forceinline__ void stage_record(const BoxHF& b, bool body, unsigned state, unsigned origin, float4* dst,
                                             bool cover_finite = false) {
  dst[0] = make_float4(b.pos[0], b.pos[1], b.pos[2], b.R[0]);
  dst[1] = make_float4(b.R[1], b.R[2], b.R[3], b.R[4]);
  dst[2] = make_float4(b.R[5], b.R[6], b.R[7], b.R[8]);
  dst[3] = make_float4(b.aabb[0], b.aabb[1], b.aabb[2], b.aabb[3]);
  const unsigned wx = ((unsigned)(unsigned short)b.minX) | ((unsigned)(unsigned short)b.maxX << 16);
  const unsigned wz = ((unsigned)(unsigned short)b.minZ) | ((unsigned)(unsigned short)b.maxZ << 16);
  dst[4] = make_float4(b.aabb[4], b.aabb[5], __uint_as_float(wx), __uint_as_float(wz));
  dst[5] = make_float4(__uint_as_float(state), __uint_as_float(body ? 0u : 1u), __uint_as_float(origin),
                       __uint_as_float(cover_finite ? 1u : 0u));
}

__device__ __forceinline__ unsigned record_kind(bool body, bool exits_negative, bool all_finite) {
  return (body ? 0u : 1u) | (exits_negative ? ARTP_REC_EXITS_NEGATIVE : 0u) |
         (exits_negative && all_finite ? ARTP_REC_ALL_FINITE : 0u);
}

// ---- stage 1: one lane per (state, box), then one lane per OPEN box ---------------------------------------
// A workgroup owns SUB x 64 consecutive states and runs 5 x SUB wavefronts.
// Phase A: wavefront w handles box k = w / SUB (0 torso, 1..4 feet) of the 64 states of sub-block w % SUB, so the
// box index is wave-uniform (no divergence between torso and foot geometry, the layer is picked by a uniform
// branch) and the five table-lookup latency chains of one state run in five different wavefronts.  classify_head
// decides ~6 of 7 boxes out of the stride tables.
// Phase B: the boxes still open -- of states none of whose boxes has failed yet -- are compacted through LDS, in
// the queue's record format, into a torso list and a foot list, and ONE LANE PER LIST ENTRY runs classify_tail.
// A wavefront executes every branch one of its lanes takes: left in place, the ~1 in 7 open lanes made all ten
// wavefronts of the workgroup issue the exact-statistics / probe code (a third of the kernel's VALU instructions);
// compacted, two or three wavefronts do.
// Phase C: the five verdicts of a state meet in LDS; a state with a decided failing box is finished (label 0) and
// queues nothing.  Records still pending are already in LDS in queue format: every wavefront of phase B takes its
// queue slots with one atomic (sub-queue blockIdx % ARTP_NSUB: a single counter word sustains only ~88 returning
// atomics per microsecond, MI355X_MICROARCH.md "dequeue") and copies them out as full coalesced 16-byte lanes
// (scattered 16-byte stores into 96-byte records cost ~7x the bytes in partial-line write traffic).
// 64 states and five wavefronts per workgroup: smaller workgroups start and drain faster than 128-state ones
// (2^22 states: 0.555 -> 0.49 ms), and their open-box lists still fill most of a wavefront
#ifndef ARTP_CLASSIFY_SUB
#define ARTP_CLASSIFY_SUB 1
#endif
#define ARTP_CLASSIFY_THREADS (64 * 5 * ARTP_CLASSIFY_SUB)
#ifndef ARTP_CLASSIFY_CAP_T
#define ARTP_CLASSIFY_CAP_T 64   // open torso boxes a workgroup lists (all of its 64; typically ~8 are open)
#define ARTP_CLASSIFY_CAP_F 128  // open foot boxes it lists (of 256; typically ~40)
#endif
#ifdef ARTP_STAGE_TIMING
__device__ unsigned long long g_classify_cycles[2][8];  // [list waves | early-exit waves][phase], [7] = waves
#define ARTP_C_MARK(slot) do { const long long n_ = clock64(); c_acc[slot] += (unsigned long long)(n_ - c_prev); c_prev = n_; } while (0)
#define ARTP_C_FLUSH(grp) do { if (lane == 0) { for (int p_ = 0; p_ < 7; ++p_) atomicAdd(&g_classify_cycles[grp][p_], c_acc[p_]); atomicAdd(&g_classify_cycles[grp][7], 1ull); } } while (0)
#else
#define ARTP_C_MARK(slot) do { } while (0)
#define ARTP_C_FLUSH(grp) do { } while (0)
#endif

__global__ void __launch_bounds__(ARTP_CLASSIFY_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8)))
classify_states_kernel(FieldDev fb, FieldDev ff, TablesDev tb, TablesDev tf, MapGeom g, RobotDev rb,
                       const PoseRec* __restrict__ recs, size_t n, uint8_t* __restrict__ valid,
                       PipelineQueues q) {
  constexpr int SUB = ARTP_CLASSIFY_SUB, CAP_T = ARTP_CLASSIFY_CAP_T, CAP_F = ARTP_CLASSIFY_CAP_F;
  constexpr int WAVES_T = CAP_T / 64, WAVES_B = (CAP_T + CAP_F) / 64;  // phase B: torso list, then foot list
  // The PoseRecs of the workgroup's SUB * 64 states come in ONCE, cooperatively (one 16-byte chunk per lane, fully
  // coalesced), and are handed to the five box wavefronts of each state through LDS: five wavefronts pulling
  // the same 64-byte line per lane through the L1 were half of the kernel's L1 accesses.  80-byte stride:
  // conflict-free b128 reads.
  // The open-box records reuse the PoseRecs' LDS (the last PoseRec read is two barriers before the first record
  // write): 18 KB per workgroup, so LDS never limits how many workgroups a CU holds.
  static_assert((CAP_T + CAP_F) * 6 >= SUB * 64 * 5, "the record area also holds the PoseRecs");
  __shared__ float4 open_recs[(CAP_T + CAP_F) * 6];
  float4* prec = open_recs;
  __shared__ unsigned short plist[CAP_T + CAP_F];
  __shared__ uint8_t codes[SUB][5][64];
  __shared__ unsigned cnts[5 * SUB];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int k = wave / SUB, sub = wave % SUB;
  const bool body = (k == 0);
  const size_t i_raw = ((size_t)blockIdx.x * SUB + sub) * 64 + lane;
  const bool live = i_raw < n;
  const size_t i = live ? i_raw : n - 1;  // dead lanes shadow the last state and write nothing
#ifdef ARTP_STAGE_TIMING
  unsigned long long c_acc[7] = {0, 0, 0, 0, 0, 0, 0};
  long long c_prev = clock64();
#endif
  // the state's PoseRec: float pose + the box rotation in the field frame (shared by its five boxes)
  if (threadIdx.x < 2) cnts[threadIdx.x] = 0u;  // open torso / foot boxes of the workgroup (visible after the barrier)
  if (threadIdx.x < SUB * 64 * 4) {
    const int sl = threadIdx.x >> 2, part = threadIdx.x & 3;
    const size_t gi_raw = (size_t)blockIdx.x * SUB * 64 + sl;
    const size_t gi = gi_raw < n ? gi_raw : n - 1;
    // read once, 268 MB per batch: non-temporal, so that the records do not evict the 10 MB of exact tables from the L2s
    typedef float native_f4 __attribute__((ext_vector_type(4)));
    const native_f4 pv = __builtin_nontemporal_load(reinterpret_cast<const native_f4*>(recs + gi) + part);
    prec[sl * 5 + part] = make_float4(pv[0], pv[1], pv[2], pv[3]);
  }
  __syncthreads();
  ARTP_C_MARK(0);
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  int rank = 0;
  bool open, cover_finite;
  BoxHF b;
  {
    const float4* rp = &prec[(sub * 64 + lane) * 5];
    const float4 r0 = rp[0], r1 = rp[1], r2 = rp[2], r3 = rp[3];
    const float t[3] = {r0.x, r0.y, r0.z};
    const float bR[9] = {r1.w, r2.x, r2.y, r2.z, r2.w, r3.x, r3.y, r3.z, r3.w};
    float R[9];
    rot_from_quat(r0.w, r1.x, r1.y, r1.z, R);
    int code;
    if (body)
      code = classify_head(fb, tb, g, rb, t, R, bR, 0, b, cover_finite);
    else
      code = classify_head(ff, tf, g, rb, t, R, bR, k, b, cover_finite);
    if (!live) code = 0;
    codes[sub][k][lane] = (uint8_t)code;
    ARTP_C_MARK(1);
    __syncthreads();
    bool ok = true;
#pragma unroll
    for (int kk = 0; kk < 5; ++kk) ok = ok && (codes[sub][kk][lane] != 1);
    open = ok && code == ARTP_CODE_OPEN;
    // list slots: one LDS atomic per wavefront and list (no second barrier for a cross-wavefront prefix sum; the
    // order of the list, and so of the queue, is whatever order the wavefronts arrive in -- labels do not depend on it)
    const unsigned long long bal = __ballot(open);
    rank = __popcll(bal & lt_mask);
    if (bal) {
      unsigned base_w = 0;
      if (lane == 0) base_w = atomicAdd(&cnts[body ? 0 : 1], (unsigned)__popcll(bal));
      rank += (int)__shfl(base_w, 0);
    }
  }
  ARTP_C_MARK(2);
  // rank = position of this lane's box in its list (valid where `open`)
  if (open) {
    const int cap = body ? CAP_T : CAP_F;
    if (rank < cap) {
      stage_record(b, body, (unsigned)i, threadIdx.x, open_recs + 6 * ((body ? 0 : CAP_T) + rank), cover_finite);
    } else {
      // more open boxes than the list holds (it takes a map the stride tables cannot decide much on): the box goes
      // to the queue as it is, marked "tables could not answer", and the scan stages evaluate its exits.  One
      // atomic and six scattered 16-byte stores per box.
      float4 r6[6];
      stage_record(b, body, (unsigned)i, 0u, r6);
      const int sq = blockIdx.x % ARTP_NSUB, kind = body ? 0 : 1;
      const unsigned long long at = sub_base(q, kind, sq) + atomicAdd(sub_counter(q, kind, sq), 1ull);
      float4* out = reinterpret_cast<float4*>(q.q1 + at);
#pragma unroll
      for (int j = 0; j < 6; ++j) out[j] = r6[j];
      // a foot record without a table verdict is the lane scan's (queue 4; phase C lists the others of that kind)
      if (!body) q.q4[fwd_base(q, 1, sq) + atomicAdd(sub_fwd(q, 1, sq), 1ull)] = (unsigned)at;
      codes[sub][k][lane] = 3;
    }
  }
  __syncthreads();
  int n_t = (int)cnts[0], n_f = (int)cnts[1];
  if (n_t + n_f == 0) {  // uniform: nothing open in this workgroup
    ARTP_C_FLUSH(1);
    if (body && live) {
      bool ok = true;
#pragma unroll
      for (int kk = 0; kk < 5; ++kk) ok = ok && (codes[sub][kk][lane] != 1);
      valid[i] = (uint8_t)ok;
    }
    return;
  }
  if (n_t > CAP_T) n_t = CAP_T;
  if (n_f > CAP_F) n_f = CAP_F;
  // Phase B: wave 0 walks the torso list, waves [WAVES_T, WAVES_B) the foot list, one lane per entry.  Wavefronts
  // without a list, or with an empty one, leave here (the torso wavefronts 0 .. SUB-1 stay: they write the labels at
  // the end).  S_BARRIER waits on the surviving wavefronts of a workgroup only, and the wave slots freed let the
  // next workgroup start while this one's tail (two dependent trips to the exact tables, then the queue atomic) is in
  // flight.
  const bool list_t = wave < WAVES_T;
  const int slot = (int)threadIdx.x;  // list entry = record slot: torso [0, CAP_T), feet [CAP_T, CAP_T + CAP_F)
  const bool have = wave < WAVES_B && (list_t ? slot < n_t : slot - CAP_T < n_f);
  ARTP_C_MARK(3);
  if (wave >= SUB && !__any(have)) {
    ARTP_C_FLUSH(1);
    return;
  }
  int code2 = 0;
  bool all_finite = false;
  unsigned origin = 0, state = 0;
  if (have) {
    PendingBox rec;
    const float4* src = open_recs + 6 * slot;
    float4* dst = reinterpret_cast<float4*>(&rec);
#pragma unroll
    for (int j = 0; j < 6; ++j) dst[j] = src[j];
    BoxHF bb;
    box_from_record(rec, rb, bb);
    state = rec.state;
    origin = rec.pad[0];
#ifdef ARTP_STAGE_TIMING
    long long t_stats = 0;
    if (list_t)
      code2 = classify_tail(fb, tb, bb, true, rec.pad[1] != 0u, all_finite, &t_stats);
    else
      code2 = classify_tail(ff, tf, bb, false, rec.pad[1] != 0u, all_finite, &t_stats);
    t_stats = __shfl(t_stats, 0);
    if (t_stats) { c_acc[4] += (unsigned long long)(t_stats - c_prev); c_prev = t_stats; }
#else
    if (list_t)
      code2 = classify_tail(fb, tb, bb, true, rec.pad[1] != 0u, all_finite);
    else
      code2 = classify_tail(ff, tf, bb, false, rec.pad[1] != 0u, all_finite);
#endif
    codes[(origin >> 6) % SUB][(origin >> 6) / SUB][origin & 63] = (uint8_t)code2;
  }
  ARTP_C_MARK(5);
  __syncthreads();
  // phase C
  {
    bool pending = have && code2 >= 2;
    if (pending) {
      const int os = (origin >> 6) % SUB, ol = origin & 63;
#pragma unroll
      for (int kk = 0; kk < 5; ++kk) pending = pending && (codes[os][kk][ol] != 1);
    }
    const unsigned long long pbal = __ballot(pending);
    const int pc = __popcll(pbal);
    if (pc) {  // wave-uniform
      if (pending) {
        // the record's last 16 bytes with the verdict flags; its slot goes on the wavefront's copy list
        open_recs[6 * slot + 5] = make_float4(__uint_as_float(state), __uint_as_float(record_kind(list_t, code2 == 2, all_finite)), 0.0f, 0.0f);
        plist[wave * 64 + __popcll(pbal & lt_mask)] = (unsigned short)slot;
      }
      unsigned long long base = 0;
      if (lane == 0) {
        const int sq = blockIdx.x % ARTP_NSUB;
        base = sub_base(q, list_t ? 0 : 1, sq) + atomicAdd(sub_counter(q, list_t ? 0 : 1, sq), (unsigned long long)pc);
      }
      base = __shfl(base, 0);
      if (!list_t) {
        // foot records WITHOUT a table verdict (code 3: a NaN in the window, a window thinner than the smallest block) also go
        // on the lane scan's list (queue 4, the segment of this sub-queue) right here, one atomic per wavefront that has any
        // -- feet_stream_kernel used to forward them with an atomic per box on one word: 15 ms per batch on a map with a
        // band of unknown cells, 0.3 ms for everything else it does
        const bool nostat = pending && code2 == 3;
        const unsigned long long nbal = __ballot(nostat);
        if (nbal) {
          const int sq = blockIdx.x % ARTP_NSUB;
          unsigned long long fb = 0;
          if (lane == 0) fb = atomicAdd(sub_fwd(q, 1, sq), (unsigned long long)__popcll(nbal));
          fb = __shfl(fb, 0);
          if (nostat)
            q.q4[fwd_base(q, 1, sq) + fb + (unsigned long long)__popcll(nbal & lt_mask)] =
                (unsigned)(base + (unsigned long long)__popcll(pbal & lt_mask));
        }
      }
      wave_lds_sync();
      float4* out = reinterpret_cast<float4*>(q.q1 + base);
      const unsigned short* pl = plist + wave * 64;
      for (int j = lane; j < pc * 6; j += 64) {
        const int e = j / 6;
        out[j] = open_recs[6 * (int)pl[e] + (j - 6 * e)];  // plain stores: the stream kernels find the records in L2 / MALL
        //                                                     (non-temporal: classify -11 us, the two stream kernels +18)
      }
    }
  }
  // verdicts: every code is final (the open ones were replaced in phase B, in front of a barrier)
  if (body && live) {
    bool ok = true;
#pragma unroll
    for (int kk = 0; kk < 5; ++kk) ok = ok && (codes[sub][kk][lane] != 1);
    valid[i] = (uint8_t)ok;
  }
  ARTP_C_MARK(6);
  ARTP_C_FLUSH(0);
}

// ---- fallback stage: foot boxes without a table verdict (queue 4), one LANE per box ---------------------
// The tables cannot answer when the window holds a NaN (ODE's running dMAX makes the maximum depend on
// the scan order) or is thinner than the smallest block.  Such a window (~100 samples) is walked
// sequentially by one lane -- literally the reference's loop nest (x outer, z inner), NaN quirk included:
// maxY (running dMAX), minY over finite, allFinite and, speculatively, (f) "colliding vertex of an
// all-finite triangle inside the box"; then exits (b)-(e), then (f).  Boxes still undecided (they need the
// plane stage) are compacted into queue 3 for the lane-group stage, one atomic per wavefront.
#define ARTP_LANE_THREADS 256
// One lane per box is the way to get through MILLIONS of such boxes (a map with unknown regions: 64 boxes per wavefront in
// flight), and the slowest way to get through a handful: a lane on its own pays every memory latency and ~10^4 dependent
// instructions -- 50 us, the duration of the launch, for ONE box of an edge batch.  Below this many boxes the lane scan steps
// aside and the 16-lanes-per-box staged pass (PASS 1) does exits, (f) and the corner stage itself.
#ifndef ARTP_LANE_SCAN_MIN
#define ARTP_LANE_SCAN_MIN 65536ull
#endif
#define ARTP_STREAM_WAVES 4
#ifndef ARTP_TORSO_WGS_PER_CU
#define ARTP_TORSO_WGS_PER_CU 14  // 2 wavefronts each: 7 per SIMD (65 VGPRs)
#endif
#ifndef ARTP_TORSO_U
#define ARTP_TORSO_U 2  // loads in flight per lane in the torso vertex stream (the hot blocks are ~3 steps of 64 vertices)
#endif
#ifndef ARTP_FEET_U
#define ARTP_FEET_U 4  // loads in flight per lane in the feet vertex stream (a 10 x 10 window is 6 per lane in all)
#endif
#ifndef ARTP_FEET_CHUNK
#define ARTP_FEET_CHUNK 32
#endif
#ifndef ARTP_TORSO_CHUNK
#define ARTP_TORSO_CHUNK 8
#endif

__global__ void __launch_bounds__(ARTP_LANE_THREADS)
feet_lane_kernel(FieldDev ff, RobotDev rb, PipelineQueues q, uint8_t* __restrict__ valid) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  const unsigned long long count = fwd_total(q, 1);
  if (count < ARTP_LANE_SCAN_MIN) return;  // a few boxes: resolve_boxes_kernel<., 16, 1> takes them straight from queue 4
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  const unsigned long long rounds = (count + stride - 1) / stride;  // uniform trip count: ballots below
  for (unsigned long long rnd = 0; rnd < rounds; ++rnd) {
    const unsigned long long it = rnd * stride + (unsigned long long)blockIdx.x * blockDim.x + tid;
    const bool live = it < count;
    const unsigned long long item = live ? (unsigned long long)fwd_item(q, 1, q.q4, it) : q.feet_base;  // dead lanes: a valid address
    bool undecided = false;
    if (live) {
      const PendingBox rec = q.q1[item];
      if (valid[rec.state] != 0) {
        BoxHF b;
        box_from_record(rec, rb, b);
        const int numX = b.maxX - b.minX + 1, numZ = b.maxZ - b.minZ + 1;
        const int cellsX = numX - 1, cellsZ = numZ - 1;
        const float minO2 = b.aabb[2];
        const int nW = ff.nW;
        const float* base = ff.data + b.minX + (size_t)b.minZ * nW;
        WindowStats w;  // heightfield.cpp:1002-1026
        w.maxY = -INFINITY;
        w.minY = INFINITY;
        w.allFinite = true;
        bool vhit = false;
        for (int xl = 0; xl < numX; ++xl) {
          const float vx = (float)(b.minX + xl) * ff.sample_w;
          // eight samples of the column are requested before the first is looked at: a lane on its own pays the full
          // memory latency per dependent load, and ONE such box in a batch is the whole duration of this launch
          for (int zl0 = 0; zl0 < numZ; zl0 += 8) {
           float hh[8];
#pragma unroll
           for (int k = 0; k < 8; ++k) hh[k] = (zl0 + k < numZ) ? base[xl + (zl0 + k) * nW] : 0.0f;
#pragma unroll
           for (int k = 0; k < 8; ++k) {
            const int zl = zl0 + k;
            if (zl >= numZ) break;
            const float* c = base + xl + zl * nW;
            const float h = hh[k];
            w.maxY = (w.maxY > h) ? w.maxY : h;  // dMAX(maxY, h): a NaN replaces maxY
            if (is_finite(h)) {
              w.minY = (w.minY > h) ? h : w.minY;
              if (!vhit && h > minO2) {
                const float vz = (float)(b.minZ + zl) * ff.sample_d;
                if (point_in_box(b, vx, h, vz)) {
                  // member of a triangle whose three vertices are finite (the six neighbours of
                  // grp_vertex_pass)
                  const bool xm = xl > 0, xp = xl < cellsX, zm = zl > 0, zp = zl < cellsZ;
                  const bool f_xp = xp && is_finite(c[1]);
                  const bool f_xm = xm && is_finite(c[-1]);
                  const bool f_zp = zp && is_finite(c[nW]);
                  const bool f_zm = zm && is_finite(c[-nW]);
                  const bool f_xm_zp = xm && zp && is_finite(c[nW - 1]);
                  const bool f_xp_zm = xp && zm && is_finite(c[1 - nW]);
                  vhit = (f_xp && f_zp) || (f_xm && f_xm_zp) || (f_xm_zp && f_zp) || (f_zm && f_xp_zm) ||
                         (f_xp_zm && f_xp) || (f_zm && f_xm);
                }
              }
            } else {
              w.allFinite = false;
            }
           }
          }
        }
        int result = 0, ec;
        bool decided = !(rec.kind & ARTP_REC_EXITS_NEGATIVE) && decide_exits(b, w, result, ec);
        if (!decided && vhit) {
          result = 1;
          decided = true;
        }
        if (decided && result == 0) valid[rec.state] = 0;  // a foot that touches nothing fails the state
        undecided = !decided;
      }
    }
    const unsigned long long b3 = __ballot(undecided);
    unsigned long long base3 = 0;
    if (lane == 0 && b3) base3 = atomicAdd(&q.counters[5], (unsigned long long)__popcll(b3));
    base3 = __shfl(base3, 0, 64);
    if (undecided) q.q3[base3 + __popcll(b3 & lt_mask)] = (unsigned)item;
  }
}

// ---- stage 1c: the foot queue, one 16-lane row per box, straight off the map ---------------------------
// For the records whose exits (b)-(e) the tables already ruled out (all of them unless the window holds a
// NaN or is thinner than the smallest table block): (f) streamed from the map, then the list-free corner
// stage.  Decides everything except boxes whose corner candidates may have partners (-> queue 5, list
// pass).  Records without table verdict go to queue 4 (sequential lane scan with the running-dMAX quirk).
// 6 wavefronts per SIMD (80 VGPRs, nothing spilled): -2 % against 5; 7 lose it again.  (With LLVM's SLP pass on,
// the kernel needed 96 VGPRs + 14 spills for 5 wavefronts and 6 were out of reach.)
#ifdef ARTP_STAGE_TIMING
__device__ unsigned long long g_feet_cycles[4];  // stream cycles, corner cycles, boxes that reached the corners
#endif
#ifndef ARTP_FEET_WAVES_PER_SIMD
#define ARTP_FEET_WAVES_PER_SIMD 6
#endif
template <int WAVES>
__global__ void __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(ARTP_FEET_WAVES_PER_SIMD, ARTP_FEET_WAVES_PER_SIMD)))
feet_stream_kernel(FieldDev ff, RobotDev rb, PipelineQueues q, uint8_t* __restrict__ valid) {
  constexpr int G = 16, GPW = 4;
  const int lane = threadIdx.x & 63;
  const int gl = lane & (G - 1);
  const int sq = blockIdx.x % ARTP_NSUB;  // this workgroup's foot sub-queue
  const unsigned long long count = *sub_counter(q, 1, sq);
  const unsigned long long first = sub_base(q, 1, sq);
  // every wavefront takes ARTP_FEET_CHUNK boxes at a time from the sub-queue's cursor (4 rounds of its 4 groups)
  unsigned long long* cursor = sub_cursor(q, 1, sq);
  for (;;) {
    unsigned long long chunk = 0;
    if (lane == 0) chunk = atomicAdd(cursor, (unsigned long long)ARTP_FEET_CHUNK);
    chunk = __shfl(chunk, 0);
    if (chunk >= count) break;
    for (int r = 0; r < ARTP_FEET_CHUNK / GPW; ++r) {
      const unsigned long long it = chunk + (unsigned long long)(r * GPW + lane / G);
      if (it < count) {
        const unsigned long long item = first + it;
        const PendingBox rec = q.q1[item];
        if (valid[rec.state] != 0) {  // else another box of this state already failed
          if (rec.kind & ARTP_REC_EXITS_NEGATIVE) {  // (records without a table verdict are the lane scan's: classify listed them)
            BoxHF b;
            box_from_record(rec, rb, b);
#ifdef ARTP_STAGE_TIMING
            const long long tf0 = clock64();
#endif
            const bool touches = grp_vertex_stream<G, ARTP_FEET_U>(ff, b, lane, (rec.kind & ARTP_REC_ALL_FINITE) != 0);  // ~80 samples
#ifdef ARTP_STAGE_TIMING
            const long long tf1 = clock64();
            if (gl == 0) atomicAdd(&g_feet_cycles[0], (unsigned long long)(tf1 - tf0));
#endif
            if (!touches) {
              const int rr = grp_corner_stage_direct<G>(ff, b, lane);
#ifdef ARTP_STAGE_TIMING
              if (gl == 0) {
                atomicAdd(&g_feet_cycles[1], (unsigned long long)(clock64() - tf1));
                atomicAdd(&g_feet_cycles[2], 1ull);
              }
#endif
              if (gl == 0) {
                if (rr == 2)
                  q.q5[atomicAdd(&q.counters[6], 1ull)] = (unsigned)item;
                else if (rr == 0)
                  valid[rec.state] = 0;  // a foot that touches nothing fails the state
              }
            }
          }
        }
      }
    }
  }
}

// (stage 1c, round 5 -- feet_stream2_kernel, the corner stage on dense lanes: measured no faster, pipeline_variants.h)


#ifdef ARTP_STAGE_TIMING
// tuning aid (never in the shipped build): cycles per stage of resolve_boxes_kernel, summed over boxes
__device__ unsigned long long g_stage_cycles[2][10];
#define ARTP_T_MARK(slot)                    \
  do {                                       \
    const long long now_ = clock64();        \
    t_acc[slot] += (unsigned long long)(now_ - t_prev); \
    t_prev = now_;                           \
  } while (0)
#else
#define ARTP_T_MARK(slot) do { } while (0)
#endif

// ---- stage 2: one lane group per undecided box ---------------------------------------------------------
// G = 64: torso queue, one wavefront per box.  G = 16: foot queue, four boxes per wavefront.
// Static striding over the queue (a shared work cursor would serialise on one atomic word).
// PASS 0: torso queue, streaming only; what it cannot finish goes to queue 6.  PASS 3: queue 6, the staged
// torso path (LDS tile, exits, (f), list, partner search).  PASS 1: foot queue 3 (boxes without a table
// verdict, after the lane scan), fast -- no kept-triangle list; boxes whose corner candidates might have a
// partner go to queue 5.  PASS 2: queue 5 with the list and the partner search.
template <int WAVES, int G, int PASS>
__global__ void __launch_bounds__(64 * WAVES)
resolve_boxes_kernel(FieldDev fld, RobotDev rb, PipelineQueues q, uint8_t* __restrict__ valid,
                     ScratchCaps caps, int* __restrict__ error_flag, const float2* __restrict__ mm4 = nullptr) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int GPW = 64 / G;  // groups per wavefront
  const int lane = threadIdx.x & 63;
  const int gl = lane & (G - 1);
  const int unit_in_block = (threadIdx.x >> 6) * GPW + (lane / G);
  const WaveScratch s = carve_scratch(smem, unit_in_block, caps);
  // PASS 1 in "direct" mode (few boxes without a table verdict: the lane scan stepped aside) takes queue 4 itself and does
  // exits and (f); otherwise it takes what the lane scan left (queue 3), like PASS 2 takes queue 5: exits and (f) behind them
  const bool direct = PASS == 1 && fwd_total(q, 1) < ARTP_LANE_SCAN_MIN;
  const bool feet = (PASS == 1 && !direct) || PASS == 2;
  // PASS 0 walks torso sub-queue blockIdx % ARTP_NSUB; the other passes walk their (short) index queues
  const int sq = blockIdx.x % ARTP_NSUB;
  const unsigned long long count =
      PASS == 0 ? *sub_counter(q, 0, sq)
                : (PASS == 3 ? fwd_total(q, 0) : (direct ? fwd_total(q, 1) : q.counters[PASS == 1 ? 5 : 6]));
  const unsigned long long first = PASS == 0 ? sub_base(q, 0, sq) : 0ull;
  const unsigned long long nblk = PASS == 0 ? gridDim.x / ARTP_NSUB : gridDim.x;
  const unsigned long long blk = PASS == 0 ? blockIdx.x / ARTP_NSUB : blockIdx.x;
  const unsigned long long stride = nblk * WAVES * GPW;
#ifdef ARTP_STAGE_TIMING
  unsigned long long t_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const long long t_begin = clock64();
#endif
  // PASS 0: every wavefront takes ARTP_TORSO_CHUNK boxes at a time from the sub-queue's cursor (see sub_cursor);
  // the other passes stride statically over their short index queues
  unsigned long long next = PASS == 0 ? 0ull : blk * WAVES * GPW + unit_in_block, chunk_end = 0ull;
  // PASS 0: boxes handed on to the staged pass (queue 6) leave once per chunk
  __shared__ unsigned fwd_slots[PASS == 0 ? WAVES : 1][ARTP_TORSO_CHUNK];
  unsigned* fs = fwd_slots[PASS == 0 ? (threadIdx.x >> 6) : 0];
  if (PASS == 0 && lane < ARTP_TORSO_CHUNK) fs[lane] = 0u;
  unsigned long long chunk_first = 0ull;
  for (;;) {
    if constexpr (PASS == 0) {
      if (next >= chunk_end) {
        fwd_flush_slots<ARTP_TORSO_CHUNK>(fs, sub_fwd(q, 0, sq), q.q6 + fwd_base(q, 0, sq), lane);
        unsigned long long c0 = 0;
        if (lane == 0) c0 = atomicAdd(sub_cursor(q, 0, sq), (unsigned long long)ARTP_TORSO_CHUNK);
        c0 = __shfl(c0, 0);
        if (c0 >= count) break;
        next = c0;
        chunk_first = c0;
        chunk_end = c0 + ARTP_TORSO_CHUNK < count ? c0 + ARTP_TORSO_CHUNK : count;
      }
    } else {
      if (next >= count) break;
    }
    const unsigned long long it = next;
    next += PASS == 0 ? 1ull : stride;
    const unsigned long long item =
        PASS == 0 ? first + it
                  : (unsigned long long)(PASS == 1 ? (direct ? fwd_item(q, 1, q.q4, it) : q.q3[it])
                                                   : (PASS == 2 ? q.q5[it] : fwd_item(q, 0, q.q6, it)));
#ifdef ARTP_STAGE_TIMING
    long long t_prev = clock64();
#endif
    const PendingBox rec = q.q1[item];
    if (valid[rec.state] == 0) continue;        // another box of this state already failed
    BoxHF b;
    box_from_record(rec, rb, b);
    ARTP_T_MARK(0);
    int result = 0, ec, fast_r = 2;
    bool decided = false;
    if constexpr (PASS == 0) {
      // Streaming pass: the tables already ruled out exits (b)-(e) and found the window all finite, so
      // (f) runs straight off the map and the corner stage reads its few cells from the map too: no LDS
      // tile, no list.  A box whose candidates may have partners (or whose window is not known to be all
      // finite) goes to queue 6 for the staged pass (PASS 3) -- keeping that code out of this kernel keeps
      // its register count down.
      if (fld.partner_flags != nullptr && (rec.kind & ARTP_REC_ALL_FINITE)) {
        __shared__ unsigned short hot_blocks[WAVES][ARTP_HOT_BLOCKS];
        if (wave_vertex_stream_blocks<ARTP_TORSO_U>(fld, mm4, b, hot_blocks[threadIdx.x >> 6], lane)) {
          result = 1;
          decided = true;
        } else {
          ARTP_T_MARK(2);
          const int r = grp_corner_stage_direct<G>(fld, b, lane);
          if (r != 2) {
            result = r;
            decided = true;
          }
        }
        ARTP_T_MARK(4);
      }
      if (!decided) {
        if (gl == 0) fs[it - chunk_first] = (unsigned)item + 1u;
      } else if (gl == 0 && result != 0) {
        valid[rec.state] = 0;  // the torso touches
      }
      continue;
    }
    const int total = (b.maxX - b.minX + 1) * (b.maxZ - b.minZ + 1);
    if (total > s.cap_verts) {  // the staged passes hold the window in LDS
      if (gl == 0) atomicExch(error_flag, 1);
      continue;
    }
    WindowStats w;
    if (!decided) {
      grp_scan_window<G>(fld, b, s, lane, w);
      ARTP_T_MARK(1);
      // queue 5 boxes (feet) already went through exits and (f) in the lane-per-box stage
      decided = !feet && decide_exits(b, w, result, ec);
    }
    if (!decided) {
      const bool vhit = !feet && grp_vertex_pass<G>(fld, b, s, lane, w.allFinite);
      ARTP_T_MARK(2);
      if (vhit) {
        result = 1;
        decided = true;
      } else if (PASS == 1) {
        const int r = grp_plane_stage_corners<G>(fld, b, s, lane, 0, true);
        ARTP_T_MARK(4);
        if (r != 2) {
          result = r;
          decided = true;
        } else if (gl == 0) {
          const unsigned long long slot = atomicAdd(&q.counters[6], 1ull);
          q.q5[slot] = (unsigned)item;
        }
      } else if (PASS == 3 && fld.partner_flags != nullptr &&
                 (fast_r = grp_plane_stage_corners<G>(fld, b, s, lane, 0, true)) != 2) {
        ARTP_T_MARK(4);
        result = fast_r;
        decided = true;
      } else {
        const int T = grp_compact_triangles<G, true>(b, s, lane);
        ARTP_T_MARK(3);
        // T < 0: more kept triangles than the short list of this stage holds -> exact-grouping stage
#ifdef ARTP_STAGE_TIMING
        const int r = (T == 0) ? 0 : (T < 0 ? 2 : grp_plane_stage_corners<G>(fld, b, s, lane, T, false, t_acc));
#else
        const int r = (T == 0) ? 0 : (T < 0 ? 2 : grp_plane_stage_corners<G>(fld, b, s, lane, T));
#endif
        ARTP_T_MARK(4);
        if (r != 2) {
          result = r;
          decided = true;
        } else if (gl == 0) {  // a corner candidate has an epsilon-equal partner: exact grouping
          const unsigned long long slot = atomicAdd(&q.counters[1], 1ull);
          q.q2[slot] = (unsigned)item;
        }
      }
    }
    if (decided && gl == 0) {
      const bool ok = (rec.kind & 1u) ? (result != 0) : (result == 0);
      if (!ok) valid[rec.state] = 0;
    }
    wave_lds_sync();
  }
#ifdef ARTP_STAGE_TIMING
  if (gl == 0) {
    for (int k = 0; k < 8; ++k) atomicAdd(&g_stage_cycles[G == 64 ? 0 : 1][k], t_acc[k]);
    atomicAdd(&g_stage_cycles[G == 64 ? 0 : 1][8], (unsigned long long)(clock64() - t_begin));
    atomicAdd(&g_stage_cycles[G == 64 ? 0 : 1][9], 1ull);
  }
#endif
}

// ---- stage 3: plane stage ---------------------------------------------------------------------------
template <int WAVES>
__global__ void __launch_bounds__(64 * WAVES)
plane_stage_kernel(FieldDev fb, FieldDev ff, RobotDev rb, PipelineQueues q, uint8_t* __restrict__ valid,
                   ScratchCaps caps, int* __restrict__ error_flag) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const WaveScratch s = carve_scratch(smem, threadIdx.x >> 6, caps);
  const unsigned long long count = q.counters[1];
  const unsigned long long stride = (unsigned long long)gridDim.x * WAVES;
  for (unsigned long long item = (unsigned long long)blockIdx.x * WAVES + (threadIdx.x >> 6); item < count;
       item += stride) {
    const PendingBox rec = q.q1[q.q2[item]];
    if (valid[rec.state] == 0) continue;
    BoxHF b;
    box_from_record(rec, rb, b);
    const bool foot = (rec.kind & 1u) != 0;
    WindowStats w;
    if (foot)
      wave_scan_window(ff, b, s, lane, w);
    else
      wave_scan_window(fb, b, s, lane, w);
    const int T = wave_compact_triangles<true>(b, s, lane);
    if (T < 0) {
      if (lane == 0) atomicExch(error_flag, 1);
      continue;
    }
    const int result = (T > 0 && (foot ? wave_plane_stage(ff, b, s, lane, T) : wave_plane_stage(fb, b, s, lane, T))) ? 1 : 0;
    if (lane == 0) {
      const bool ok = (rec.kind & 1u) ? (result != 0) : (result == 0);
      if (!ok) valid[rec.state] = 0;
    }
    wave_lds_sync();
  }
}

#ifdef ARTP_VARIANTS
#include "pipeline_variants.h"   // inside namespace artp
#endif

}  // namespace artp
