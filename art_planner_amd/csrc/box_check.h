// box_check.h -- wave-cooperative box-vs-heightfield zone test for gfx950 (one 64-lane wavefront
// decides one box), written as separable stages so the batch pipeline can run them in different
// kernels.
//
// Restates dxHeightfield::dCollideHeightfieldZone (ode/ode/src/heightfield.cpp:973-1789, with the
// art_planner patches at :989,1020-1024,1052-1064,1139,1329-1378) as a data-parallel algorithm with
// IDENTICAL results for "at most one contact requested" (HeightMapBoxChecker::checkCollision,
// art_planner/src/validity_checker/height_map_box_checker.cpp:67):
//
//  (a) wave_scan_window: the index window is streamed from HBM/L2 row-major-coalesced (x is the fast
//      axis of the ODE sample layout) into an LDS tile while max / min-of-finite / all-finite are
//      reduced across the wave; the running `maxY = dMAX(maxY, h)` NaN quirk (:1019) is honoured
//      through the scan index of the last NaN sample (ODE scans x-outer, z-inner);
//  (b)(c)(d)(e) decide_exits: the early-outs, a pure function of (maxY, minY, allFinite, box);
//  (f) wave_vertex_pass: "terrain vertex inside the box": every window vertex that is colliding
//      (finite, above the box bottom) and belongs to a triangle whose three vertices are finite is
//      tested lane-parallel; any hit ends the check (order independent because the reference returns
//      1 at the first hit);
//  (g) wave_compact_triangles + wave_plane_stage: kept triangles are compacted IN THE REFERENCE'S
//      BUFFER ORDER into an LDS list.  A triangle whose plane offset d has no epsilon-close partner
//      among the others can never be grouped with another triangle (|d_k - d_m| < eps is one of the
//      four conditions of the greedy grouping, :1539-1544), so it forms a group of its own whatever
//      happens to the rest: its box-plane contacts are tested lane-parallel.  Partner candidates are
//      found with two staggered quantisation grids in an LDS hash table (conservative: no false
//      negatives); only they run the sequential greedy grouping, one plane group per step, members
//      compared lane-parallel.  The bubble sort of planes (:933-955,1559) only orders contact
//      generation and cannot change "is there a contact", so it is not needed;
//  (h) the final vertex pass (:1651-1719) can only re-test vertices already rejected in (f) for box
//      geoms (same pure function, same arguments), so it never produces a contact and is omitted.
#pragma once

#include "artp_math.h"

namespace artp {

enum ExitCode : int {
  EXIT_AABB_OFF = 0,
  EXIT_ABOVE = 1,
  EXIT_UNDER = 2,
  EXIT_SPANS = 3,
  EXIT_FLAT_PLANE = 4,
  EXIT_VERTEX = 5,
  EXIT_PLANE = 6,
  EXIT_VERTEX2 = 7,
  EXIT_NONE = 8
};

// ---- lane-group primitives ------------------------------------------------------------------------
// A box is decided by a GROUP of G lanes: G = 64 (a whole wavefront: torso windows, ~1000 samples) or
// G = 16 (one DPP row: foot windows, ~70 samples, four boxes per wavefront).  All lanes of a group
// follow the same control flow; different groups of one wavefront may diverge.
template <int G>
__device__ __forceinline__ int grp_lane(int lane) { return lane & (G - 1); }

template <int G>
__device__ __forceinline__ unsigned long long grp_ballot(bool p, int lane) {
  const unsigned long long bal = __ballot(p);
  if (G == 64) return bal;
  return (bal >> (lane & ~(G - 1))) & ((1ull << (G & 63)) - 1ull);
}
template <int G>
__device__ __forceinline__ bool grp_any(bool p, int lane) { return grp_ballot<G>(p, lane) != 0ull; }

// lanes of the group strictly below this lane
template <int G>
__device__ __forceinline__ unsigned long long grp_lt_mask(int lane) {
  const int gl = lane & (G - 1);
  return gl == 0 ? 0ull : (~0ull >> (64 - gl));
}

#define ARTP_DPP_QUAD_XOR1 0xB1        /* quad_perm:[1,0,3,2] */
#define ARTP_DPP_QUAD_XOR2 0x4E        /* quad_perm:[2,3,0,1] */
#define ARTP_DPP_ROW_HALF_MIRROR 0x141
#define ARTP_DPP_ROW_MIRROR 0x140

template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false); }

// max / min over the group (max and min are idempotent, so butterfly order is irrelevant; values are
// compared with the same (o > v) ? o : v select the scan uses).  4 DPP steps inside a 16-lane row,
// then the four row results are combined through scalar readlanes for G = 64.
template <int G, bool IS_MAX>
__device__ __forceinline__ float grp_minmax(float v) {
#define ARTP_STEP(CTRL)                                                  \
  {                                                                      \
    const float o = __int_as_float(dpp_i<CTRL>(__float_as_int(v)));      \
    v = IS_MAX ? ((o > v) ? o : v) : ((o < v) ? o : v);                  \
  }
  ARTP_STEP(ARTP_DPP_QUAD_XOR1)
  ARTP_STEP(ARTP_DPP_QUAD_XOR2)
  ARTP_STEP(ARTP_DPP_ROW_HALF_MIRROR)
  ARTP_STEP(ARTP_DPP_ROW_MIRROR)
#undef ARTP_STEP
  if (G == 64) {
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    const float a = IS_MAX ? ((r1 > r0) ? r1 : r0) : ((r1 < r0) ? r1 : r0);
    const float c = IS_MAX ? ((r3 > r2) ? r3 : r2) : ((r3 < r2) ? r3 : r2);
    v = IS_MAX ? ((c > a) ? c : a) : ((c < a) ? c : a);
  }
  return v;
}
template <int G>
__device__ __forceinline__ float grp_max(float v) { return grp_minmax<G, true>(v); }
template <int G>
__device__ __forceinline__ float grp_min(float v) { return grp_minmax<G, false>(v); }

template <int G>
__device__ __forceinline__ int grp_max_i(int v) {
#define ARTP_STEP(CTRL)              \
  {                                  \
    const int o = dpp_i<CTRL>(v);    \
    v = (o > v) ? o : v;             \
  }
  ARTP_STEP(ARTP_DPP_QUAD_XOR1)
  ARTP_STEP(ARTP_DPP_QUAD_XOR2)
  ARTP_STEP(ARTP_DPP_ROW_HALF_MIRROR)
  ARTP_STEP(ARTP_DPP_ROW_MIRROR)
#undef ARTP_STEP
  if (G == 64) {
    const int r0 = __builtin_amdgcn_readlane(v, 0), r1 = __builtin_amdgcn_readlane(v, 16);
    const int r2 = __builtin_amdgcn_readlane(v, 32), r3 = __builtin_amdgcn_readlane(v, 48);
    const int a = r1 > r0 ? r1 : r0, c = r3 > r2 ? r3 : r2;
    v = c > a ? c : a;
  }
  return v;
}

__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    const int o = __shfl_xor(v, m, 64);
    v = (o < v) ? o : v;
  }
  return v;
}

__device__ __forceinline__ void wave_lds_sync() {
  // LDS traffic of one wave is ordered; make the compiler respect the cross-lane dependency.
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// Per-wave LDS scratch.  Stages that do not need a member get a null pointer / zero capacity.
struct WaveScratch {
  float* cand;          // corner-candidate planes, raw cross products and ids (cand bytes, may be null)
  float* h;             // heights tile, z-major rows of numX; capacity cap_verts
  unsigned short* tri;  // kept-triangle list; id = cx | cz << 6 | down << 12 (window-local cell); cap_tris
  unsigned* tab;        // open-addressing hash table of quantised plane offsets; tab_size (pow2)
  int cap_verts;
  int cap_tris;
  int tab_size;
};

struct WindowStats {
  float maxY, minY;
  bool allFinite;
};

// (b)(c)(d)(e): returns true when the check is decided (result/exit_code set).
ARTP_HD bool decide_exits(const BoxHF& b, const WindowStats& w, int& result, int& exit_code) {
  const float minO2 = b.aabb[2], maxO2 = b.aabb[3];
  if (minO2 - w.maxY > -ARTP_EPS) {
    exit_code = EXIT_ABOVE;
    result = 0;
    return true;
  }
  if (w.minY - maxO2 > -ARTP_EPS) {
    exit_code = EXIT_UNDER;
    result = 0;
    return true;
  }
  if (w.allFinite && w.minY - minO2 > -ARTP_EPS && maxO2 - w.maxY > -ARTP_EPS) {
    exit_code = EXIT_SPANS;
    result = 1;
    return true;
  }
  if (w.allFinite && (w.maxY - w.minY < ARTP_EPS)) {
    float cpos[4][3];
    exit_code = EXIT_FLAT_PLANE;
    result = box_plane_contacts(b, 0.0f, 1.0f, 0.0f, w.minY, 1, cpos) ? 1 : 0;
    return true;
  }
  return false;
}

// (a).  Requires b.on_field and numX*numZ <= s.cap_verts (checked by the caller).
template <int G>
__device__ __forceinline__ void grp_scan_window(const FieldDev& f, const BoxHF& b, const WaveScratch& s,
                                                int lane, WindowStats& w) {
  const int gl = grp_lane<G>(lane);
  const int numX = b.maxX - b.minX + 1;
  const int numZ = b.maxZ - b.minZ + 1;
  const int total = numX * numZ;
  float lmax = -INFINITY, lmin = INFINITY;
  int lnonfinite = 0;
  int llast_nan = -1;  // ODE scan index (xl*numZ + zl) of this lane's last NaN
  const int qz = G / numX, rx = G - qz * numX;  // advance of (xl, zl) per G elements
  {
    int xl = gl % numX, zl = gl / numX;
    const unsigned base = (unsigned)(b.minX + b.minZ * f.nW);  // sample index: 32-bit offsets from the uniform f.data
    // 8 independent loads in flight per lane before the first use (the window comes from L2: a
    // dependent load per step would pay the full L2 latency 17 times for a torso window)
    constexpr int U = 8;
    for (int e0 = gl; e0 < total; e0 += G * U) {
      float hv[U];
      int xs[U], zs[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        xs[u] = xl;
        zs[u] = zl;
        hv[u] = (e0 + G * u < total) ? gather32(f.data, base + (unsigned)(xl + zl * f.nW)) : 0.0f;
        xl += rx;
        zl += qz;
        if (xl >= numX) {
          xl -= numX;
          zl += 1;
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e = e0 + G * u;
        if (e < total) {
          const float h = hv[u];
          s.h[e] = h;
          lmax = (h > lmax) ? h : lmax;  // NaN never wins here; handled below
          if (is_finite(h)) {
            lmin = (lmin > h) ? h : lmin;
          } else {
            lnonfinite = 1;
            if (is_nan(h)) {
              const int si = xs[u] * numZ + zs[u];
              llast_nan = si > llast_nan ? si : llast_nan;
            }
          }
        }
      }
    }
  }
  wave_lds_sync();
  w.maxY = grp_max<G>(lmax);
  w.minY = grp_min<G>(lmin);
  w.allFinite = !grp_any<G>(lnonfinite != 0, lane);
  if (f.has_nan && !w.allFinite) {
    const int last_nan = grp_max_i<G>(llast_nan);
    if (last_nan >= 0) {
      // running dMAX: the maximum restarts after every NaN, and a trailing NaN survives.
      if (last_nan == total - 1) {
        w.maxY = __uint_as_float(0x7fc00000u);
      } else {
        float m2 = -INFINITY;
        int xl = gl % numX, zl = gl / numX;
        for (int e = gl; e < total; e += G) {
          const float h = s.h[e];
          if (xl * numZ + zl > last_nan) m2 = (h > m2) ? h : m2;
          xl += rx;
          zl += qz;
          if (xl >= numX) {
            xl -= numX;
            zl += 1;
          }
        }
        w.maxY = grp_max<G>(m2);
      }
    }
  }
}

// (f).  Heights must be staged in s.h.
template <int G>
__device__ __forceinline__ bool grp_vertex_pass(const FieldDev& f, const BoxHF& b, const WaveScratch& s,
                                                int lane, bool allFinite) {
  const int gl = grp_lane<G>(lane);
  const int numX = b.maxX - b.minX + 1;
  const int numZ = b.maxZ - b.minZ + 1;
  const int total = numX * numZ;
  const int cellsX = numX - 1, cellsZ = numZ - 1;
  const float minO2 = b.aabb[2];
  bool hit = false;
  const int qz = G / numX, rx = G - qz * numX;
  int xl = gl % numX, zl = gl / numX;
  int step = 0;
  for (int e0 = 0; e0 < total; e0 += G, ++step) {
    const int e = e0 + gl;
    if (e < total) {
      const float h = s.h[e];
      const bool coll = is_finite(h) && (h > minO2);
      if (coll) {
        bool member;
        if (allFinite) {
          member = (cellsX > 0) && (cellsZ > 0);
        } else {
          // finite flags of the 6 neighbours that share a triangle with (xl, zl)
          const bool xm = xl > 0, xp = xl < cellsX, zm = zl > 0, zp = zl < cellsZ;
          const bool f_xp = xp && is_finite(s.h[e + 1]);                  // (xl+1, zl)
          const bool f_xm = xm && is_finite(s.h[e - 1]);                  // (xl-1, zl)
          const bool f_zp = zp && is_finite(s.h[e + numX]);               // (xl, zl+1)
          const bool f_zm = zm && is_finite(s.h[e - numX]);               // (xl, zl-1)
          const bool f_xm_zp = xm && zp && is_finite(s.h[e + numX - 1]);  // (xl-1, zl+1)
          const bool f_xp_zm = xp && zm && is_finite(s.h[e - numX + 1]);  // (xl+1, zl-1)
          member = (f_xp && f_zp)         // A of cell (xl, zl): ABC
                   || (f_xm && f_xm_zp)   // B of cell (xl-1, zl): ABC
                   || (f_xm_zp && f_zp)   // B of cell (xl-1, zl): DBC
                   || (f_zm && f_xp_zm)   // C of cell (xl, zl-1): ABC
                   || (f_xp_zm && f_xp)   // C of cell (xl, zl-1): DBC
                   || (f_zm && f_xm);     // D of cell (xl-1, zl-1): DBC
        }
        if (member) {
          const float vx = (float)(b.minX + xl) * f.sample_w;
          const float vz = (float)(b.minZ + zl) * f.sample_d;
          hit = hit || point_in_box(b, vx, h, vz);
        }
      }
    }
    xl += rx;
    zl += qz;
    if (xl >= numX) {
      xl -= numX;
      zl += 1;
    }
    if ((step & 3) == 3 && grp_any<G>(hit, lane)) return true;  // group-wide poll every 4th step
  }
  return grp_any<G>(hit, lane);
}

// (f) straight from global memory (no LDS tile); the test is order-free.  In an all-finite window every
// colliding vertex belongs to an all-finite triangle; otherwise the six neighbours that share a triangle
// with the vertex are looked up (only for colliding vertices inside the box: rare).  8 loads in flight per
// lane; the group polls for a hit after every chunk.
template <int G, int U = 8>
__device__ __forceinline__ bool grp_vertex_stream(const FieldDev& f, const BoxHF& b, int lane, bool all_finite) {
  const int gl = grp_lane<G>(lane);
  const int numX = b.maxX - b.minX + 1;
  const int numZ = b.maxZ - b.minZ + 1;
  if (numX < 2 || numZ < 2) return false;  // no cell, no triangle, no member vertex
  const int cellsX = numX - 1, cellsZ = numZ - 1;
  const float minO2 = b.aabb[2];
  // The index window is the AABB widened to the sample grid (+ a sample of rounding slack on every side), and
  // a vertex inside the box is inside the AABB: the outermost rows / columns are skipped when they lie outside
  // it (10 um of slack against the roundings of the AABB; samples are centimetres apart).
  const float slack = 1.0e-5f;
  const int x0 = ((float)b.minX * f.sample_w < b.aabb[0] - slack) ? 1 : 0;
  const int x1 = ((float)b.maxX * f.sample_w > b.aabb[1] + slack) ? 1 : 0;
  const int z0 = ((float)b.minZ * f.sample_d < b.aabb[4] - slack) ? 1 : 0;
  const int z1 = ((float)b.maxZ * f.sample_d > b.aabb[5] + slack) ? 1 : 0;
  const int inX = numX - x0 - x1, inZ = numZ - z0 - z1;
  if (inX < 1 || inZ < 1) return false;
  const int total = inX * inZ;
  const int qz = G / inX, rx = G - qz * inX;
  int xl = gl % inX, zl = gl / inX;  // position in the inner grid; window-local = (x0 + xl, z0 + zl)
  const int nW = f.nW;
  const unsigned base = (unsigned)((b.minX + x0) + (b.minZ + z0) * nW);
  bool hit = false;
  for (int e0 = gl; e0 < total; e0 += G * U) {
    float hv[U];
    int xs[U], zs[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      xs[u] = xl;
      zs[u] = zl;
      hv[u] = (e0 + G * u < total) ? gather32(f.data, base + (unsigned)(xl + zl * nW)) : -INFINITY;
      xl += rx;
      zl += qz;
      if (xl >= inX) {
        xl -= inX;
        zl += 1;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float h = hv[u];
      if (is_finite(h) && h > minO2 && !hit &&
          point_in_box(b, (float)(b.minX + x0 + xs[u]) * f.sample_w, h, (float)(b.minZ + z0 + zs[u]) * f.sample_d)) {
        if (all_finite) {
          hit = true;
        } else {
          const float* c = f.data + base + xs[u] + zs[u] * nW;
          const int wx = x0 + xs[u], wz = z0 + zs[u];  // window-local: the triangles counted are the window's
          const bool xm = wx > 0, xp = wx < cellsX, zm = wz > 0, zp = wz < cellsZ;
          const bool f_xp = xp && is_finite(c[1]);
          const bool f_xm = xm && is_finite(c[-1]);
          const bool f_zp = zp && is_finite(c[nW]);
          const bool f_zm = zm && is_finite(c[-nW]);
          const bool f_xm_zp = xm && zp && is_finite(c[nW - 1]);
          const bool f_xp_zm = xp && zm && is_finite(c[1 - nW]);
          hit = (f_xp && f_zp) || (f_xm && f_xm_zp) || (f_xm_zp && f_zp) || (f_zm && f_xp_zm) ||
                (f_xp_zm && f_xp) || (f_zm && f_xm);
        }
      }
    }
    if (grp_any<G>(hit, lane)) return true;
  }
  return false;
}

// (f) for a torso box in an all-finite window, block form.  Only a vertex above the box's lowest point (h > minO2)
// can lie inside it, and under a torso that is 1 vertex in 15 of the ~900 of its window, in a few clusters.  The exact
// range table of 4 x 4 blocks (any anchor) says which blocks hold such a vertex at all: the wavefront looks up the
// maxima of the window's ~80 blocks (the last block of a row / column is pulled back inside the window), packs the
// ids of the blocks with max > minO2 into `hot` (LDS) and streams only their vertices -- the same vertices the full
// scan would have passed on to point_in_box (max <= minO2 means none of the 16 passes the pre-filter), a few of
// them twice where pulled-back blocks overlap.  Existence test: order and repeats do not matter.
#define ARTP_HOT_BLOCKS 512
template <int U>
__device__ __forceinline__ bool wave_vertex_stream_blocks(const FieldDev& f, const float2* __restrict__ mm4,
                                                          const BoxHF& b, unsigned short* hot, int lane) {
  constexpr int G = 64;
  const int numX = b.maxX - b.minX + 1;
  const int numZ = b.maxZ - b.minZ + 1;
  const float minO2 = b.aabb[2];
  const float slack = 1.0e-5f;  // see grp_vertex_stream
  const int x0 = ((float)b.minX * f.sample_w < b.aabb[0] - slack) ? 1 : 0;
  const int x1 = ((float)b.maxX * f.sample_w > b.aabb[1] + slack) ? 1 : 0;
  const int z0 = ((float)b.minZ * f.sample_d < b.aabb[4] - slack) ? 1 : 0;
  const int z1 = ((float)b.maxZ * f.sample_d > b.aabb[5] + slack) ? 1 : 0;
  const int inX = numX - x0 - x1, inZ = numZ - z0 - z1;
  const int nbx = (inX + 3) >> 2, nbz = (inZ + 3) >> 2, nb = nbx * nbz;
  if (mm4 == nullptr || inX < 4 || inZ < 4 || nb > ARTP_HOT_BLOCKS) return grp_vertex_stream<G, U>(f, b, lane, true);
  const int xs0 = b.minX + x0, zs0 = b.minZ + z0;              // first inner sample
  const int xl0 = xs0 + inX - 4, zl0 = zs0 + inZ - 4;          // last admissible block anchor
  const int nW = f.nW;
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  int nhot = 0;
  for (int j0 = 0; j0 < nb; j0 += G) {  // uniform
    const int j = j0 + lane;
    bool is_hot = false;
    const int bx = j % nbx, bz = j / nbx;
    if (j < nb) {
      const int ax = min(xs0 + 4 * bx, xl0), az = min(zs0 + 4 * bz, zl0);
      is_hot = gather32(mm4, (unsigned)(ax + az * nW)).x > minO2;
    }
    const unsigned long long m = __ballot(is_hot);
    if (is_hot) hot[nhot + __popcll(m & lt_mask)] = (unsigned short)(bx | (bz << 8));
    nhot += __popcll(m);
  }
  if (nhot == 0) return false;
  wave_lds_sync();
  const int total = nhot * 16;
  bool hit = false;
  for (int e0 = lane; e0 - lane < total; e0 += G * U) {  // uniform trip count
    float hv[U];
    int ix[U], iz[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = e0 + G * u;
      hv[u] = -INFINITY;
      ix[u] = 0;
      iz[u] = 0;
      if (e < total) {
        const int j = hot[e >> 4], v = e & 15;
        const int bx = j & 255, bz = j >> 8;
        ix[u] = min(xs0 + 4 * bx, xl0) + (v & 3);
        iz[u] = min(zs0 + 4 * bz, zl0) + (v >> 2);
        hv[u] = gather32(f.data, (unsigned)(ix[u] + iz[u] * nW));
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float h = hv[u];
      if (is_finite(h) && h > minO2 && !hit &&
          point_in_box(b, (float)ix[u] * f.sample_w, h, (float)iz[u] * f.sample_d))
        hit = true;
    }
    if (__any(hit)) return true;
  }
  return false;
}

// Kept triangles of the window in the reference's buffer order (x_local outer, z_local inner, ABC
// before DBC; :1306-1441).  With WRITE_LIST the ids go to s.tri; returns T, or -1 on list overflow.
template <int G, bool WRITE_LIST>
__device__ __forceinline__ int grp_compact_triangles(const BoxHF& b, const WaveScratch& s, int lane) {
  const int gl = grp_lane<G>(lane);
  const int numX = b.maxX - b.minX + 1;
  const int numZ = b.maxZ - b.minZ + 1;
  const int cellsX = numX - 1, cellsZ = numZ - 1;
  const int ncells = cellsX * cellsZ;
  const float minO2 = b.aabb[2];
  int T = 0;
  const unsigned long long lt_mask = grp_lt_mask<G>(lane);
  const int cz_safe = cellsZ > 0 ? cellsZ : 1;
  const int qc = G / cz_safe, rc = G - qc * cz_safe;  // advance of (cx, cz) per G cells
  int cx = gl / cz_safe, cz = gl - cx * cz_safe;
  for (int c0 = 0; c0 < ncells; c0 += G) {
    const int c = c0 + gl;
    bool keepUp = false, keepDown = false;
    if (c < ncells) {
      const int e = cz * numX + cx;
      const float hA = s.h[e], hB = s.h[e + 1], hC = s.h[e + numX], hD = s.h[e + numX + 1];
      const bool fA = is_finite(hA), fB = is_finite(hB), fC = is_finite(hC), fD = is_finite(hD);
      const bool cA = fA && hA > minO2, cB = fB && hB > minO2, cC = fC && hC > minO2,
                 cD = fD && hD > minO2;
      keepUp = (cA || cB || cC) && (fA && fB && fC);
      keepDown = (cB || cC || cD) && (fB && fC && fD);
    }
    const unsigned long long bu = grp_ballot<G>(keepUp, lane), bd = grp_ballot<G>(keepDown, lane);
    const int n_here = __popcll(bu) + __popcll(bd);
    if (WRITE_LIST) {
      if (T + n_here > s.cap_tris) return -1;
      int w = T + __popcll(bu & lt_mask) + __popcll(bd & lt_mask);
      if (keepUp) s.tri[w++] = (unsigned short)(cx | (cz << 6));
      if (keepDown) s.tri[w] = (unsigned short)(cx | (cz << 6) | 4096);
    }
    T += n_here;
    cx += qc;
    cz += rc;
    if (cz >= cz_safe) {
      cz -= cz_safe;
      cx += 1;
    }
  }
  if (WRITE_LIST) wave_lds_sync();
  return T;
}

// 64-lane spellings used by the single-call path below
__device__ __forceinline__ void wave_scan_window(const FieldDev& f, const BoxHF& b, const WaveScratch& s,
                                                 int lane, WindowStats& w) {
  grp_scan_window<64>(f, b, s, lane, w);
}
__device__ __forceinline__ bool wave_vertex_pass(const FieldDev& f, const BoxHF& b, const WaveScratch& s,
                                                 int lane, bool allFinite) {
  return grp_vertex_pass<64>(f, b, s, lane, allFinite);
}
template <bool WRITE_LIST>
__device__ __forceinline__ int wave_compact_triangles(const BoxHF& b, const WaveScratch& s, int lane) {
  return grp_compact_triangles<64, WRITE_LIST>(b, s, lane);
}

__device__ __forceinline__ unsigned hash_u64(unsigned long long x) {
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdULL;
  x ^= x >> 33;
  return (unsigned)x;
}

// (g) over the T triangles listed in s.tri.  Returns true when a contact is accepted.
__device__ __forceinline__ bool wave_plane_stage(const FieldDev& f, const BoxHF& b, const WaveScratch& s,
                                                 int lane, int T) {
  const int numX = b.maxX - b.minX + 1;
  // Plane of kept triangle `id` from the LDS tile, and the global sample coordinates of its
  // IsOnHeightfield2 corner vertex (A for ABC, D for DBC).
  auto tri_plane = [&](int id, float pl[4], int& gx, int& gz, bool& up) {
    up = !(id & 4096);
    const int cx = id & 63, cz = (id >> 6) & 63;
    const int e = cz * numX + cx;
    const float xA = (float)(b.minX + cx) * f.sample_w, xB = (float)(b.minX + cx + 1) * f.sample_w;
    const float zA = (float)(b.minZ + cz) * f.sample_d, zC = (float)(b.minZ + cz + 1) * f.sample_d;
    const float hA = s.h[e], hB = s.h[e + 1], hC = s.h[e + numX], hD = s.h[e + numX + 1];
    if (up)
      triangle_plane(xA, hA, zA, xB, hB, zA, xA, hC, zC, true, pl);
    else
      triangle_plane(xB, hD, zC, xB, hB, zA, xA, hC, zC, false, pl);
    gx = b.minX + cx + (up ? 0 : 1);
    gz = b.minZ + cz + (up ? 0 : 1);
  };

  const int slots = (T + 63) >> 6;
  unsigned long long done = 0;  // bit sl: this lane's triangle j = lane + 64*sl is decided

  // Fast path: partner detection on the plane offset d.  Two values with |a-b| < eps (float
  // subtraction, i.e. exact difference < eps up to half an ulp) fall into the same cell of at least
  // one of two grids of cell width 4*eps staggered by 2*eps (FLT_EPSILON = 2^-23, so the scaling by
  // 2^21 is exact).  A triangle with no same-cell neighbour in either grid has no partner.
  if (s.tab_size >= 2 * T && T <= 64 * 64) {
    const unsigned mask = (unsigned)s.tab_size - 1u;
    const unsigned EMPTY = 0xffffffffu, DUP = 0x80000000u;
    unsigned long long partner = 0;  // bit sl: triangle (lane, sl) may have a partner
    for (int grid = 0; grid < 2; ++grid) {
      for (int i = lane; i < s.tab_size; i += 64) s.tab[i] = EMPTY;
      wave_lds_sync();
      for (int sl = 0; sl < slots; ++sl) {
        const int j = lane + 64 * sl;
        if (j < T) {
          float pl[4];
          int gx, gz;
          bool up;
          tri_plane(s.tri[j], pl, gx, gz, up);
          const float d = pl[3];
          if (is_finite(d)) {
            if (fabsf(d) >= 1024.0f) {
              partner |= (1ull << sl);  // outside the exact-quantisation range: be conservative
            } else {
              const long long q = (long long)floor((double)d * 2097152.0 + (grid ? 0.5 : 0.0));
              unsigned tag = hash_u64((unsigned long long)q) & 0x7fffffffu;
              if (tag == 0x7fffffffu) tag = 0x7ffffffeu;
              unsigned slot = (hash_u64((unsigned long long)q * 0x9E3779B97F4A7C15ULL)) & mask;
              for (;;) {
                const unsigned old = atomicCAS(&s.tab[slot], EMPTY, tag);
                if (old == EMPTY) break;
                if ((old & 0x7fffffffu) == tag) {
                  atomicOr(&s.tab[slot], DUP);
                  break;
                }
                slot = (slot + 1) & mask;
              }
            }
          }  // NaN / inf offsets are never epsilon-equal to anything
        }
      }
      wave_lds_sync();
      for (int sl = 0; sl < slots; ++sl) {
        const int j = lane + 64 * sl;
        if (j < T && !((partner >> sl) & 1ull)) {
          float pl[4];
          int gx, gz;
          bool up;
          tri_plane(s.tri[j], pl, gx, gz, up);
          const float d = pl[3];
          if (is_finite(d) && fabsf(d) < 1024.0f) {
            const long long q = (long long)floor((double)d * 2097152.0 + (grid ? 0.5 : 0.0));
            unsigned tag = hash_u64((unsigned long long)q) & 0x7fffffffu;
            if (tag == 0x7fffffffu) tag = 0x7ffffffeu;
            unsigned slot = (hash_u64((unsigned long long)q * 0x9E3779B97F4A7C15ULL)) & mask;
            for (;;) {
              const unsigned cur = s.tab[slot];
              if ((cur & 0x7fffffffu) == tag && cur != EMPTY) {
                if (cur & DUP) partner |= (1ull << sl);
                break;
              }
              if (cur == EMPTY) break;  // cannot happen (it was inserted); defensive
              slot = (slot + 1) & mask;
            }
          }
        }
      }
      wave_lds_sync();
    }
    // singleton groups: own plane, own contacts, own IsOnHeightfield2
    bool hit = false;
    for (int sl = 0; sl < slots; ++sl) {
      const int j = lane + 64 * sl;
      if (j < T && !((partner >> sl) & 1ull)) {
        done |= (1ull << sl);
        float pl[4];
        int gx, gz;
        bool up;
        tri_plane(s.tri[j], pl, gx, gz, up);
        float cpos[4][3];
        const int nc = box_plane_contacts(b, pl[0], pl[1], pl[2], pl[3], 10, cpos);
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (i < nc) hit = hit || is_on_heightfield2(f, gx, gz, cpos[i][0], cpos[i][2], up);
      }
    }
    if (__any(hit)) return true;
  }

  // greedy grouping + contact test over the still-undecided triangles, one plane group per step
  int first_slot = 0;  // lowest slot that may still hold an unassigned triangle
  for (;;) {
    int mine = 0x7fffffff;  // smallest unassigned triangle index of this lane
    for (int sl = first_slot; sl < slots; ++sl) {
      const int j = lane + 64 * sl;
      if (j < T && !((done >> sl) & 1ull)) {
        mine = j;
        break;
      }
    }
    const int k = wave_min_i(mine);
    if (k == 0x7fffffff) break;
    first_slot = k >> 6;
    float base[4];  // base plane (wave-uniform: every lane computes it from broadcast LDS reads)
    {
      int gx, gz;
      bool up;
      tri_plane(s.tri[k], base, gx, gz, up);
    }
    float cpos[4][3];
    const int nc = box_plane_contacts(b, base[0], base[1], base[2], base[3], 10, cpos);
    bool hit = false;
    for (int sl = first_slot; sl < slots; ++sl) {
      const int j = lane + 64 * sl;
      if (j < T && j >= k && !((done >> sl) & 1ull)) {
        float pl[4];
        int gx, gz;
        bool up;
        tri_plane(s.tri[j], pl, gx, gz, up);
        const bool same = (j == k) || planes_eps_equal(base, pl);
        if (same) {
          done |= (1ull << sl);
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (i < nc) hit = hit || is_on_heightfield2(f, gx, gz, cpos[i][0], cpos[i][2], up);
        }
      }
    }
    if (__any(hit)) return true;
  }
  return false;
}

// (g), common case.  Every contact dCollideBoxPlane produces is a corner of the box (deepest corner,
// its two neighbours along the smallest-projection sides, and the fourth corner of that face;
// ode/ode/src/box.cpp:789-861), and a contact is only accepted by a group triangle whose CELL contains
// its (x, z) (IsOnHeightfield2, heightfield.cpp:264-321).  So only kept triangles in the <= 8 cells
// under the box corners (widened by a margin that dwarfs any float rounding of the corner positions)
// can ever accept a contact: the "candidates".  A candidate whose plane has no epsilon-equal partner
// among ALL kept triangles is a group of its own in the greedy grouping (:1511-1556), so its base
// plane is its own plane and it can be decided alone.  Returns 0 / 1 when that settles the check, 2
// when some candidate has a partner (the caller then runs the exact sequential grouping).
// Packed kept-triangle id: window-local cell (cx, cz) and orientation; cx, cz < 64 (host-checked).
__device__ __forceinline__ int tri_pack(int cx, int cz, bool down) { return cx | (cz << 6) | (down ? 4096 : 0); }

template <int G>
struct CandCap { static constexpr int value = (G == 64) ? 64 : 32; };

template <int G, bool GLOBAL_H = false>
__device__ __forceinline__ int grp_plane_stage_corners(const FieldDev& f, const BoxHF& b,
                                                       const WaveScratch& s, int lane, int T, bool fast = false
#ifdef ARTP_STAGE_TIMING
                                                       , unsigned long long* t_acc = nullptr
#endif
) {
#ifdef ARTP_STAGE_TIMING
  long long t_prev = clock64();
#define ARTP_TC_MARK(slot) do { const long long n_ = clock64(); if (t_acc) t_acc[slot] += (unsigned long long)(n_ - t_prev); t_prev = n_; } while (0)
#else
#define ARTP_TC_MARK(slot) do { } while (0)
#endif
  constexpr int CAP = CandCap<G>::value;
  const int gl = grp_lane<G>(lane);
  const int numX = b.maxX - b.minX + 1;
  const int numZ = b.maxZ - b.minZ + 1;
  const int cellsX = numX - 1, cellsZ = numZ - 1;
  const float minO2 = b.aabb[2];
  // Corner positions are a handful of float ops on coordinates of a few tens of metres (rounding
  // ~1e-5 m at most); the contact positions of dCollideBoxPlane are the same corners computed in another
  // order.  A 0.1 mm margin around each corner covers both.
  const float margin = 1.0e-4f;
  float4* cand_pl = reinterpret_cast<float4*>(s.cand);         // [CAP] candidate planes (compacted)
  float4* cand_raw = cand_pl + CAP;                             // [CAP] raw cross x, z, tolerance, unused
  int* cand_id = reinterpret_cast<int*>(cand_raw + CAP);        // [CAP] packed triangle ids
  // 16 base slots = 8 corners x 2 orientations for the cell under the corner; a corner within the margin
  // of a cell border also nominates the neighbouring cell(s): offsets (1,0), (0,1), (1,1).
  // G = 64 walks the four offsets side by side; G = 16 runs an offset round only when some corner needs it.
  // fast mode (T unknown, no kept-triangle list): decide from the candidates alone when the map's partner
  // table rules out a partner for every one of them; 2 = "needs the list" otherwise.
  int ncand = 0;
  bool maybe_partner = false;
  const bool window_covered = f.partner_flags != nullptr && cellsX <= f.partner_R && cellsZ <= f.partner_R;
  const int base_slot = gl & 15;
  const int corner = base_slot >> 1;
  const bool c_up = !(base_slot & 1);
  const float s0 = (corner & 1) ? 0.5f : -0.5f, s1 = (corner & 2) ? 0.5f : -0.5f, s2 = (corner & 4) ? 0.5f : -0.5f;
  const float px = b.pos[0] + s0 * b.side[0] * b.R[0] + s1 * b.side[1] * b.R[1] + s2 * b.side[2] * b.R[2];
  const float py = b.pos[1] + s0 * b.side[0] * b.R[3] + s1 * b.side[1] * b.R[4] + s2 * b.side[2] * b.R[5];
  const float pz = b.pos[2] + s0 * b.side[0] * b.R[6] + s1 * b.side[1] * b.R[7] + s2 * b.side[2] * b.R[8];
  // A contact of dCollideBoxPlane sits on a box corner whose depth below the plane is >= 0.  The plane of a
  // triangle, extended over its whole cell (+ the margin), stays below `top + spread * reach`: a corner
  // higher than that (1 mm slack for the roundings of depth and of this corner) cannot be a contact of
  // that triangle's plane -- nor of a partner's epsilon-equal plane -- so the pair is no candidate.
  const float reach = 2.0f * margin * fmaxf(f.inv_w, f.inv_d);
  const int cxa = (int)floorf((px - margin) * f.inv_w), cxb = (int)floorf((px + margin) * f.inv_w);
  const int cza = (int)floorf((pz - margin) * f.inv_d), czb = (int)floorf((pz + margin) * f.inv_d);
  for (int r = 0; r < 4; ++r) {
    const int off = (G == 64) ? (gl >> 4) : r;
    const int dx = off & 1, dz = off >> 1;
    const bool wanted = (cxa + dx <= cxb) && (cza + dz <= czb);
    if (G != 64 && r > 0 && !grp_any<G>(wanted, lane)) continue;
    const int cx = cxa + dx - b.minX, cz = cza + dz - b.minZ;  // window-local cell
    bool is_cand = wanted && cx >= 0 && cz >= 0 && cx < cellsX && cz < cellsZ;
    float cpl[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    float craw[3] = {0.0f, 0.0f, 0.0f};
    if (is_cand) {
      // GLOBAL_H: no LDS tile was staged (fast mode only), read the four samples of the cell from the map
      float hA, hB, hC, hD;
      if (GLOBAL_H) {
        const unsigned at = (unsigned)((b.minX + cx) + (b.minZ + cz) * f.nW);
        hA = gather32(f.data, at);
        hB = gather32(f.data, at + 1u);
        hC = gather32(f.data, at + (unsigned)f.nW);
        hD = gather32(f.data, at + (unsigned)f.nW + 1u);
      } else {
        const float* hp = s.h + cz * numX + cx;
        hA = hp[0], hB = hp[1], hC = hp[numX], hD = hp[numX + 1];
      }
      const bool fA = is_finite(hA), fB = is_finite(hB), fC = is_finite(hC), fD = is_finite(hD);
      const bool kA = fA && hA > minO2, kB = fB && hB > minO2, kC = fC && hC > minO2, kD = fD && hD > minO2;
      bool kept = c_up ? ((kA || kB || kC) && (fA && fB && fC)) : ((kB || kC || kD) && (fB && fC && fD));
      if (kept) {
        // plane height at the cell's fourth corner: hB + hC - hA (ABC), hB + hC - hD (DBC)
        const float h0 = c_up ? hA : hD;
        const float h4 = (hB + hC) - h0;
        const float top = fmaxf(fmaxf(h0, h4), fmaxf(hB, hC));
        const float spread = fabsf(hB - h0) + fabsf(hC - h0);
        kept = !(py > top + spread * reach + 1.0e-3f);
      }
      is_cand = kept;
      if (kept && fast)
        maybe_partner = maybe_partner || !window_covered ||
                        ((gather32(f.partner_flags, (unsigned)((b.minX + cx) + (b.minZ + cz) * f.nW)) >> (c_up ? 0 : 1)) & 1);
      if (kept) {
        const float xA = (float)(b.minX + cx) * f.sample_w, xB = (float)(b.minX + cx + 1) * f.sample_w;
        const float zA = (float)(b.minZ + cz) * f.sample_d, zC = (float)(b.minZ + cz + 1) * f.sample_d;
        if (c_up)
          triangle_plane(xA, hA, zA, xB, hB, zA, xA, hC, zC, true, cpl, craw);
        else
          triangle_plane(xB, hD, zC, xB, hB, zA, xA, hC, zC, false, cpl, craw);
      }
    }
    const unsigned long long m = grp_ballot<G>(is_cand, lane);
    const int n_here = __popcll(m);
    if (ncand + n_here > CAP) return 2;  // more candidates than the scratch holds: exact grouping stage
    if (is_cand) {
      const int w = ncand + __popcll(m & grp_lt_mask<G>(lane));
      cand_pl[w] = make_float4(cpl[0], cpl[1], cpl[2], cpl[3]);
      cand_id[w] = tri_pack(cx, cz, !c_up);  // same id as the kept-triangle list
      // pre-filter tolerance; near-vertical candidate planes (|ny| < 0.05) are never filtered
      const float len = sqrtf(craw[0] * craw[0] + craw[1] * craw[1] + craw[2] * craw[2]);
      const float tol = (fabsf(cpl[1]) >= 0.05f) ? f.partner_tol * len : INFINITY;
      cand_raw[w] = make_float4(craw[0], craw[2], tol, 0.0f);
    }
    ncand += n_here;
    if (G == 64) break;  // the four offsets were handled side by side
  }
  ARTP_TC_MARK(5);
  if (ncand == 0) return 0;  // no kept triangle under any box corner: nothing can accept a contact
  wave_lds_sync();
  if (fast || GLOBAL_H) {
    if (!fast || grp_any<G>(maybe_partner, lane)) return 2;
  } else {
  // does any kept triangle have a plane epsilon-equal to a candidate's (other than itself)?
  // (the same triangle may be nominated by two corners: duplicates are harmless)
  bool partner = false;
  for (int j = gl; j < T; j += G) {
    const int id = s.tri[j];
    const bool up = !(id & 4096);
    const int tx = id & 63, tz = (id >> 6) & 63;
    const int e = tz * numX + tx;
    const float xA = (float)(b.minX + tx) * f.sample_w, xB = (float)(b.minX + tx + 1) * f.sample_w;
    const float zA = (float)(b.minZ + tz) * f.sample_d, zC = (float)(b.minZ + tz + 1) * f.sample_d;
    const float hA = s.h[e], hB = s.h[e + 1], hC = s.h[e + numX], hD = s.h[e + numX + 1];
    // cheap necessary condition first: the raw cross product (no sqrt / divide) must be close to a
    // candidate's; only then the normalised plane is computed and compared exactly
    float raw[3];
    if (up)
      triangle_cross(xA, hA, zA, xB, hB, zA, xA, hC, zC, true, raw);
    else
      triangle_cross(xB, hD, zC, xB, hB, zA, xA, hC, zC, false, raw);
    bool close = false;
    for (int q = 0; q < ncand; ++q) {
      const float4 cr = cand_raw[q];  // broadcast read
      close = close || (!(fabsf(raw[0] - cr.x) > cr.z) && !(fabsf(raw[2] - cr.y) > cr.z));
    }
    if (close) {
      float pl[4];
      if (up)
        triangle_plane(xA, hA, zA, xB, hB, zA, xA, hC, zC, true, pl);
      else
        triangle_plane(xB, hD, zC, xB, hB, zA, xA, hC, zC, false, pl);
      for (int q = 0; q < ncand; ++q) {
        const float4 cp = cand_pl[q];
        if (fabsf(pl[3] - cp.w) < ARTP_EPS) {
          if (cand_id[q] != id && fabsf(pl[1] - cp.y) < ARTP_EPS && fabsf(pl[0] - cp.x) < ARTP_EPS &&
              fabsf(pl[2] - cp.z) < ARTP_EPS)
            partner = true;
        }
      }
    }
  }
  ARTP_TC_MARK(6);
  if (grp_any<G>(partner, lane)) return 2;
  }
  // all candidates are singleton groups: own plane, own contacts, own cell
  bool hit = false;
  for (int q = gl; q < ncand; q += G) {
    const float4 cp = cand_pl[q];
    const int id = cand_id[q];
    const bool up = !(id & 4096);
    const int tx = id & 63, tz = (id >> 6) & 63;
    const int gx = b.minX + tx + (up ? 0 : 1), gz = b.minZ + tz + (up ? 0 : 1);
    float cpos[4][3];
    const int nc = box_plane_contacts(b, cp.x, cp.y, cp.z, cp.w, 10, cpos);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (i < nc) hit = hit || is_on_heightfield2(f, gx, gz, cpos[i][0], cpos[i][2], up);
  }
  ARTP_TC_MARK(7);
  return grp_any<G>(hit, lane) ? 1 : 0;
}

// The corner stage of the streaming kernels: no LDS tile, no kept-triangle list, no candidate arrays.  The lane that
// nominates a (corner, triangle) pair reads the cell's four samples from the map, forms the plane and tests its
// contacts right there; when the map's partner table cannot rule out a partner for some kept candidate the box
// needs the list (2).  Same candidates, same planes and same contact tests as grp_plane_stage_corners in fast mode
// (a triangle nominated by two corners is tested twice: harmless).
template <int G>
__device__ __forceinline__ int grp_corner_stage_direct(const FieldDev& f, const BoxHF& b, int lane) {
  const int gl = grp_lane<G>(lane);
  const int cellsX = b.maxX - b.minX, cellsZ = b.maxZ - b.minZ;
  const float minO2 = b.aabb[2];
  const float margin = 1.0e-4f;  // see grp_plane_stage_corners
  const bool window_covered = f.partner_flags != nullptr && cellsX <= f.partner_R && cellsZ <= f.partner_R;
  const int base_slot = gl & 15;
  const int corner = base_slot >> 1;
  const bool c_up = !(base_slot & 1);
  const float s0 = (corner & 1) ? 0.5f : -0.5f, s1 = (corner & 2) ? 0.5f : -0.5f, s2 = (corner & 4) ? 0.5f : -0.5f;
  const float px = b.pos[0] + s0 * b.side[0] * b.R[0] + s1 * b.side[1] * b.R[1] + s2 * b.side[2] * b.R[2];
  const float py = b.pos[1] + s0 * b.side[0] * b.R[3] + s1 * b.side[1] * b.R[4] + s2 * b.side[2] * b.R[5];
  const float pz = b.pos[2] + s0 * b.side[0] * b.R[6] + s1 * b.side[1] * b.R[7] + s2 * b.side[2] * b.R[8];
  const float reach = 2.0f * margin * fmaxf(f.inv_w, f.inv_d);
  const int cxa = (int)floorf((px - margin) * f.inv_w), cxb = (int)floorf((px + margin) * f.inv_w);
  const int cza = (int)floorf((pz - margin) * f.inv_d), czb = (int)floorf((pz + margin) * f.inv_d);
  bool maybe_partner = false, hit = false;
  for (int r = 0; r < 4; ++r) {
    const int off = (G == 64) ? (gl >> 4) : r;
    const int dx = off & 1, dz = off >> 1;
    const bool wanted = (cxa + dx <= cxb) && (cza + dz <= czb);
    if (G != 64 && r > 0 && !grp_any<G>(wanted, lane)) continue;
    const int cx = cxa + dx - b.minX, cz = cza + dz - b.minZ;  // window-local cell
    if (wanted && cx >= 0 && cz >= 0 && cx < cellsX && cz < cellsZ) {
      const unsigned at = (unsigned)((b.minX + cx) + (b.minZ + cz) * f.nW);
      const float hA = gather32(f.data, at), hB = gather32(f.data, at + 1u);
      const float hC = gather32(f.data, at + (unsigned)f.nW), hD = gather32(f.data, at + (unsigned)f.nW + 1u);
      const bool fA = is_finite(hA), fB = is_finite(hB), fC = is_finite(hC), fD = is_finite(hD);
      const bool kA = fA && hA > minO2, kB = fB && hB > minO2, kC = fC && hC > minO2, kD = fD && hD > minO2;
      bool kept = c_up ? ((kA || kB || kC) && (fA && fB && fC)) : ((kB || kC || kD) && (fB && fC && fD));
      if (kept) {
        const float h0 = c_up ? hA : hD;
        const float h4 = (hB + hC) - h0;
        const float top = fmaxf(fmaxf(h0, h4), fmaxf(hB, hC));
        const float spread = fabsf(hB - h0) + fabsf(hC - h0);
        kept = !(py > top + spread * reach + 1.0e-3f);
      }
      if (kept) {
        maybe_partner = maybe_partner || !window_covered || ((gather32(f.partner_flags, at) >> (c_up ? 0 : 1)) & 1);
        const float xA = (float)(b.minX + cx) * f.sample_w, xB = (float)(b.minX + cx + 1) * f.sample_w;
        const float zA = (float)(b.minZ + cz) * f.sample_d, zC = (float)(b.minZ + cz + 1) * f.sample_d;
        float cpl[4];
        if (c_up)
          triangle_plane(xA, hA, zA, xB, hB, zA, xA, hC, zC, true, cpl);
        else
          triangle_plane(xB, hD, zC, xB, hB, zA, xA, hC, zC, false, cpl);
        const int gx = b.minX + cx + (c_up ? 0 : 1), gz = b.minZ + cz + (c_up ? 0 : 1);
        float cpos[4][3];
        const int nc = box_plane_contacts(b, cpl[0], cpl[1], cpl[2], cpl[3], 10, cpos);
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (i < nc) hit = hit || is_on_heightfield2(f, gx, gz, cpos[i][0], cpos[i][2], c_up);
      }
    }
    if (G == 64) break;  // the four offsets were handled side by side
  }
  if (grp_any<G>(maybe_partner, lane)) return 2;
  return grp_any<G>(hit, lane) ? 1 : 0;
}

__device__ __forceinline__ int wave_plane_stage_corners(const FieldDev& f, const BoxHF& b,
                                                        const WaveScratch& s, int lane, int T) {
  return grp_plane_stage_corners<64>(f, b, s, lane, T);
}

// The whole zone test in one call (used at the HeightMapBoxChecker boundary, artp_check_boxes).
// Returns 0/1 like dCollide(box, field, 1, ...) != 0.  All 64 lanes must call it with identical
// arguments; the result and *exit_code are wave-uniform.  Returns -1 when the window does not fit the
// LDS scratch (the host sizes the scratch from the box diagonal).
__device__ __forceinline__ int wave_check_box(const FieldDev& f, const BoxHF& b, const WaveScratch& s,
                                              int lane, int* exit_code) {
  if (!b.on_field) {
    *exit_code = EXIT_AABB_OFF;
    return 0;
  }
  const int total = (b.maxX - b.minX + 1) * (b.maxZ - b.minZ + 1);
  if (total > s.cap_verts) {
    *exit_code = -1;
    return -1;
  }
  WindowStats w;
  wave_scan_window(f, b, s, lane, w);
  int result;
  if (decide_exits(b, w, result, *exit_code)) return result;
  if (wave_vertex_pass(f, b, s, lane, w.allFinite)) {
    *exit_code = EXIT_VERTEX;
    return 1;
  }
  const int T = wave_compact_triangles<true>(b, s, lane);
  if (T < 0) {
    *exit_code = -1;
    return -1;
  }
  if (T > 0) {
    int r = s.cand ? wave_plane_stage_corners(f, b, s, lane, T) : 2;
    if (r == 2) r = wave_plane_stage(f, b, s, lane, T) ? 1 : 0;
    if (r) {
      *exit_code = EXIT_PLANE;
      return 1;
    }
  }
  *exit_code = EXIT_NONE;
  return 0;
}

}  // namespace artp
