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

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    const float o = __shfl_xor(v, m, 64);
    v = (o > v) ? o : v;
  }
  return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    const float o = __shfl_xor(v, m, 64);
    v = (o < v) ? o : v;
  }
  return v;
}
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    const int o = __shfl_xor(v, m, 64);
    v = (o > v) ? o : v;
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
  float* cand;          // 64 x float4: planes of the corner-candidate triangles (may be null)
  float* h;             // heights tile, z-major rows of numX; capacity cap_verts
  unsigned short* tri;  // kept-triangle list; id = 2*(cx*(numZ-1)+cz) + (down ? 1 : 0); cap_tris
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
__device__ __forceinline__ void wave_scan_window(const FieldDev& f, const BoxHF& b, const WaveScratch& s,
                                                 int lane, WindowStats& w) {
  const int numX = b.maxX - b.minX + 1;
  const int numZ = b.maxZ - b.minZ + 1;
  const int total = numX * numZ;
  float lmax = -INFINITY, lmin = INFINITY;
  int lnonfinite = 0;
  int llast_nan = -1;  // ODE scan index (xl*numZ + zl) of this lane's last NaN
  const int qz = 64 / numX, rx = 64 - qz * numX;  // advance of (xl, zl) per 64 elements
  {
    int xl = lane % numX, zl = lane / numX;
    const float* base = f.data + b.minX + (size_t)b.minZ * f.nW;
    for (int e = lane; e < total; e += 64) {
      const float h = base[xl + zl * f.nW];
      s.h[e] = h;
      lmax = (h > lmax) ? h : lmax;  // NaN never wins here; handled below
      if (is_finite(h)) {
        lmin = (lmin > h) ? h : lmin;
      } else {
        lnonfinite = 1;
        if (is_nan(h)) {
          const int si = xl * numZ + zl;
          llast_nan = si > llast_nan ? si : llast_nan;
        }
      }
      xl += rx;
      zl += qz;
      if (xl >= numX) {
        xl -= numX;
        zl += 1;
      }
    }
  }
  wave_lds_sync();
  w.maxY = wave_max(lmax);
  w.minY = wave_min(lmin);
  w.allFinite = !__any(lnonfinite);
  if (f.has_nan && !w.allFinite) {
    const int last_nan = wave_max_i(llast_nan);
    if (last_nan >= 0) {
      // running dMAX: the maximum restarts after every NaN, and a trailing NaN survives.
      if (last_nan == total - 1) {
        w.maxY = __uint_as_float(0x7fc00000u);
      } else {
        float m2 = -INFINITY;
        int xl = lane % numX, zl = lane / numX;
        for (int e = lane; e < total; e += 64) {
          const float h = s.h[e];
          if (xl * numZ + zl > last_nan) m2 = (h > m2) ? h : m2;
          xl += rx;
          zl += qz;
          if (xl >= numX) {
            xl -= numX;
            zl += 1;
          }
        }
        w.maxY = wave_max(m2);
      }
    }
  }
}

// (f).  Heights must be staged in s.h.
__device__ __forceinline__ bool wave_vertex_pass(const FieldDev& f, const BoxHF& b, const WaveScratch& s,
                                                 int lane, bool allFinite) {
  const int numX = b.maxX - b.minX + 1;
  const int numZ = b.maxZ - b.minZ + 1;
  const int total = numX * numZ;
  const int cellsX = numX - 1, cellsZ = numZ - 1;
  const float minO2 = b.aabb[2];
  bool hit = false;
  const int qz = 64 / numX, rx = 64 - qz * numX;
  int xl = lane % numX, zl = lane / numX;
  for (int e0 = 0; e0 < total; e0 += 64) {
    const int e = e0 + lane;
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
    if (__any(hit)) return true;
  }
  return false;
}

// Kept triangles of the window in the reference's buffer order (x_local outer, z_local inner, ABC
// before DBC; :1306-1441).  With WRITE_LIST the ids go to s.tri; returns T, or -1 on list overflow.
template <bool WRITE_LIST>
__device__ __forceinline__ int wave_compact_triangles(const BoxHF& b, const WaveScratch& s, int lane) {
  const int numX = b.maxX - b.minX + 1;
  const int numZ = b.maxZ - b.minZ + 1;
  const int cellsX = numX - 1, cellsZ = numZ - 1;
  const int ncells = cellsX * cellsZ;
  const float minO2 = b.aabb[2];
  int T = 0;
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  for (int c0 = 0; c0 < ncells; c0 += 64) {
    const int c = c0 + lane;
    bool keepUp = false, keepDown = false;
    if (c < ncells) {
      const int cx = c / cellsZ, cz = c - cx * cellsZ;
      const int e = cz * numX + cx;
      const float hA = s.h[e], hB = s.h[e + 1], hC = s.h[e + numX], hD = s.h[e + numX + 1];
      const bool fA = is_finite(hA), fB = is_finite(hB), fC = is_finite(hC), fD = is_finite(hD);
      const bool cA = fA && hA > minO2, cB = fB && hB > minO2, cC = fC && hC > minO2,
                 cD = fD && hD > minO2;
      keepUp = (cA || cB || cC) && (fA && fB && fC);
      keepDown = (cB || cC || cD) && (fB && fC && fD);
    }
    const unsigned long long bu = __ballot(keepUp), bd = __ballot(keepDown);
    const int n_here = __popcll(bu) + __popcll(bd);
    if (WRITE_LIST) {
      if (T + n_here > s.cap_tris) return -1;
      int w = T + __popcll(bu & lt_mask) + __popcll(bd & lt_mask);
      if (keepUp) s.tri[w++] = (unsigned short)(2 * c);
      if (keepDown) s.tri[w] = (unsigned short)(2 * c + 1);
    }
    T += n_here;
  }
  if (WRITE_LIST) wave_lds_sync();
  return T;
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
  const int numZ = b.maxZ - b.minZ + 1;
  const int cellsZ = numZ - 1;
  // Plane of kept triangle `id` from the LDS tile, and the global sample coordinates of its
  // IsOnHeightfield2 corner vertex (A for ABC, D for DBC).
  auto tri_plane = [&](int id, float pl[4], int& gx, int& gz, bool& up) {
    const int c = id >> 1;
    up = !(id & 1);
    const int cx = c / cellsZ, cz = c - cx * cellsZ;
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
__device__ __forceinline__ int wave_plane_stage_corners(const FieldDev& f, const BoxHF& b,
                                                        const WaveScratch& s, int lane, int T) {
  const int numX = b.maxX - b.minX + 1;
  const int numZ = b.maxZ - b.minZ + 1;
  const int cellsX = numX - 1, cellsZ = numZ - 1;
  const float minO2 = b.aabb[2];
  // candidate (lane): corner = lane>>3, cell offset (dx, dz) = (lane&1, (lane>>1)&1), up/down = lane>>2 &1
  const int corner = lane >> 3;
  const float s0 = (corner & 1) ? 0.5f : -0.5f, s1 = (corner & 2) ? 0.5f : -0.5f, s2 = (corner & 4) ? 0.5f : -0.5f;
  const float px = b.pos[0] + s0 * b.side[0] * b.R[0] + s1 * b.side[1] * b.R[1] + s2 * b.side[2] * b.R[2];
  const float pz = b.pos[2] + s0 * b.side[0] * b.R[6] + s1 * b.side[1] * b.R[7] + s2 * b.side[2] * b.R[8];
  const float margin = 1.0e-3f;
  const int cxa = (int)floorf((px - margin) * f.inv_w), cxb = (int)floorf((px + margin) * f.inv_w);
  const int cza = (int)floorf((pz - margin) * f.inv_d), czb = (int)floorf((pz + margin) * f.inv_d);
  const int dx = lane & 1, dz = (lane >> 1) & 1;
  const bool c_up = !((lane >> 2) & 1);
  const int cx = cxa + dx - b.minX, cz = cza + dz - b.minZ;  // window-local cell
  bool is_cand = (cxa + dx <= cxb) && (cza + dz <= czb) && cx >= 0 && cz >= 0 && cx < cellsX && cz < cellsZ;
  float cpl[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  int cgx = 0, cgz = 0;
  if (is_cand) {
    const int e = cz * numX + cx;
    const float hA = s.h[e], hB = s.h[e + 1], hC = s.h[e + numX], hD = s.h[e + numX + 1];
    const bool fA = is_finite(hA), fB = is_finite(hB), fC = is_finite(hC), fD = is_finite(hD);
    const bool kA = fA && hA > minO2, kB = fB && hB > minO2, kC = fC && hC > minO2, kD = fD && hD > minO2;
    const bool kept = c_up ? ((kA || kB || kC) && (fA && fB && fC)) : ((kB || kC || kD) && (fB && fC && fD));
    is_cand = kept;
    if (kept) {
      const float xA = (float)(b.minX + cx) * f.sample_w, xB = (float)(b.minX + cx + 1) * f.sample_w;
      const float zA = (float)(b.minZ + cz) * f.sample_d, zC = (float)(b.minZ + cz + 1) * f.sample_d;
      if (c_up)
        triangle_plane(xA, hA, zA, xB, hB, zA, xA, hC, zC, true, cpl);
      else
        triangle_plane(xB, hD, zC, xB, hB, zA, xA, hC, zC, false, cpl);
      cgx = b.minX + cx + (c_up ? 0 : 1);
      cgz = b.minZ + cz + (c_up ? 0 : 1);
    }
  }
  const unsigned long long cand_mask = __ballot(is_cand);
  if (cand_mask == 0ull) return 0;  // no kept triangle under any box corner: nothing can accept
  const int my_id = 2 * (cx * cellsZ + cz) + (c_up ? 0 : 1);
  // compact candidate planes into LDS (ids alongside, so a triangle does not partner with itself)
  const int ncand = __popcll(cand_mask);
  if (is_cand) {
    const int w = __popcll(cand_mask & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))));
    reinterpret_cast<float4*>(s.cand)[w] = make_float4(cpl[0], cpl[1], cpl[2], cpl[3]);
    reinterpret_cast<int*>(s.cand)[256 + w] = my_id;
  }
  wave_lds_sync();
  // does any kept triangle have a plane epsilon-equal to a candidate's (other than itself)?
  bool partner = false;
  const int slots = (T + 63) >> 6;
  for (int sl = 0; sl < slots; ++sl) {
    const int j = lane + 64 * sl;
    if (j < T) {
      const int id = s.tri[j];
      const int c = id >> 1;
      const bool up = !(id & 1);
      const int tx = c / cellsZ, tz = c - tx * cellsZ;
      const int e = tz * numX + tx;
      const float xA = (float)(b.minX + tx) * f.sample_w, xB = (float)(b.minX + tx + 1) * f.sample_w;
      const float zA = (float)(b.minZ + tz) * f.sample_d, zC = (float)(b.minZ + tz + 1) * f.sample_d;
      const float hA = s.h[e], hB = s.h[e + 1], hC = s.h[e + numX], hD = s.h[e + numX + 1];
      float pl[4];
      if (up)
        triangle_plane(xA, hA, zA, xB, hB, zA, xA, hC, zC, true, pl);
      else
        triangle_plane(xB, hD, zC, xB, hB, zA, xA, hC, zC, false, pl);
      for (int q = 0; q < ncand; ++q) {
        const float4 cp = reinterpret_cast<const float4*>(s.cand)[q];  // broadcast read
        if (fabsf(pl[3] - cp.w) < ARTP_EPS) {
          const int cid = reinterpret_cast<const int*>(s.cand)[256 + q];
          if (cid != id && fabsf(pl[1] - cp.y) < ARTP_EPS && fabsf(pl[0] - cp.x) < ARTP_EPS &&
              fabsf(pl[2] - cp.z) < ARTP_EPS)
            partner = true;
        }
      }
    }
  }
  if (__any(partner)) return 2;
  // all candidates are singleton groups: own plane, own contacts, own cell
  bool hit = false;
  if (is_cand) {
    float cpos[4][3];
    const int nc = box_plane_contacts(b, cpl[0], cpl[1], cpl[2], cpl[3], 10, cpos);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (i < nc) hit = hit || is_on_heightfield2(f, cgx, cgz, cpos[i][0], cpos[i][2], c_up);
  }
  return __any(hit) ? 1 : 0;
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
