// box_check.h -- wave-cooperative box-vs-heightfield zone test for gfx950 (one 64-lane wavefront
// decides one box).
//
// Restates dxHeightfield::dCollideHeightfieldZone (ode/ode/src/heightfield.cpp:973-1789, with the
// art_planner patches at :989,1020-1024,1052-1064,1139,1329-1378) as a data-parallel algorithm with
// IDENTICAL results for "at most one contact requested" (HeightMapBoxChecker::checkCollision,
// art_planner/src/validity_checker/height_map_box_checker.cpp:67):
//
//  (a) the index window is streamed from HBM/L2 row-major-coalesced (x is the fast axis of the ODE
//      sample layout) into an LDS tile, one element per lane per step, while max / min-of-finite /
//      all-finite are reduced across the wave;
//  (b)(c)(d)(e) the early-outs are decided wave-uniformly from the reduced values;
//  (f) "terrain vertex inside the box": every window vertex that is colliding (finite, above the box
//      bottom) and belongs to a triangle whose three vertices are finite is tested lane-parallel,
//      any hit ends the check (order independent because the reference returns 1 at the first hit);
//  (g) kept triangles are compacted IN THE REFERENCE'S BUFFER ORDER into an LDS list; the greedy
//      epsilon-equality plane grouping (:1511-1556) runs one group per step with the members
//      compared lane-parallel, and each group's box-plane contacts are tested against its member
//      triangles lane-parallel (IsOnHeightfield2).  The bubble sort of planes (:933-955,1559) only
//      orders contact generation and cannot change "is there a contact", so it is not needed;
//  (h) the final vertex pass (:1651-1719) can only re-test vertices already rejected in (f) for box
//      geoms (same pure function, same arguments), so it never produces a contact and is omitted.
//
// The running `maxY = dMAX(maxY, h)` NaN quirk (:1019) is honoured through the scan index of the last
// NaN sample (ODE scans x-outer, z-inner).
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

// Per-wave LDS scratch: heights tile (z-major rows of numX) + kept-triangle list.
struct WaveScratch {
  float* h;             // capacity cap_verts
  unsigned short* tri;  // capacity cap_tris; id = 2*(cx*(numZ-1)+cz) + (down ? 1 : 0)
  int cap_verts;
  int cap_tris;
};

// Returns 0/1 like dCollide(box, field, 1, ...) != 0.  All 64 lanes must call it with identical
// arguments (wave-uniform); the result and *exit_code are wave-uniform.  Returns -1 when the window
// does not fit the LDS scratch (caller must size the scratch from the box diagonal; see host code).
__device__ __forceinline__ int wave_check_box(const FieldDev& f, const BoxHF& b, const WaveScratch& s,
                                           int lane, int* exit_code) {
  if (!b.on_field) {
    *exit_code = EXIT_AABB_OFF;
    return 0;
  }
  const int numX = b.maxX - b.minX + 1;
  const int numZ = b.maxZ - b.minZ + 1;
  const int total = numX * numZ;
  if (total > s.cap_verts) {
    *exit_code = -1;
    return -1;
  }
  const float minO2 = b.aabb[2];
  const float maxO2 = b.aabb[3];

  // ---- (a) stream the window into LDS, reduce max / min-of-finite / all-finite -----------------
  float lmax = -INFINITY, lmin = INFINITY;
  int lnonfinite = 0;
  int llast_nan = -1;  // ODE scan index (xl*numZ + zl) of this lane's last NaN
  {
    const int qz = 64 / numX, rx = 64 - qz * numX;  // advance of (xl, zl) per 64 elements
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
  float maxY = wave_max(lmax);
  const float minY = wave_min(lmin);
  const bool allFinite = !__any(lnonfinite);
  if (f.has_nan && !allFinite) {
    const int last_nan = wave_max_i(llast_nan);
    if (last_nan >= 0) {
      // running dMAX: the maximum restarts after every NaN, and a trailing NaN survives.
      if (last_nan == total - 1) {
        maxY = __uint_as_float(0x7fc00000u);
      } else {
        float m2 = -INFINITY;
        int xl = lane % numX, zl = lane / numX;
        const int qz = 64 / numX, rx = 64 - qz * numX;
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
        maxY = wave_max(m2);
      }
    }
  }

  // ---- (b)(c)(d)(e) ---------------------------------------------------------------------------
  if (minO2 - maxY > -ARTP_EPS) {
    *exit_code = EXIT_ABOVE;
    return 0;
  }
  if (minY - maxO2 > -ARTP_EPS) {
    *exit_code = EXIT_UNDER;
    return 0;
  }
  if (allFinite && minY - minO2 > -ARTP_EPS && maxO2 - maxY > -ARTP_EPS) {
    *exit_code = EXIT_SPANS;
    return 1;
  }
  if (allFinite && (maxY - minY < ARTP_EPS)) {
    float cpos[4][3];
    *exit_code = EXIT_FLAT_PLANE;
    return box_plane_contacts(b, 0.0f, 1.0f, 0.0f, minY, 1, cpos) ? 1 : 0;
  }

  // ---- (f) vertex-in-box ----------------------------------------------------------------------
  const int cellsX = numX - 1, cellsZ = numZ - 1;
  {
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
      if (__any(hit)) {
        *exit_code = EXIT_VERTEX;
        return 1;
      }
    }
  }

  // ---- (g) kept triangles, in the reference's buffer order --------------------------------------
  const int ncells = cellsX * cellsZ;
  int T = 0;
  {
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
      const int before = __popcll(bu & lt_mask) + __popcll(bd & lt_mask);
      const int n_here = __popcll(bu) + __popcll(bd);
      if (T + n_here > s.cap_tris) {
        *exit_code = -1;
        return -1;
      }
      int w = T + before;
      if (keepUp) s.tri[w++] = (unsigned short)(2 * c);
      if (keepDown) s.tri[w] = (unsigned short)(2 * c + 1);
      T += n_here;
    }
  }
  wave_lds_sync();
  if (T == 0) {
    *exit_code = EXIT_NONE;
    return 0;
  }

  // greedy grouping + contact test, one plane group per step
  {
    unsigned long long done = 0;  // bit sl: this lane's triangle j = lane + 64*sl is assigned
    const int slots = (T + 63) >> 6;
    int first_slot = 0;           // lowest slot that may still hold an unassigned triangle
    for (;;) {
      // smallest unassigned triangle index over the wave
      int mine = 0x7fffffff;
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
      // base plane (wave-uniform, every lane computes it from broadcast LDS reads)
      float base[4];
      {
        const int id = s.tri[k];
        const int c = id >> 1;
        const bool up = !(id & 1);
        const int cx = c / cellsZ, cz = c - cx * cellsZ;
        const int e = cz * numX + cx;
        const float xA = (float)(b.minX + cx) * f.sample_w, xB = (float)(b.minX + cx + 1) * f.sample_w;
        const float zA = (float)(b.minZ + cz) * f.sample_d, zC = (float)(b.minZ + cz + 1) * f.sample_d;
        const float hA = s.h[e], hB = s.h[e + 1], hC = s.h[e + numX], hD = s.h[e + numX + 1];
        if (up)
          triangle_plane(xA, hA, zA, xB, hB, zA, xA, hC, zC, true, base);
        else
          triangle_plane(xB, hD, zC, xB, hB, zA, xA, hC, zC, false, base);
      }
      float cpos[4][3];
      const int nc = box_plane_contacts(b, base[0], base[1], base[2], base[3], 10, cpos);
      bool hit = false;
      for (int sl = first_slot; sl < slots; ++sl) {
        const int j = lane + 64 * sl;
        if (j < T && j >= k && !((done >> sl) & 1ull)) {
          const int id = s.tri[j];
          const int c = id >> 1;
          const bool up = !(id & 1);
          const int cx = c / cellsZ, cz = c - cx * cellsZ;
          bool same = (j == k);
          if (!same) {
            const int e = cz * numX + cx;
            const float xA = (float)(b.minX + cx) * f.sample_w,
                        xB = (float)(b.minX + cx + 1) * f.sample_w;
            const float zA = (float)(b.minZ + cz) * f.sample_d,
                        zC = (float)(b.minZ + cz + 1) * f.sample_d;
            const float hA = s.h[e], hB = s.h[e + 1], hC = s.h[e + numX], hD = s.h[e + numX + 1];
            float pl[4];
            if (up)
              triangle_plane(xA, hA, zA, xB, hB, zA, xA, hC, zC, true, pl);
            else
              triangle_plane(xB, hD, zC, xB, hB, zA, xA, hC, zC, false, pl);
            same = planes_eps_equal(base, pl);
          }
          if (same) {
            done |= (1ull << sl);
            // corner vertex: A = (cx, cz) for ABC, D = (cx+1, cz+1) for DBC (global sample coords)
            const int gx = b.minX + cx + (up ? 0 : 1);
            const int gz = b.minZ + cz + (up ? 0 : 1);
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (i < nc) hit = hit || is_on_heightfield2(f, gx, gz, cpos[i][0], cpos[i][2], up);
          }
        }
      }
      if (__any(hit)) {
        *exit_code = EXIT_PLANE;
        return 1;
      }
    }
  }
  *exit_code = EXIT_NONE;
  return 0;
}

}  // namespace artp
