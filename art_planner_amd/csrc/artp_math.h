// artp_math.h -- exact float32 building blocks shared by every validity kernel.
//
// Everything here must round exactly like the reference's baseline x86-64 build of the patched ODE
// (one IEEE rounding per operation, no FMA contraction, correctly rounded divide / sqrt).  The
// translation unit is compiled with -ffp-contract=off; hipcc's default
// -fhip-fp32-correctly-rounded-divide-sqrt keeps `/` and sqrtf correctly rounded on gfx950, and f32
// denormals are preserved (float_denorm_mode_32 = 3).
//
// Reference functions restated (paths relative to the reference tree):
//   _dCalcVectorDot3      ode/include/ode/odemath.h:213-216
//   _dCalcVectorCross3    ode/include/ode/odemath.h:234-243
//   dxSafeNormalize3      ode/ode/src/odemath.cpp:95-162
//   dxOrthogonalizeR      ode/ode/src/odemath.cpp:260-313
//   dxBox::computeAABB    ode/ode/src/box.cpp:60-77
//   dGeomBoxPointDepth    ode/ode/src/box.cpp:109-173
//   dCollideBoxPlane      ode/ode/src/box.cpp:745-880
//   IsOnHeightfield2      ode/ode/src/heightfield.cpp:264-321
//   dCollideHeightfield   ode/ode/src/heightfield.cpp:1838-1893 (frame change + index window)
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define ARTP_HD __host__ __device__ __forceinline__
#define ARTP_EPS 1.1920928955078125e-07f /* dEpsilon = FLT_EPSILON, ode/ode/src/common.h:42 */

namespace artp {

// Layer as the patched ODE sees it (dxHeightfieldData::SetData, heightfield.cpp:130-169 +
// HeightMapBoxChecker ctor/setHeightField, height_map_box_checker.cpp:11-26,38-54).
struct FieldDev {
  const float* data;  // ODE sample layout in HBM: h(x, z) = data[x + z*nW]  (x fastest, coalesced)
  int nW, nD;
  float width, depth, half_w, half_d;
  float sample_w, sample_d, inv_w, inv_d, zx_aspect;
  float pos[3];
  float R[12];
  int has_nan;  // any NaN sample in the layer (enables the running-dMAX quirk path)
  // Partner pre-filter of the plane stage (box_check.h): two triangles whose normalised planes are
  // epsilon-equal have raw cross products that differ by at most partner_tol * |cross| per component
  // (derivation in DESIGN.md 4.1); +inf disables the filter.
  float partner_tol;
  // Partner table (pipeline.h: partner_flags_kernel), indexed like `data` by the cell's lower vertex:
  // bit 0 / bit 1 = the ABC / DBC triangle of the cell has ANOTHER all-finite triangle with an
  // epsilon-equal plane within Chebyshev distance partner_R cells.  Planes depend on the map only, so a
  // window of at most partner_R x partner_R cells whose corner candidates are all unflagged cannot hold
  // a partner.  nullptr = not built for this layer.
  const unsigned char* partner_flags;
  int partner_R;
};

// Box in heightfield frame, ready for the zone test.
// base[idx] with the BYTE offset formed in 32 bits: the load becomes `global_load v, v_off, s[base]` (uniform
// base in SGPRs, one VGPR of offset) instead of a per-lane 64-bit address built with three more VALU operations.
// Every table and layer this is used on is far smaller than 4 GB.
template <class T>
ARTP_HD T gather32(const T* __restrict__ base, unsigned idx) {
  return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + idx * (unsigned)sizeof(T));
}

struct BoxHF {
  float pos[3];
  float R[9];  // row-major 3x3 (R1 = Rt^T * R)
  float side[3];
  float aabb[6];
  int minX, maxX, minZ, maxZ;  // clamped index window
  int on_field;                // 0 = AABB rejected (heightfield.cpp:1868-1877)
};

ARTP_HD float dot3(float a0, float a1, float a2, float b0, float b1, float b2) {
  return a0 * b0 + a1 * b1 + a2 * b2;
}

ARTP_HD bool is_finite(float v) { return (__float_as_uint(v) & 0x7f800000u) != 0x7f800000u; }
ARTP_HD bool is_nan(float v) { return v != v; }

// nextafterf(x, +-inf) for finite x (what heightfield.cpp:1881-1885 needs); NaN/inf pass through.
ARTP_HD float next_toward_neg_inf(float x) {
  uint32_t u = __float_as_uint(x);
  if ((u & 0x7f800000u) == 0x7f800000u) return x;
  if ((u << 1) == 0u) return __uint_as_float(0x80000001u);  // +-0 -> -min subnormal
  return __uint_as_float((u & 0x80000000u) ? u + 1u : u - 1u);
}
ARTP_HD float next_toward_pos_inf(float x) {
  uint32_t u = __float_as_uint(x);
  if ((u & 0x7f800000u) == 0x7f800000u) return x;
  if ((u << 1) == 0u) return __uint_as_float(0x00000001u);
  return __uint_as_float((u & 0x80000000u) ? u - 1u : u + 1u);
}

ARTP_HD void safe_normalize3(float& a0, float& a1, float& a2) {
  const float abs0 = fabsf(a0), abs1 = fabsf(a1), abs2 = fabsf(a2);
  int idx;
  if (abs1 > abs0) {
    idx = (abs2 > abs1) ? 2 : 1;
  } else if (abs2 > abs0) {
    idx = 2;
  } else {
    if (!(abs0 > 0.0f)) return;
    idx = 0;
  }
  // The three branches of the reference differ only in which component is the largest; written with
  // value selects (not per-branch stores through references, which the compiler turns into a
  // dynamically indexed scratch array).  "first"/"second" = the other two components in index order, so
  // 1 + first^2 + second^2 is summed exactly like each branch of dxSafeNormalize3.
  const float big = idx == 0 ? a0 : (idx == 1 ? a1 : a2);
  const float first = idx == 0 ? a1 : a0;
  const float second = idx == 2 ? a1 : a2;
  const float recip = 1.0f / fabsf(big);
  const float p1 = first * recip, p2 = second * recip;
  const float l = 1.0f / sqrtf(1.0f + p1 * p1 + p2 * p2);
  const float nb = copysignf(l, big), n1 = p1 * l, n2 = p2 * l;
  a0 = idx == 0 ? nb : n1;
  a1 = idx == 1 ? nb : (idx == 0 ? n1 : n2);
  a2 = idx == 2 ? nb : n2;
}

// dBodySetRotation's dxOrthogonalizeR on the 3x3 part of a row-major rotation, all nine entries as
// scalars (a pointer-to-array version ends up in per-lane scratch memory on the GPU).
ARTP_HD void orthogonalize_R9(float& m0, float& m1, float& m2, float& m4, float& m5, float& m6, float& m8,
                              float& m9, float& m10) {
  if (!(m0 != 0.0f || m1 != 0.0f || m2 != 0.0f)) return;
  const float n0 = m0 * m0 + m1 * m1 + m2 * m2;
  const float proj = dot3(m0, m1, m2, m4, m5, m6);
  float r0 = m4, r1 = m5, r2 = m6;
  const bool in_place = !(proj != 0);
  if (!in_place) {
    const float pd = proj / n0;
    r0 = m4 - pd * m0;
    r1 = m5 - pd * m1;
    r2 = m6 - pd * m2;
  }
  if (!(r0 != 0.0f || r1 != 0.0f || r2 != 0.0f)) return;
  if (n0 != 1.0f) safe_normalize3(m0, m1, m2);
  const float n1 = r0 * r0 + r1 * r1 + r2 * r2;
  if (n1 != 1.0f) safe_normalize3(r0, r1, r2);
  if (in_place) {  // the Gram-Schmidt temporary aliases row 1 only when proj == 0
    m4 = r0;
    m5 = r1;
    m6 = r2;
  }
  m8 = m1 * r2 - m2 * r1;
  m9 = m2 * r0 - m0 * r2;
  m10 = m0 * r1 - m1 * r0;
}

// same on a 3x4 row-major matrix m[12] (host side: the field rotation at map upload)
ARTP_HD void orthogonalize_R(float* m) {
  orthogonalize_R9(m[0], m[1], m[2], m[4], m[5], m[6], m[8], m[9], m[10]);
  if (m[0] != 0.0f || m[1] != 0.0f || m[2] != 0.0f) m[3] = m[7] = m[11] = 0.0f;
}

// The rotation part of HeightMapBoxChecker::checkCollision's pose handling, which depends on the STATE only (the
// five boxes of a state share its rotation, validity_checker.cpp:40-43 / validity_checker_feet.cpp:64-68):
// dBodySetRotation's dxOrthogonalizeR, then dCollideHeightfield's frame change R1 = Rt^T R (dMultiply1_333,
// heightfield.cpp:1846).  rot = row-major 3x3 (the dPose's rotation without its padding column).
ARTP_HD void box_rotation_in_field(const FieldDev& f, const float rot[9], float bR[9]) {
  float w0 = rot[0], w1 = rot[1], w2 = rot[2], w4 = rot[3], w5 = rot[4], w6 = rot[5], w8 = rot[6], w9 = rot[7],
        w10 = rot[8];
  orthogonalize_R9(w0, w1, w2, w4, w5, w6, w8, w9, w10);
  const float Rw[12] = {w0, w1, w2, 0.0f, w4, w5, w6, 0.0f, w8, w9, w10, 0.0f};
  // dMultiply1_333(R1, Rt, R): R1[i][j] = R[0][j]*Rt[0][i] + R[1][j]*Rt[1][i] + R[2][j]*Rt[2][i]
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      bR[3 * i + j] = dot3(Rw[j], Rw[4 + j], Rw[8 + j], f.R[i], f.R[4 + i], f.R[8 + i]);
}

// The rest of the pose handling for one box: position in the field frame, AABB, on-field test, index window
// (heightfield.cpp:1838-1893, box.cpp:60-77).  origin = the dPose's origin, bR = box_rotation_in_field().
ARTP_HD void setup_box_rotated(const FieldDev& f, const float origin[3], const float bR[9], float sx, float sy,
                               float sz, BoxHF& b) {
  const float p0 = origin[0] - f.pos[0];
  const float p1 = origin[1] - f.pos[1];
  const float p2 = origin[2] - f.pos[2];
  // dMultiply1_331(pos1, Rt, pos0): pos1[i] = Rt[i]*p0 + Rt[4+i]*p1 + Rt[8+i]*p2
  b.pos[0] = dot3(f.R[0], f.R[4], f.R[8], p0, p1, p2);
  b.pos[1] = dot3(f.R[1], f.R[5], f.R[9], p0, p1, p2);
  b.pos[2] = dot3(f.R[2], f.R[6], f.R[10], p0, p1, p2);
#pragma unroll
  for (int i = 0; i < 9; ++i) b.R[i] = bR[i];
  b.pos[0] += f.half_w;
  b.pos[2] += f.half_d;
  b.side[0] = sx;
  b.side[1] = sy;
  b.side[2] = sz;
  const float xr = 0.5f * (fabsf(b.R[0] * sx) + fabsf(b.R[1] * sy) + fabsf(b.R[2] * sz));
  const float yr = 0.5f * (fabsf(b.R[3] * sx) + fabsf(b.R[4] * sy) + fabsf(b.R[5] * sz));
  const float zr = 0.5f * (fabsf(b.R[6] * sx) + fabsf(b.R[7] * sy) + fabsf(b.R[8] * sz));
  b.aabb[0] = b.pos[0] - xr;
  b.aabb[1] = b.pos[0] + xr;
  b.aabb[2] = b.pos[1] - yr;
  b.aabb[3] = b.pos[1] + yr;
  b.aabb[4] = b.pos[2] - zr;
  b.aabb[5] = b.pos[2] + zr;
  b.on_field = !(b.aabb[0] > f.width || b.aabb[4] > f.depth) && !(b.aabb[1] < 0 || b.aabb[5] < 0);
  b.minX = b.maxX = b.minZ = b.maxZ = 0;
  if (b.on_field) {
    int nMinX = (int)floorf(next_toward_neg_inf(b.aabb[0] * f.inv_w));
    int nMaxX = (int)ceilf(next_toward_pos_inf(b.aabb[1] * f.inv_w));
    int nMinZ = (int)floorf(next_toward_neg_inf(b.aabb[4] * f.inv_d));
    int nMaxZ = (int)ceilf(next_toward_pos_inf(b.aabb[5] * f.inv_d));
    b.minX = nMinX > 0 ? nMinX : 0;
    b.maxX = nMaxX > f.nW - 1 ? f.nW - 1 : nMaxX;
    b.minZ = nMinZ > 0 ? nMinZ : 0;
    b.maxZ = nMaxZ > f.nD - 1 ? f.nD - 1 : nMaxZ;
  }
}

// HeightMapBoxChecker::checkCollision pose -> box in heightfield frame + index window.
// pose = dPose{origin[4], rotation[12]}.
ARTP_HD void setup_box(const FieldDev& f, const float* pose, float sx, float sy, float sz,
                       BoxHF& b) {
  const float rot[9] = {pose[4], pose[5], pose[6], pose[8], pose[9], pose[10], pose[12], pose[13], pose[14]};
  float bR[9];
  box_rotation_in_field(f, rot, bR);
  setup_box_rotated(f, pose, bR, sx, sy, sz, b);
}

// dGeomBoxPointDepth(...) > dEpsilon  (ode/ode/src/box.cpp:109-173).
// The reference forms the six face distances d = h -+ q, "inside" = none negative, and the depth = the
// smallest of them (started from (dReal)(unsigned)-1).  For a point inside, depth > eps is min(d) > eps, and
// min(h - q, h + q) = h - |q| in exact IEEE arithmetic (the same single subtraction either way), so the
// whole test is  min_i (h_i - |q_i|) > eps.  For a point outside some d is negative: the reference returns
// a non-positive depth, and the minimum here is negative too.
ARTP_HD bool point_in_box(const BoxHF& b, float x, float y, float z) {
  const float p0 = x - b.pos[0], p1 = y - b.pos[1], p2 = z - b.pos[2];
  const float q0 = dot3(b.R[0], b.R[3], b.R[6], p0, p1, p2);
  const float q1 = dot3(b.R[1], b.R[4], b.R[7], p0, p1, p2);
  const float q2 = dot3(b.R[2], b.R[5], b.R[8], p0, p1, p2);
  const float m0 = b.side[0] * 0.5f - fabsf(q0);
  const float m1 = b.side[1] * 0.5f - fabsf(q1);
  const float m2 = b.side[2] * 0.5f - fabsf(q2);
  return fminf(fminf(m0, m1), m2) > ARTP_EPS;
}

// dCollideBoxPlane: up to maxc (<= 4) contact positions; returns their number.
ARTP_HD int box_plane_contacts(const BoxHF& b, float n0, float n1, float n2, float d, int maxc,
                               float cpos[4][3]) {
  const float* R = b.R;
  const float Q1 = dot3(n0, n1, n2, R[0], R[3], R[6]);
  const float Q2 = dot3(n0, n1, n2, R[1], R[4], R[7]);
  const float Q3 = dot3(n0, n1, n2, R[2], R[5], R[8]);
  float A[3], B[3];
  A[0] = b.side[0] * Q1;
  A[1] = b.side[1] * Q2;
  A[2] = b.side[2] * Q3;
  B[0] = fabsf(A[0]);
  B[1] = fabsf(A[1]);
  B[2] = fabsf(A[2]);
  const float depth = d + 0.5f * (B[0] + B[1] + B[2]) - dot3(n0, n1, n2, b.pos[0], b.pos[1], b.pos[2]);
  if (depth < 0) return 0;
  if (maxc > 4) maxc = 4;
  float p[3] = {b.pos[0], b.pos[1], b.pos[2]};
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float hs = 0.5f * b.side[i];
    if (A[i] > 0) {
      p[0] = p[0] - hs * R[0 + i];
      p[1] = p[1] - hs * R[3 + i];
      p[2] = p[2] - hs * R[6 + i];
    } else {
      p[0] = p[0] + hs * R[0 + i];
      p[1] = p[1] + hs * R[3 + i];
      p[2] = p[2] + hs * R[6 + i];
    }
  }
  cpos[0][0] = p[0];
  cpos[0][1] = p[1];
  cpos[0][2] = p[2];
  int ret = 1;
  if (maxc == 1) return ret;
  int s1, s2;
  if (B[0] < B[1]) {
    if (B[2] < B[0]) {
      s1 = 2;
      s2 = (B[0] < B[1]) ? 0 : 1;
    } else {
      s1 = 0;
      s2 = (B[1] < B[2]) ? 1 : 2;
    }
  } else {
    if (B[2] < B[1]) {
      s1 = 2;
      s2 = (B[0] < B[1]) ? 0 : 1;
    } else {
      s1 = 1;
      s2 = (B[0] < B[2]) ? 0 : 2;
    }
  }
  // select by comparison instead of dynamic indexing (keeps everything in registers)
  const float side1 = s1 == 0 ? b.side[0] : (s1 == 1 ? b.side[1] : b.side[2]);
  const float A1 = s1 == 0 ? A[0] : (s1 == 1 ? A[1] : A[2]);
  const float B1 = s1 == 0 ? B[0] : (s1 == 1 ? B[1] : B[2]);
  const float r10 = s1 == 0 ? R[0] : (s1 == 1 ? R[1] : R[2]);
  const float r11 = s1 == 0 ? R[3] : (s1 == 1 ? R[4] : R[5]);
  const float r12 = s1 == 0 ? R[6] : (s1 == 1 ? R[7] : R[8]);
  float depth1 = 0.0f, depth2 = 0.0f;
  if (!(depth - B1 < 0)) {
    if (A1 > 0) {
      cpos[1][0] = p[0] + side1 * r10;
      cpos[1][1] = p[1] + side1 * r11;
      cpos[1][2] = p[2] + side1 * r12;
    } else {
      cpos[1][0] = p[0] - side1 * r10;
      cpos[1][1] = p[1] - side1 * r11;
      cpos[1][2] = p[2] - side1 * r12;
    }
    depth1 = depth - B1;
    ret = 2;
    if (maxc > 2) {
      const float side2 = s2 == 0 ? b.side[0] : (s2 == 1 ? b.side[1] : b.side[2]);
      const float A2 = s2 == 0 ? A[0] : (s2 == 1 ? A[1] : A[2]);
      const float B2 = s2 == 0 ? B[0] : (s2 == 1 ? B[1] : B[2]);
      const float r20 = s2 == 0 ? R[0] : (s2 == 1 ? R[1] : R[2]);
      const float r21 = s2 == 0 ? R[3] : (s2 == 1 ? R[4] : R[5]);
      const float r22 = s2 == 0 ? R[6] : (s2 == 1 ? R[7] : R[8]);
      if (!(depth - B2 < 0)) {
        if (A2 > 0) {
          cpos[2][0] = p[0] + side2 * r20;
          cpos[2][1] = p[1] + side2 * r21;
          cpos[2][2] = p[2] + side2 * r22;
        } else {
          cpos[2][0] = p[0] - side2 * r20;
          cpos[2][1] = p[1] - side2 * r21;
          cpos[2][2] = p[2] - side2 * r22;
        }
        depth2 = depth - B2;
        ret = 3;
      }
    }
  }
  if (maxc == 4 && ret == 3) {
    const float d4 = depth1 + depth2 - depth;
    if (d4 > 0) {
      cpos[3][0] = cpos[1][0] + cpos[2][0] - p[0];
      cpos[3][1] = cpos[1][1] + cpos[2][1] - p[1];
      cpos[3][2] = cpos[1][2] + cpos[2][2] - p[2];
      ret = 4;
    }
  }
  return ret;
}

// IsOnHeightfield2 with the cell corner given by its integer sample coordinates (cx, cz):
// corner vertex = (cx*sample_w, *, cz*sample_d).
ARTP_HD bool is_on_heightfield2(const FieldDev& f, int cx, int cz, float px, float pz, bool is_abc) {
  if (is_abc) {
    const float MinX = (float)cx * f.sample_w;
    if (px < MinX) return false;
    const float MaxX = (float)(cx + 1) * f.sample_w;
    if (px >= MaxX) return false;
    const float MinZ = (float)cz * f.sample_d;
    if (pz < MinZ) return false;
    const float MaxZ = (float)(cz + 1) * f.sample_d;
    if (pz >= MaxZ) return false;
    return (MaxZ - pz) > (px - MinX) * f.zx_aspect;
  } else {
    const float MaxX = (float)cx * f.sample_w;
    if (px >= MaxX) return false;
    const float MinX = (float)(cx - 1) * f.sample_w;
    if (px < MinX) return false;
    const float MaxZ = (float)cz * f.sample_d;
    if (pz >= MaxZ) return false;
    const float MinZ = (float)(cz - 1) * f.sample_d;
    if (pz < MinZ) return false;
    return (MaxZ - pz) <= (px - MinX) * f.zx_aspect;
  }
}

// Plane (n, d) of a heightfield triangle (heightfield.cpp:1474-1501).
// up:   vertices (A,B,C): Edge1 = C-A, Edge2 = B-A, n = Edge1 x Edge2
// down: vertices (D,B,C): Edge1 = C-D, Edge2 = B-D, n = Edge2 x Edge1
// raw (optional) receives the un-normalised cross product.
ARTP_HD void triangle_plane(float v0x, float v0y, float v0z, float v1x, float v1y, float v1z,
                            float v2x, float v2y, float v2z, bool is_up, float pl[4],
                            float* raw = nullptr) {
  const float e1x = v2x - v0x, e1y = v2y - v0y, e1z = v2z - v0z;
  const float e2x = v1x - v0x, e2y = v1y - v0y, e2z = v1z - v0z;
  float ax, ay, az, bx, by, bz;
  if (is_up) {
    ax = e1x; ay = e1y; az = e1z; bx = e2x; by = e2y; bz = e2z;
  } else {
    ax = e2x; ay = e2y; az = e2z; bx = e1x; by = e1y; bz = e1z;
  }
  float t0 = ay * bz - az * by;
  float t1 = az * bx - ax * bz;
  float t2 = ax * by - ay * bx;
  if (raw) {
    raw[0] = t0;
    raw[1] = t1;
    raw[2] = t2;
  }
  const float dinv = 1.0f / sqrtf(t0 * t0 + t1 * t1 + t2 * t2);
  t0 *= dinv;
  t1 *= dinv;
  t2 *= dinv;
  pl[0] = t0;
  pl[1] = t1;
  pl[2] = t2;
  pl[3] = dot3(t0, t1, t2, v0x, v0y, v0z);
}

// Un-normalised cross product of a heightfield triangle: exactly the (t0, t1, t2) of triangle_plane.
ARTP_HD void triangle_cross(float v0x, float v0y, float v0z, float v1x, float v1y, float v1z, float v2x,
                            float v2y, float v2z, bool is_up, float raw[3]) {
  const float e1x = v2x - v0x, e1y = v2y - v0y, e1z = v2z - v0z;
  const float e2x = v1x - v0x, e2y = v1y - v0y, e2z = v1z - v0z;
  float ax, ay, az, bx, by, bz;
  if (is_up) {
    ax = e1x; ay = e1y; az = e1z; bx = e2x; by = e2y; bz = e2z;
  } else {
    ax = e2x; ay = e2y; az = e2z; bx = e1x; by = e1y; bz = e1z;
  }
  raw[0] = ay * bz - az * by;
  raw[1] = az * bx - ax * bz;
  raw[2] = ax * by - ay * bx;
}

ARTP_HD bool planes_eps_equal(const float a[4], const float b[4]) {
  return fabsf(a[1] - b[1]) < ARTP_EPS && fabsf(a[3] - b[3]) < ARTP_EPS &&
         fabsf(a[0] - b[0]) < ARTP_EPS && fabsf(a[2] - b[2]) < ARTP_EPS;
}

}  // namespace artp
