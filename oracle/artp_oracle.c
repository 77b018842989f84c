/*
 * artp_oracle.c -- CPU ORACLE (test infrastructure only; see artp_oracle.h for the contract).
 *
 * Plain-C, float32, operation-for-operation restatement of art_planner's validity hot path.
 * "Faithful mode": the algorithmic structure of ODE's dCollideHeightfieldZone is kept (window vertex
 * buffer, per-triangle plane, greedy O(T^2) plane grouping, bubble sort, vertex re-pass) so that
 * timing this file is a fair stand-in for the reference CPU path on machines where
 * /root/reference cannot be built (the GPU box).
 *
 * Must be compiled with -ffp-contract=off (no FMA) on an ABI with FLT_EVAL_METHOD == 0:
 * the reference build is baseline x86-64 SSE2 (art_planner/CMakeLists.txt:5, no -march).
 * All file:line references are relative to /root/reference.
 */
#include "artp_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#if defined(FLT_EVAL_METHOD) && FLT_EVAL_METHOD != 0
#error "artp_oracle.c needs FLT_EVAL_METHOD == 0 (float ops rounded to float)"
#endif

#define dEpsilon FLT_EPSILON /* ode/ode/src/common.h:42 */

/* ------------------------------------------------------------------------------------------------
 * ODE float helpers (ode/include/ode/odemath.h)
 * ---------------------------------------------------------------------------------------------- */

/* _dCalcVectorDot3, odemath.h:213-216: a0*b0 + a1*b1 + a2*b2, left to right. */
static inline float dot3(const float* a, const float* b, int sa, int sb) {
  return a[0] * b[0] + a[sa] * b[sb] + a[2 * sa] * b[2 * sb];
}

/* _dCalcVectorCross3, odemath.h:234-243 */
static inline void cross3(float* res, const float* a, const float* b) {
  const float r0 = a[1] * b[2] - a[2] * b[1];
  const float r1 = a[2] * b[0] - a[0] * b[2];
  const float r2 = a[0] * b[1] - a[1] * b[0];
  res[0] = r0;
  res[1] = r1;
  res[2] = r2;
}

/* dxCouldBeNormalized3, ode/ode/src/odemath.cpp:68-82 */
static int could_be_normalized3(const float* a) {
  return a[0] != 0.0f || a[1] != 0.0f || a[2] != 0.0f;
}

/* dxSafeNormalize3, ode/ode/src/odemath.cpp:95-162 (max-component-scaled normalisation). */
static int safe_normalize3(float* a) {
  const float abs_a0 = fabsf(a[0]);
  const float abs_a1 = fabsf(a[1]);
  const float abs_a2 = fabsf(a[2]);
  int idx;
  if (abs_a1 > abs_a0) {
    idx = (abs_a2 > abs_a1) ? 2 : 1;
  } else if (abs_a2 > abs_a0) {
    idx = 2;
  } else {
    if (!(abs_a0 > 0.0f)) return 0;
    idx = 0;
  }
  if (idx == 0) {
    const float recip = 1.0f / abs_a0;
    const float a1 = a[1] * recip;
    const float a2 = a[2] * recip;
    const float l = 1.0f / sqrtf(1.0f + a1 * a1 + a2 * a2);
    a[1] = a1 * l;
    a[2] = a2 * l;
    a[0] = copysignf(l, a[0]);
  } else if (idx == 1) {
    const float recip = 1.0f / abs_a1;
    const float a0 = a[0] * recip;
    const float a2 = a[2] * recip;
    const float l = 1.0f / sqrtf(1.0f + a0 * a0 + a2 * a2);
    a[0] = a0 * l;
    a[2] = a2 * l;
    a[1] = copysignf(l, a[1]);
  } else {
    const float recip = 1.0f / abs_a2;
    const float a0 = a[0] * recip;
    const float a1 = a[1] * recip;
    const float l = 1.0f / sqrtf(1.0f + a0 * a0 + a1 * a1);
    a[0] = a0 * l;
    a[1] = a1 * l;
    a[2] = copysignf(l, a[2]);
  }
  return 1;
}

/* dxOrthogonalizeR, ode/ode/src/odemath.cpp:260-313.  Called by dBodySetRotation (ode.cpp:359-374)
 * on a memcpy of the caller's matrix.  NOTE the quirk: when proj != 0 the Gram-Schmidt result lives
 * in a temporary, so row 1 OF THE MATRIX stays the caller's row 1. */
static int orthogonalize_R(float* m) {
  if (!could_be_normalized3(m + 0)) return 0;
  const float n0 = m[0] * m[0] + m[1] * m[1] + m[2] * m[2];
  float row2_store[3];
  float* row2 = m + 4;
  const float proj = dot3(m + 0, m + 4, 1, 1);
  if (proj != 0) {
    const float proj_div_n0 = proj / n0;
    row2_store[0] = m[4] - proj_div_n0 * m[0];
    row2_store[1] = m[5] - proj_div_n0 * m[1];
    row2_store[2] = m[6] - proj_div_n0 * m[2];
    row2 = row2_store;
  }
  if (!could_be_normalized3(row2)) return 0;
  if (n0 != 1.0f) safe_normalize3(m + 0);
  const float n1 = row2[0] * row2[0] + row2[1] * row2[1] + row2[2] * row2[2];
  if (n1 != 1.0f) safe_normalize3(row2);
  cross3(m + 8, m + 0, row2);
  m[3] = m[7] = m[11] = 0;
  return 1;
}

/* dRFrom2Axes, ode/ode/src/rotation.cpp:94-130 */
static void r_from_2_axes(float* R, float ax, float ay, float az, float bx, float by, float bz) {
  float l, k;
  l = sqrtf(ax * ax + ay * ay + az * az);
  if (l <= 0.0f) return;
  l = 1.0f / l;
  ax *= l;
  ay *= l;
  az *= l;
  k = ax * bx + ay * by + az * bz;
  bx -= k * ax;
  by -= k * ay;
  bz -= k * az;
  l = sqrtf(bx * bx + by * by + bz * bz);
  if (l <= 0.0f) return;
  l = 1.0f / l;
  bx *= l;
  by *= l;
  bz *= l;
  R[0] = ax;
  R[4] = ay;
  R[8] = az;
  R[1] = bx;
  R[5] = by;
  R[9] = bz;
  R[2] = -by * az + ay * bz;
  R[6] = -bz * ax + az * bx;
  R[10] = -bx * ay + ax * by;
  R[3] = 0.0f;
  R[7] = 0.0f;
  R[11] = 0.0f;
}

/* ------------------------------------------------------------------------------------------------
 * R5: HeightMapBoxChecker ctor + setHeightField
 * ---------------------------------------------------------------------------------------------- */
void artp_oracle_field_init(artp_oracle_field* f, float* storage, const float* layer, int rows,
                            int cols, double len_x, double len_y, double pos_x, double pos_y) {
  /* field_.mat = layer.rowwise().reverse(): mat(i,j) = layer(i, cols-1-j), col-major
   * (art_planner/src/validity_checker/height_map_box_checker.cpp:44). */
  for (int j = 0; j < cols; ++j)
    for (int i = 0; i < rows; ++i)
      storage[(size_t)i + (size_t)j * rows] = layer[(size_t)i + (size_t)(cols - 1 - j) * rows];
  f->data = storage;
  /* dGeomHeightfieldDataBuildSingle(data, mat, 0, length.x, length.y, size.x, size.y, 1,0,0,0)
   * (height_map_box_checker.cpp:49) -> dxHeightfieldData::SetData (heightfield.cpp:130-169). */
  f->nW = rows;
  f->nD = cols;
  f->width = (float)len_x;
  f->depth = (float)len_y;
  f->half_w = f->width / 2.0f;
  f->half_d = f->depth / 2.0f;
  f->sample_w = f->width / ((float)f->nW - 1.0f);
  f->sample_d = f->depth / ((float)f->nD - 1.0f);
  f->zx_aspect = f->sample_d / f->sample_w;
  f->inv_w = 1.0f / f->sample_w;
  f->inv_d = 1.0f / f->sample_d;
  /* dBodySetPosition(body_field_, pos.x, pos.y, 0) (height_map_box_checker.cpp:53) */
  f->pos[0] = (float)pos_x;
  f->pos[1] = (float)pos_y;
  f->pos[2] = 0.0f;
  /* dRFrom2Axes(rot, -1,0,0, 0,0,1); dBodySetRotation(body_field_, rot)
   * (height_map_box_checker.cpp:22,25). */
  memset(f->R, 0, sizeof(f->R));
  r_from_2_axes(f->R, -1, 0, 0, 0, 0, 1);
  orthogonalize_R(f->R);
}

/* ------------------------------------------------------------------------------------------------
 * Box helpers (ode/ode/src/box.cpp)
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  float pos[3];
  float R[12];
  float side[3];
  float aabb[6];
} box_t;

/* dxBox::computeAABB, box.cpp:60-77 */
static void box_compute_aabb(box_t* b) {
  const float* R = b->R;
  const float* side = b->side;
  const float xrange =
      0.5f * (fabsf(R[0] * side[0]) + fabsf(R[1] * side[1]) + fabsf(R[2] * side[2]));
  const float yrange =
      0.5f * (fabsf(R[4] * side[0]) + fabsf(R[5] * side[1]) + fabsf(R[6] * side[2]));
  const float zrange =
      0.5f * (fabsf(R[8] * side[0]) + fabsf(R[9] * side[1]) + fabsf(R[10] * side[2]));
  b->aabb[0] = b->pos[0] - xrange;
  b->aabb[1] = b->pos[0] + xrange;
  b->aabb[2] = b->pos[1] - yrange;
  b->aabb[3] = b->pos[1] + yrange;
  b->aabb[4] = b->pos[2] - zrange;
  b->aabb[5] = b->pos[2] + zrange;
}

/* dGeomBoxPointDepth, box.cpp:109-173 */
static float box_point_depth(const box_t* b, float x, float y, float z) {
  float p[3], q[3], dist[6];
  p[0] = x - b->pos[0];
  p[1] = y - b->pos[1];
  p[2] = z - b->pos[2];
  /* dMultiply1_331: q = R^T p, odemath.h:326-333 */
  q[0] = dot3(b->R + 0, p, 4, 1);
  q[1] = dot3(b->R + 1, p, 4, 1);
  q[2] = dot3(b->R + 2, p, 4, 1);
  int inside = 1;
  for (int i = 0; i < 3; i++) {
    const float side = b->side[i] * 0.5f;
    dist[i] = side - q[i];
    dist[i + 3] = side + q[i];
    if ((dist[i] < 0) || (dist[i + 3] < 0)) inside = 0;
  }
  if (inside) {
    float smallest = (float)(unsigned)-1;
    for (int i = 0; i < 6; i++)
      if (dist[i] < smallest) smallest = dist[i];
    return smallest;
  }
  float largest = 0;
  for (int i = 0; i < 6; i++)
    if (dist[i] > largest) largest = dist[i];
  return -largest;
}

/* dCollideBoxPlane, box.cpp:745-880.  plane = (n, d).  Writes up to maxc (<=4) contact positions,
 * returns their number. */
static int box_plane_contacts(const box_t* box, const float* plane, int maxc, float cpos[4][3]) {
  const float* R = box->R;
  const float* n = plane;
  const float Q1 = dot3(n, R + 0, 1, 4);
  const float Q2 = dot3(n, R + 1, 1, 4);
  const float Q3 = dot3(n, R + 2, 1, 4);
  const float A1 = box->side[0] * Q1;
  const float A2 = box->side[1] * Q2;
  const float A3 = box->side[2] * Q3;
  const float B1 = fabsf(A1);
  const float B2 = fabsf(A2);
  const float B3 = fabsf(A3);
  const float A[3] = {A1, A2, A3};
  const float B[3] = {B1, B2, B3};
  float cdepth[4];

  const float depth = plane[3] + 0.5f * (B1 + B2 + B3) - dot3(n, box->pos, 1, 1);
  if (depth < 0) return 0;
  if (maxc > 4) maxc = 4;

  float p[3] = {box->pos[0], box->pos[1], box->pos[2]};
  for (int i = 0; i < 3; i++) { /* BAR(0,1) BAR(1,2) BAR(2,3), box.cpp:789-798 */
    if (A[i] > 0) {
      p[0] -= 0.5f * box->side[i] * R[0 + i];
      p[1] -= 0.5f * box->side[i] * R[4 + i];
      p[2] -= 0.5f * box->side[i] * R[8 + i];
    } else {
      p[0] += 0.5f * box->side[i] * R[0 + i];
      p[1] += 0.5f * box->side[i] * R[4 + i];
      p[2] += 0.5f * box->side[i] * R[8 + i];
    }
  }
  cpos[0][0] = p[0];
  cpos[0][1] = p[1];
  cpos[0][2] = p[2];
  cdepth[0] = depth;
  int ret = 1;
  if (maxc == 1) goto done;

  { /* second and third contact: walk along the two sides with the smallest projection
     * (box.cpp:811-846).  side order decided exactly like the goto ladder. */
    int s1, s2;
    if (B1 < B2) {
      if (B3 < B1) {
        s1 = 2; /* use_side_3 */
        s2 = (B1 < B2) ? 0 : 1;
      } else {
        s1 = 0;
        s2 = (B2 < B3) ? 1 : 2;
      }
    } else {
      if (B3 < B2) {
        s1 = 2; /* use_side_3 */
        s2 = (B1 < B2) ? 0 : 1;
      } else {
        s1 = 1;
        s2 = (B1 < B3) ? 0 : 2;
      }
    }
    /* BAR(1, s1, s1+1) */
    if (depth - B[s1] < 0) goto done;
    if (A[s1] > 0) {
      cpos[1][0] = p[0] + box->side[s1] * R[0 + s1];
      cpos[1][1] = p[1] + box->side[s1] * R[4 + s1];
      cpos[1][2] = p[2] + box->side[s1] * R[8 + s1];
    } else {
      cpos[1][0] = p[0] - box->side[s1] * R[0 + s1];
      cpos[1][1] = p[1] - box->side[s1] * R[4 + s1];
      cpos[1][2] = p[2] - box->side[s1] * R[8 + s1];
    }
    cdepth[1] = depth - B[s1];
    ret++;
    if (maxc == 2) goto done;
    /* BAR(2, s2, s2+1) */
    if (depth - B[s2] < 0) goto done;
    if (A[s2] > 0) {
      cpos[2][0] = p[0] + box->side[s2] * R[0 + s2];
      cpos[2][1] = p[1] + box->side[s2] * R[4 + s2];
      cpos[2][2] = p[2] + box->side[s2] * R[8 + s2];
    } else {
      cpos[2][0] = p[0] - box->side[s2] * R[0 + s2];
      cpos[2][1] = p[1] - box->side[s2] * R[4 + s2];
      cpos[2][2] = p[2] - box->side[s2] * R[8 + s2];
    }
    cdepth[2] = depth - B[s2];
    ret++;
  }

done:
  if (maxc == 4 && ret == 3) { /* box.cpp:850-861 */
    const float d4 = cdepth[1] + cdepth[2] - depth;
    if (d4 > 0) {
      cpos[3][0] = cpos[1][0] + cpos[2][0] - p[0];
      cpos[3][1] = cpos[1][1] + cpos[2][1] - p[1];
      cpos[3][2] = cpos[1][2] + cpos[2][2] - p[2];
      ret++;
    }
  }
  return ret;
}

/* ------------------------------------------------------------------------------------------------
 * R4: dxHeightfield::dCollideHeightfieldZone, heightfield.cpp:973-1789
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  float vertex[3];
  int coords[2];
  int state;
} hf_vertex; /* HeightFieldVertex, heightfield.h:100-108 */

typedef struct {
  hf_vertex* vertices[3];
  float plane_def[4];
  float max_aaab;
  int is_up;
  int state;
} hf_triangle; /* HeightFieldTriangle, heightfield.h:118-136 */

typedef struct {
  int first;  /* index into the group-member list */
  int count;
  float max_aaab;
  float plane_def[4];
} hf_plane;

typedef struct { /* per-thread scratch, grown on demand (ODE keeps these in the geom) */
  hf_vertex* verts;
  size_t verts_cap;
  hf_triangle* tris;
  size_t tris_cap;
  hf_plane* planes;
  int* plane_order;
  int* members;
  size_t planes_cap;
} scratch_t;

static _Thread_local scratch_t g_scratch;
static _Thread_local unsigned g_last_num_tri; /* diagnostics: triangles kept by the last zone call */
unsigned artp_oracle_last_num_tri(void) { return g_last_num_tri; }

static void scratch_reserve(scratch_t* s, size_t nverts, size_t ntris) {
  if (s->verts_cap < nverts) {
    free(s->verts);
    s->verts = (hf_vertex*)malloc(nverts * sizeof(hf_vertex));
    s->verts_cap = nverts;
  }
  if (s->tris_cap < ntris) {
    free(s->tris);
    s->tris = (hf_triangle*)malloc((ntris ? ntris : 1) * sizeof(hf_triangle));
    s->tris_cap = ntris;
  }
  if (s->planes_cap < ntris) {
    free(s->planes);
    free(s->plane_order);
    free(s->members);
    s->planes = (hf_plane*)malloc((ntris ? ntris : 1) * sizeof(hf_plane));
    s->plane_order = (int*)malloc((ntris ? ntris : 1) * sizeof(int));
    s->members = (int*)malloc((ntris ? ntris : 1) * sizeof(int));
    s->planes_cap = ntris;
  }
}

/* dxHeightfieldData::GetHeight(int,int), heightfield.cpp:325-384 (finite mode, float data,
 * scale 1, offset 0). */
static inline float get_height(const artp_oracle_field* f, int x, int z) {
  if (x < 0) x = 0;
  if (z < 0) z = 0;
  if (x > f->nW - 1) x = f->nW - 1;
  if (z > f->nD - 1) z = f->nD - 1;
  const float h = f->data[x + (z * f->nW)];
  return (h * 1.0f) + 0.0f;
}

/* dxHeightfieldData::IsOnHeightfield2, heightfield.cpp:264-321 */
static int is_on_heightfield2(const artp_oracle_field* f, const hf_vertex* corner, const float* pos,
                              int is_abc) {
  float MaxX, MinX, MaxZ, MinZ;
  if (is_abc) {
    MinX = corner->vertex[0];
    if (pos[0] < MinX) return 0;
    MaxX = (float)(corner->coords[0] + 1) * f->sample_w;
    if (pos[0] >= MaxX) return 0;
    MinZ = corner->vertex[2];
    if (pos[2] < MinZ) return 0;
    MaxZ = (float)(corner->coords[1] + 1) * f->sample_d;
    if (pos[2] >= MaxZ) return 0;
    return (MaxZ - pos[2]) > (pos[0] - MinX) * f->zx_aspect;
  } else {
    MaxX = corner->vertex[0];
    if (pos[0] >= MaxX) return 0;
    MinX = (float)(corner->coords[0] - 1) * f->sample_w;
    if (pos[0] < MinX) return 0;
    MaxZ = corner->vertex[2];
    if (pos[2] >= MaxZ) return 0;
    MinZ = (float)(corner->coords[1] - 1) * f->sample_d;
    if (pos[2] < MinZ) return 0;
    return (MaxZ - pos[2]) <= (pos[0] - MinX) * f->zx_aspect;
  }
}

static int collide_zone(const artp_oracle_field* f, int minX, int maxX, int minZ, int maxZ,
                        const box_t* o2, int* exit_code) {
  const unsigned numX = (unsigned)((maxX - minX) + 1);
  const unsigned numZ = (unsigned)((maxZ - minZ) + 1);
  const float minO2Height = o2->aabb[2];
  const float maxO2Height = o2->aabb[3];
  float maxY = -INFINITY;
  float minY = INFINITY;
  int allFinite = 1;
  const float cfSampleWidth = f->sample_w;
  const float cfSampleDepth = f->sample_d;
  scratch_t* S = &g_scratch;
  g_last_num_tri = 0;
  const unsigned numTriMax = (unsigned)((maxX - minX) * (maxZ - minZ) * 2);
  scratch_reserve(S, (size_t)numX * numZ, numTriMax);
  hf_vertex* V = S->verts; /* V[x_local*numZ + z_local] == tempHeightBuffer[x_local][z_local] */

  /* (a) heightfield.cpp:1002-1026 */
  for (unsigned x_local = 0; x_local < numX; x_local++) {
    const int x = minX + (int)x_local;
    const float Xpos = (float)x * cfSampleWidth;
    hf_vertex* row = V + (size_t)x_local * numZ;
    for (unsigned z_local = 0; z_local < numZ; z_local++) {
      const int z = minZ + (int)z_local;
      const float Ypos = (float)z * cfSampleDepth;
      const float h = get_height(f, x, z);
      row[z_local].vertex[0] = Xpos;
      row[z_local].vertex[1] = h;
      row[z_local].vertex[2] = Ypos;
      row[z_local].coords[0] = x;
      row[z_local].coords[1] = z;
      maxY = (maxY > h) ? maxY : h; /* dMAX(maxY,h), heightfield.cpp:49 -- NaN replaces maxY */
      if (isfinite(h)) {
        minY = (minY > h) ? h : minY; /* dMIN(minY,h), heightfield.cpp:48 */
      } else {
        allFinite = 0;
      }
    }
  }
  if (minO2Height - maxY > -dEpsilon) { /* (b) :1027-1031 */
    *exit_code = ARTP_EXIT_ABOVE;
    return 0;
  }
  if (minY - maxO2Height > -dEpsilon) { /* (c) :1032-1058 */
    *exit_code = ARTP_EXIT_UNDER;
    return 0;
  }
  if (allFinite && minY - minO2Height > -dEpsilon && maxO2Height - maxY > -dEpsilon) {
    *exit_code = ARTP_EXIT_SPANS; /* (d) :1059-1064 */
    return 1;
  }
  if (allFinite) { /* (e) :1139-1160 */
    if (maxY - minY < dEpsilon) {
      const float triplane[4] = {0, 1, 0, minY};
      float cpos[4][3];
      *exit_code = ARTP_EXIT_FLAT_PLANE;
      return box_plane_contacts(o2, triplane, 1, cpos);
    }
  }

  /* needFurtherPasses, :1265-1280 */
  int needFurtherPasses = 0;
  {
    const float xratio = (o2->aabb[1] - o2->aabb[0]) * f->inv_w;
    if (xratio > 1.5f)
      needFurtherPasses = 1;
    else {
      const float zratio = (o2->aabb[5] - o2->aabb[4]) * f->inv_d;
      if (zratio > 1.5f) needFurtherPasses = 1;
    }
  }

  /* (f) :1306-1460 */
  unsigned numTri = 0;
  hf_triangle* T = S->tris;
  const unsigned maxX_local = (unsigned)(maxX - minX);
  const unsigned maxZ_local = (unsigned)(maxZ - minZ);
  for (unsigned x_local = 0; x_local < maxX_local; x_local++) {
    hf_vertex* Row = V + (size_t)x_local * numZ;
    hf_vertex* NextRow = V + (size_t)(x_local + 1) * numZ;
    hf_vertex *A, *B, *C, *D;
    C = &Row[0];
    D = &NextRow[0];
    for (unsigned z_local = 0; z_local < maxZ_local; z_local++) {
      A = C;
      B = D;
      C = &Row[z_local + 1];
      D = &NextRow[z_local + 1];
      const float AHeight = A->vertex[1];
      const float BHeight = B->vertex[1];
      const float CHeight = C->vertex[1];
      const float DHeight = D->vertex[1];
      const int isAfinite = isfinite(AHeight);
      const int isBfinite = isfinite(BHeight);
      const int isCfinite = isfinite(CHeight);
      const int isDfinite = isfinite(DHeight);
      const int isACollide = (AHeight > minO2Height) && isAfinite;
      const int isBCollide = (BHeight > minO2Height) && isBfinite;
      const int isCCollide = (CHeight > minO2Height) && isCfinite;
      const int isDCollide = (DHeight > minO2Height) && isDfinite;
      A->state = !isACollide;
      B->state = !isBCollide;
      C->state = !isCCollide;
      D->state = !isDCollide;

      if ((isACollide || isBCollide || isCCollide) && (isAfinite && isBfinite && isCfinite)) {
        hf_vertex* vs[3] = {A, B, C};
        for (int k = 0; k < 3; k++) {
          if (!vs[k]->state) {
            const float depth =
                box_point_depth(o2, vs[k]->vertex[0], vs[k]->vertex[1], vs[k]->vertex[2]);
            if (depth > dEpsilon) {
              *exit_code = ARTP_EXIT_VERTEX;
              return 1;
            }
            vs[k]->state = 1;
          }
        }
        hf_triangle* t = &T[numTri++];
        t->state = 0;
        t->vertices[0] = A;
        t->vertices[1] = B;
        t->vertices[2] = C;
        t->max_aaab = A->vertex[1] > B->vertex[1] ? A->vertex[1] : B->vertex[1];
        t->max_aaab = C->vertex[1] > t->max_aaab ? C->vertex[1] : t->max_aaab;
        t->is_up = 1;
      }
      if ((isBCollide || isCCollide || isDCollide) && (isBfinite && isCfinite && isDfinite)) {
        hf_vertex* vs[3] = {B, C, D};
        for (int k = 0; k < 3; k++) {
          if (!vs[k]->state) {
            const float depth =
                box_point_depth(o2, vs[k]->vertex[0], vs[k]->vertex[1], vs[k]->vertex[2]);
            if (depth > dEpsilon) {
              *exit_code = ARTP_EXIT_VERTEX;
              return 1;
            }
            vs[k]->state = 1;
          }
        }
        hf_triangle* t = &T[numTri++];
        t->state = 0;
        t->vertices[0] = D;
        t->vertices[1] = B;
        t->vertices[2] = C;
        t->max_aaab = D->vertex[1] > B->vertex[1] ? D->vertex[1] : B->vertex[1];
        t->max_aaab = C->vertex[1] > t->max_aaab ? C->vertex[1] : t->max_aaab;
        t->is_up = 0;
      }
      if (needFurtherPasses && (isBCollide || isCCollide) &&
          (AHeight > CHeight && AHeight > BHeight && DHeight > CHeight && DHeight > BHeight)) {
        B->state = 1;
        C->state = 1;
      }
    }
  }

  g_last_num_tri = numTri;
  /* (g) pass 1: triangles as planes, :1470-1646 */
  for (unsigned k = 0; k < numTri; k++) {
    hf_triangle* t = &T[k];
    float Edge1[3], Edge2[3], triplane[4];
    for (int c = 0; c < 3; c++) {
      Edge1[c] = t->vertices[2]->vertex[c] - t->vertices[0]->vertex[c];
      Edge2[c] = t->vertices[1]->vertex[c] - t->vertices[0]->vertex[c];
    }
    if (t->is_up)
      cross3(triplane, Edge1, Edge2);
    else
      cross3(triplane, Edge2, Edge1);
    const float dinvlength =
        1.0f / sqrtf(triplane[0] * triplane[0] + triplane[1] * triplane[1] + triplane[2] * triplane[2]);
    triplane[0] *= dinvlength;
    triplane[1] *= dinvlength;
    triplane[2] *= dinvlength;
    triplane[3] = dot3(triplane, t->vertices[0]->vertex, 1, 1);
    t->plane_def[0] = triplane[0];
    t->plane_def[1] = triplane[1];
    t->plane_def[2] = triplane[2];
    t->plane_def[3] = triplane[3];
  }

  unsigned numPlanes = 0;
  int n_members = 0;
  for (unsigned k = 0; k < numTri; k++) { /* greedy grouping, :1511-1556 */
    hf_triangle* base = &T[k];
    if (base->state) continue;
    hf_plane* pl = &S->planes[numPlanes];
    pl->first = n_members;
    pl->count = 0;
    S->members[n_members++] = (int)k;
    pl->count++;
    pl->plane_def[0] = base->plane_def[0];
    pl->plane_def[1] = base->plane_def[1];
    pl->plane_def[2] = base->plane_def[2];
    pl->plane_def[3] = base->plane_def[3];
    const float normx = base->plane_def[0];
    const float normy = base->plane_def[1];
    const float normz = base->plane_def[2];
    const float dist = base->plane_def[3];
    for (unsigned m = k + 1; m < numTri; m++) {
      hf_triangle* tt = &T[m];
      if (tt->state) continue;
      if (fabsf(normy - tt->plane_def[1]) < dEpsilon && fabsf(dist - tt->plane_def[3]) < dEpsilon &&
          fabsf(normx - tt->plane_def[0]) < dEpsilon && fabsf(normz - tt->plane_def[2]) < dEpsilon) {
        S->members[n_members++] = (int)m;
        pl->count++;
        tt->state = 1;
      }
    }
    base->state = 1;
    /* HeightFieldPlane::setMinMax, heightfield.h:155-168 */
    pl->max_aaab = T[S->members[pl->first]].max_aaab;
    for (int q = 1; q < pl->count; q++)
      if (T[S->members[pl->first + q]].max_aaab > pl->max_aaab)
        pl->max_aaab = T[S->members[pl->first + q]].max_aaab;
    S->plane_order[numPlanes] = (int)numPlanes;
    numPlanes++;
  }

  if (numPlanes) { /* sortPlanes, :933-955 (bubble sort, swap when A.max - B.max > eps) */
    int has_swapped = 1;
    do {
      has_swapped = 0;
      for (unsigned i = 0; i + 1 < numPlanes; i++) {
        if ((S->planes[S->plane_order[i]].max_aaab - S->planes[S->plane_order[i + 1]].max_aaab) >
            dEpsilon) {
          const int tmp = S->plane_order[i];
          S->plane_order[i] = S->plane_order[i + 1];
          S->plane_order[i + 1] = tmp;
          has_swapped = 1;
        }
      }
    } while (has_swapped);
  }

  for (unsigned k = 0; k < numPlanes; k++) { /* :1581-1645 */
    const hf_plane* pl = &S->planes[S->plane_order[k]];
    float cpos[4][3];
    /* planeTestFlags asks for HEIGHTFIELDMAXCONTACTPERCELL (=10, heightfield.h:36) contacts;
     * dCollideBoxPlane caps at 4. */
    const int numPlaneContacts = box_plane_contacts(o2, pl->plane_def, 10, cpos);
    for (int i = 0; i < numPlaneContacts; i++) {
      for (int b = 0; b < pl->count; b++) {
        const hf_triangle* t = &T[S->members[pl->first + b]];
        if (is_on_heightfield2(f, t->vertices[0], cpos[i], t->is_up)) {
          /* numTerrainContacts == numMaxContactsPossible (1) -> return, :1610-1611 */
          *exit_code = ARTP_EXIT_PLANE;
          return 1;
        }
      }
    }
    /* !didCollide: flag the group's triangles as not collided, :1640-1643 */
    for (int b = 0; b < pl->count; b++) T[S->members[pl->first + b]].state = 0;
  }

  /* (h) pass 2: triangle vertices, :1651-1719 */
  if (needFurtherPasses) {
    for (unsigned k = 0; k < numTri; k++) {
      const hf_triangle* t = &T[k];
      if (t->state) continue;
      for (int i = 0; i < 3; i++) {
        hf_vertex* v = t->vertices[i];
        if (v->state) continue;
        const float depth = box_point_depth(o2, v->vertex[0], v->vertex[1], v->vertex[2]);
        if (depth > dEpsilon) {
          *exit_code = ARTP_EXIT_VERTEX2;
          return 1;
        }
      }
    }
  }
  *exit_code = ARTP_EXIT_NONE;
  return 0;
}

/* ------------------------------------------------------------------------------------------------
 * R3: HeightMapBoxChecker::checkCollision -> dBodySetPosition/Rotation -> dCollide ->
 *     dCollideHeightfield (heightfield.cpp:1791-1964)
 * ---------------------------------------------------------------------------------------------- */
int artp_oracle_check_box(const artp_oracle_field* f, const float side[3], const float pose[16],
                          int* exit_code, int window[4]) {
  box_t box;
  int ec = ARTP_EXIT_AABB_OFF;
  float Rw[12];
  float pw[3];
  /* dBodySetPosition / dBodySetRotation, ode/ode/src/ode.cpp:345-374 */
  pw[0] = pose[0];
  pw[1] = pose[1];
  pw[2] = pose[2];
  memcpy(Rw, pose + 4, sizeof(Rw));
  orthogonalize_R(Rw);

  /* Transform o2 into heightfield space, heightfield.cpp:1838-1853 */
  float pos0[3];
  pos0[0] = pw[0] - f->pos[0];
  pos0[1] = pw[1] - f->pos[1];
  pos0[2] = pw[2] - f->pos[2];
  /* dMultiply1_331(pos1, Rt, pos0) */
  box.pos[0] = dot3(f->R + 0, pos0, 4, 1);
  box.pos[1] = dot3(f->R + 1, pos0, 4, 1);
  box.pos[2] = dot3(f->R + 2, pos0, 4, 1);
  /* dMultiply1_333(R1, Rt, R): R1[4i+j] = sum_k R[4k+j]*Rt[4k+i], odemath.h:376-381 */
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) box.R[4 * i + j] = dot3(Rw + j, f->R + i, 4, 4);
  box.R[3] = box.R[7] = box.R[11] = 0;
  box.pos[0] += f->half_w;
  box.pos[2] += f->half_d;
  box.side[0] = side[0];
  box.side[1] = side[1];
  box.side[2] = side[2];
  box_compute_aabb(&box);

  if (window) window[0] = window[1] = window[2] = window[3] = 0;
  int ret = 0;
  /* :1868-1877 */
  if (!(box.aabb[0] > f->width || box.aabb[4] > f->depth) && !(box.aabb[1] < 0 || box.aabb[5] < 0)) {
    /* :1880-1893 */
    int nMinX = (int)floorf(nextafterf(box.aabb[0] * f->inv_w, -INFINITY));
    int nMaxX = (int)ceilf(nextafterf(box.aabb[1] * f->inv_w, INFINITY));
    int nMinZ = (int)floorf(nextafterf(box.aabb[4] * f->inv_d, -INFINITY));
    int nMaxZ = (int)ceilf(nextafterf(box.aabb[5] * f->inv_d, INFINITY));
    nMinX = nMinX > 0 ? nMinX : 0;
    nMaxX = nMaxX > f->nW - 1 ? f->nW - 1 : nMaxX;
    nMinZ = nMinZ > 0 ? nMinZ : 0;
    nMaxZ = nMaxZ > f->nD - 1 ? f->nD - 1 : nMaxZ;
    if (window) {
      window[0] = nMinX;
      window[1] = nMaxX;
      window[2] = nMinZ;
      window[3] = nMaxZ;
    }
    ret = collide_zone(f, nMinX, nMaxX, nMinZ, nMaxZ, &box, &ec);
  }
  if (exit_code) *exit_code = ec;
  return ret ? 1 : 0;
}

int artp_oracle_check_boxes(const artp_oracle_field* f, const float side[3], const float* poses,
                            size_t n, uint8_t* hit, uint8_t* exit_codes, uint32_t* n_vertices) {
  int count = 0;
  for (size_t i = 0; i < n; i++) {
    int ec, win[4];
    const int h = artp_oracle_check_box(f, side, poses + 16 * i, &ec, win);
    if (hit) hit[i] = (uint8_t)h;
    if (exit_codes) exit_codes[i] = (uint8_t)ec;
    if (n_vertices)
      n_vertices[i] = ec == ARTP_EXIT_AABB_OFF
                          ? 0u
                          : (uint32_t)((win[1] - win[0] + 1) * (win[3] - win[2] + 1));
    count += h;
  }
  return count;
}

/* ------------------------------------------------------------------------------------------------
 * R1 + R2: StateValidityChecker::isValid and the body / feet checkers
 * ---------------------------------------------------------------------------------------------- */
void artp_oracle_robot_defaults(artp_oracle_robot* r) { /* art_planner/include/art_planner/params.h:91-119 */
  r->torso_length = 1.05;
  r->torso_width = 0.55;
  r->torso_height = 0.2;
  r->torso_off_x = 0.0;
  r->torso_off_y = 0.0;
  r->torso_off_z = 0.0;
  r->feet_off_x = 0.362;
  r->feet_off_y = 0.225;
  r->feet_off_z = -0.525;
  r->reach_x = 0.25;
  r->reach_y = 0.1;
  r->reach_z = 0.15;
  r->unknown_space_untraversable = 1;
  r->max_pitch_pert = 10.0 / 180 * M_PI; /* params.h:80-81 */
  r->max_roll_pert = 3.33 / 180 * M_PI;
}

void artp_oracle_robot_yaml(artp_oracle_robot* r) { /* art_planner_ros/config/params.yaml:55-71 */
  r->torso_length = 1.31;
  r->torso_width = 0.65;
  r->torso_height = 0.3;
  r->torso_off_x = 0.0;
  r->torso_off_y = 0.0;
  r->torso_off_z = 0.04;
  r->feet_off_x = 0.51;
  r->feet_off_y = 0.2;
  r->feet_off_z = -0.475;
  r->reach_x = 0.2;
  r->reach_y = 0.2;
  r->reach_z = 0.2;
  r->unknown_space_untraversable = 1;
  /* params.yaml:44-45 in degrees, converted by art_planner_ros/src/utils.cpp:205-207 */
  r->max_pitch_pert = 10 * M_PI / 180;
  r->max_roll_pert = 3.33 * M_PI / 180;
}

/* grid_map::GridMap::isInside -> checkIfPositionWithinMap (grid_map_core, not in /root/reference):
 * t = -(p - c - L/2); inside iff 0 <= t < L on both axes, in double. */
static int map_is_inside(const artp_oracle_map* m, double px, double py) {
  const double tx = -((px - m->pos_x) - 0.5 * m->len_x);
  const double ty = -((py - m->pos_y) - 0.5 * m->len_y);
  return tx >= 0.0 && ty >= 0.0 && tx < m->len_x && ty < m->len_y;
}

/* Pose3FromSE3, art_planner/include/art_planner/utils.h:25-38: double state -> float pose with
 * Eigen::Quaternionf(w,x,y,z).toRotationMatrix() (Eigen/src/Geometry/Quaternion.h). */
static void pose3_from_se3(const double se3[7], float t[3], float R[9]) {
  t[0] = (float)se3[0];
  t[1] = (float)se3[1];
  t[2] = (float)se3[2];
  const float x = (float)se3[3], y = (float)se3[4], z = (float)se3[5], w = (float)se3[6];
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

/* pose * Pose3FromXYZ(ox,oy,oz): Eigen Affine*Affine = (L_l*L_r, L_l*t_r + t_l).  L_r = I so the
 * linear part is unchanged; the 3-term dot is Eigen's unrolled redux x0 + (x1 + x2). */
static void pose_times_xyz(const float t[3], const float R[9], float ox, float oy, float oz,
                           float out_t[3]) {
  for (int i = 0; i < 3; i++)
    out_t[i] = (R[3 * i + 0] * ox + (R[3 * i + 1] * oy + R[3 * i + 2] * oz)) + t[i];
}

static void fill_dpose(const float t[3], const float R[9], float pose[16]) {
  /* validity_checker_body.cpp:36-40 / validity_checker_feet.cpp:41-45: static d_pose keeps its
   * initial pads (origin[3] = 0, rotation[3,7,11] = 0). */
  pose[0] = t[0];
  pose[1] = t[1];
  pose[2] = t[2];
  pose[3] = 0;
  for (int i = 0; i < 3; i++) {
    pose[4 + 4 * i + 0] = R[3 * i + 0];
    pose[4 + 4 * i + 1] = R[3 * i + 1];
    pose[4 + 4 * i + 2] = R[3 * i + 2];
    pose[4 + 4 * i + 3] = 0;
  }
}

void artp_oracle_state_poses(const artp_oracle_map* m, const artp_oracle_robot* r,
                             const double se3[7], float poses[5][16], int inside[5]) {
  float t[3], R[9], tb[3];
  pose3_from_se3(se3, t, R);
  /* validity_checker.cpp:40-43 */
  pose_times_xyz(t, R, (float)r->torso_off_x, (float)r->torso_off_y,
                 (float)(r->torso_off_z - r->feet_off_z), tb);
  fill_dpose(tb, R, poses[0]);
  inside[0] = map_is_inside(m, (double)tb[0], (double)tb[1]);
  /* validity_checker_feet.cpp:64-68, order (+,+),(+,-),(-,+),(-,-) */
  const float fx = (float)r->feet_off_x, fy = (float)r->feet_off_y;
  const float sx[4] = {fx, fx, -fx, -fx};
  const float sy[4] = {fy, -fy, fy, -fy};
  for (int k = 0; k < 4; k++) {
    float tf[3];
    pose_times_xyz(t, R, sx[k], sy[k], 0.0f, tf);
    fill_dpose(tf, R, poses[1 + k]);
    inside[1 + k] = map_is_inside(m, (double)tf[0], (double)tf[1]);
  }
}

static int state_valid_impl(const artp_oracle_map* m, const artp_oracle_robot* r,
                            const double se3[7], int* detail, uint64_t* alg_vertices) {
  float poses[5][16];
  int inside[5];
  int d[6] = {-2, -2, -2, -2, -2, 0};
  artp_oracle_state_poses(m, r, se3, poses, inside);
  const float torso[3] = {(float)r->torso_length, (float)r->torso_width, (float)r->torso_height};
  const float foot[3] = {(float)r->reach_x, (float)r->reach_y, (float)r->reach_z};
  uint64_t verts = 0;
  if (alg_vertices) {
    /* algorithmic bytes: window vertices of all 5 boxes, no credit for early-outs or
     * short-circuiting, 0 for a box whose centre is outside the map. */
    for (int k = 0; k < 5; k++) {
      if (!inside[k]) continue;
      int ec, win[4];
      artp_oracle_check_box(k == 0 ? &m->body : &m->feet, k == 0 ? torso : foot, poses[k], &ec, win);
      if (ec != ARTP_EXIT_AABB_OFF) verts += (uint64_t)((win[1] - win[0] + 1) * (win[3] - win[2] + 1));
    }
    *alg_vertices = verts;
  }
  int valid;
  /* ValidityCheckerBody::isValid, validity_checker_body.cpp:27-42 */
  int body_ok;
  if (!inside[0]) {
    body_ok = 1;
    d[0] = -1;
  } else {
    int ec;
    body_ok = !artp_oracle_check_box(&m->body, torso, poses[0], &ec, NULL);
    d[0] = ec;
    d[5] |= 1;
  }
  valid = body_ok;
  if (valid) {
    /* ValidityCheckerFeet::isValid / boxesAreValidAtPoses, validity_checker_feet.cpp:32-70 */
    for (int k = 0; k < 4; k++) {
      int ok;
      if (!inside[1 + k]) {
        ok = !r->unknown_space_untraversable;
        d[1 + k] = -1;
      } else {
        int ec;
        ok = artp_oracle_check_box(&m->feet, foot, poses[1 + k], &ec, NULL);
        d[1 + k] = ec;
        d[5] |= 2 << k;
      }
      valid &= ok;
      if (!valid) break;
    }
  }
  if (detail) memcpy(detail, d, sizeof(d));
  return valid;
}

int artp_oracle_state_valid(const artp_oracle_map* m, const artp_oracle_robot* r,
                            const double se3[7], int* detail) {
  return state_valid_impl(m, r, se3, detail, NULL);
}

void artp_oracle_states_valid(const artp_oracle_map* m, const artp_oracle_robot* r,
                              const double* se3, size_t n, uint8_t* valid, uint64_t* alg_vertices) {
  uint64_t total = 0;
  for (size_t i = 0; i < n; i++) {
    uint64_t v = 0;
    valid[i] = (uint8_t)state_valid_impl(m, r, se3 + 7 * i, NULL, alg_vertices ? &v : NULL);
    total += v;
  }
  if (alg_vertices) *alg_vertices = total;
}

/* ------------------------------------------------------------------------------------------------
 * R6: SE3FromSE2Sampler::sampleUniform (art_planner/src/sampler.cpp:56-131)
 * ---------------------------------------------------------------------------------------------- */
/* Counter-based replacement for ompl::RNG::uniform01 (std::mt19937 stream; OMPL is not in
 * /root/reference): splitmix64 finaliser over (seed, index, k). 53-bit mantissa -> [0,1). */
double artp_oracle_uniform01(uint64_t seed, uint64_t index, unsigned k) {
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

void artp_oracle_sample(const artp_oracle_sampler_map* m, const artp_oracle_robot* r, uint64_t seed,
                        uint64_t index, double se3[7], int rowcol[2]) {
  double px, py;
  if (!m->sample_uniform) {
    /* samplePositionInMapFromDist, sampler.cpp:56-78 */
    const double samp_col = artp_oracle_uniform01(seed, index, 0);
    const double samp_row = artp_oracle_uniform01(seed, index, 1);
    int row, col;
    for (row = 0; row < m->rows - 1; ++row)
      if (m->cum_prob_rowwise[row] > samp_row) break; /* float promoted to double */
    for (col = 0; col < m->cols - 1; ++col)
      if (m->cum_prob[(size_t)row + (size_t)col * m->rows] > samp_col) break;
    /* grid_map getPosition: c + (L/2 - res/2) - res*i  (grid_map_core GridMapMath.cpp) */
    px = (m->pos_x + (0.5 * m->len_x - 0.5 * m->res)) + m->res * (double)(-row);
    py = (m->pos_y + (0.5 * m->len_y - 0.5 * m->res)) + m->res * (double)(-col);
  } else {
    /* samplePositionInMap, sampler.cpp:38-50: RealVectorStateSampler::sampleUniform over the SE3 bounds
     * (pos -+ length, planner.cpp:146-156; three uniformReal draws per attempt, z unused) until
     * map_->isInside(pos).  Attempt a uses the draws k = 8 + 3a .. 8 + 3a + 2. */
    const double lx = m->pos_x - m->len_x, hx = m->pos_x + m->len_x;
    const double ly = m->pos_y - m->len_y, hy = m->pos_y + m->len_y;
    unsigned a = 0;
    for (;; ++a) {
      px = (hx - lx) * artp_oracle_uniform01(seed, index, 8 + 3 * a) + lx;
      py = (hy - ly) * artp_oracle_uniform01(seed, index, 8 + 3 * a + 1) + ly;
      const double tx = -((px - m->pos_x) - 0.5 * m->len_x), ty = -((py - m->pos_y) - 0.5 * m->len_y);
      if ((tx >= 0.0 && ty >= 0.0 && tx < m->len_x && ty < m->len_y) || a >= 255) break;
    }
  }
  /* getIndexOfPosition (sampler.cpp:95): i = (int)(-((p - L/2 - c)/res)); equals (row,col). */
  const int ri = (int)(-(((px - 0.5 * m->len_x) - m->pos_x) / m->res));
  const int ci = (int)(-(((py - 0.5 * m->len_y) - m->pos_y) / m->res));
  const size_t ind = (size_t)ri + (size_t)ci * m->rows;
  if (rowcol) {
    rowcol[0] = ri;
    rowcol[1] = ci;
  }
  double v[3];
  v[0] = px;
  v[1] = py;
  v[2] = (double)m->elevation[ind]; /* sampler.cpp:99 */
  const double nwx = (double)m->normal_x[ind];
  const double nwy = (double)m->normal_y[ind];
  const double nwz = (double)m->normal_z[ind];
  const float std = m->plane_fit_std_dev[ind];
  /* sampler.cpp:105: rng_.uniformReal(-1,1) * std::min(std, 0.5f) * reach.z
   * uniformReal(a,b) = (b-a)*u + a (ompl/util/RandomNumbers.h) */
  const double u_pert = artp_oracle_uniform01(seed, index, 2);
  const float std_min = (0.5f < std) ? 0.5f : std; /* std::min(std, 0.5f) */
  const double pert = ((1.0 - (-1.0)) * u_pert + (-1.0)) * (double)std_min * r->reach_z;
  v[0] += nwx * pert;
  v[1] += nwy * pert;
  v[2] += nwz * pert;
  se3[0] = v[0];
  se3[1] = v[1];
  se3[2] = v[2];
  /* rng_.eulerRPY, ompl/util/RandomNumbers.cpp */
  const double pi = 3.14159265358979323846;
  double rpy[3];
  rpy[0] = pi * (-2.0 * artp_oracle_uniform01(seed, index, 3) + 1.0);
  rpy[1] = acos(1.0 - 2.0 * artp_oracle_uniform01(seed, index, 4)) - pi / 2.0;
  rpy[2] = pi * (-2.0 * artp_oracle_uniform01(seed, index, 5) + 1.0);
  /* R_wb = AngleAxis(yaw, Z) as quaternion; normal_b = R_wb.inverse() * normal_w
   * (sampler.cpp:120-123).  Eigen: q = (cos(yaw/2), 0,0, sin(yaw/2)); inverse = conjugate/|q|^2;
   * q*v: uv = q.vec x v; uv += uv; v + q.w*uv + q.vec x uv. */
  {
    const double ha = 0.5 * rpy[2];
    const double qw = cos(ha), qz = sin(ha);
    const double n2 = qw * qw + qz * qz; /* squaredNorm = x*x+y*y+z*z+w*w with x=y=0 */
    const double iw = qw / n2, ix = -0.0 / n2, iy = -0.0 / n2, iz = -qz / n2;
    double uv[3];
    uv[0] = iy * nwz - iz * nwy;
    uv[1] = iz * nwx - ix * nwz;
    uv[2] = ix * nwy - iy * nwx;
    uv[0] += uv[0];
    uv[1] += uv[1];
    uv[2] += uv[2];
    const double nbx = nwx + iw * uv[0] + (iy * uv[2] - iz * uv[1]);
    const double nby = nwy + iw * uv[1] + (iz * uv[0] - ix * uv[2]);
    const double nbz = nwz + iw * uv[2] + (ix * uv[1] - iy * uv[0]);
    /* sampler.cpp:125-128 */
    rpy[0] = -atan2(nby, nbz) + rpy[0] * r->max_roll_pert / M_PI_2;
    rpy[1] = atan2(nbx, nbz) + rpy[1] * r->max_pitch_pert / M_PI_4;
  }
  /* setSO3FromRPY, utils.h:101-115 */
  {
    const double r2 = rpy[0] * 0.5, p2 = rpy[1] * 0.5, y2 = rpy[2] * 0.5;
    const double cr = cos(r2), cp = cos(p2), cy = cos(y2);
    const double sr = sin(r2), sp = sin(p2), sy = sin(y2);
    se3[6] = cy * cp * cr + sy * sp * sr; /* w */
    se3[3] = cy * cp * sr - sy * sp * cr; /* x */
    se3[4] = sy * cp * sr + cy * sp * cr; /* y */
    se3[5] = sy * cp * cr - cy * sp * sr; /* z */
  }
}

void artp_oracle_samples(const artp_oracle_sampler_map* m, const artp_oracle_robot* r, uint64_t seed,
                         uint64_t first_index, size_t n, double* se3, int* rowcol) {
  for (size_t i = 0; i < n; i++)
    artp_oracle_sample(m, r, seed, first_index + i, se3 + 7 * i, rowcol ? rowcol + 2 * i : NULL);
}

/* ------------------------------------------------------------------------------------------------
 * R7: OMPL 1.4.2 SE3 interpolation / DiscreteMotionValidator (OMPL is not in /root/reference;
 *     restated from ompl/base/spaces/{RealVector,SO3}StateSpace.cpp, DiscreteMotionValidator.cpp)
 * ---------------------------------------------------------------------------------------------- */
#define MAX_QUATERNION_NORM_ERROR 1e-9

static double so3_arc_length(const double* q1, const double* q2) { /* q = x y z w */
  const double dq = fabs(q1[0] * q2[0] + q1[1] * q2[1] + q1[2] * q2[2] + q1[3] * q2[3]);
  if (dq > 1.0 - MAX_QUATERNION_NORM_ERROR) return 0.0;
  return acos(dq);
}

void artp_oracle_interpolate(const double a[7], const double b[7], double t, double out[7]) {
  /* RealVectorStateSpace::interpolate */
  for (int i = 0; i < 3; i++) out[i] = a[i] + (b[i] - a[i]) * t;
  /* SO3StateSpace::interpolate (slerp) */
  const double* q1 = a + 3;
  const double* q2 = b + 3;
  const double theta = so3_arc_length(q1, q2);
  if (theta > DBL_EPSILON) {
    const double d = 1.0 / sin(theta);
    const double s0 = sin((1.0 - t) * theta);
    double s1 = sin(t * theta);
    const double dq = q1[0] * q2[0] + q1[1] * q2[1] + q1[2] * q2[2] + q1[3] * q2[3];
    if (dq < 0) s1 = -s1;
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

unsigned artp_oracle_valid_segment_count(const artp_oracle_map* m, double z_extent,
                                         const double a[7], const double b[7]) {
  /* RealVectorStateSpace: bounds = map centre -/+ FULL length (planner.cpp:146-156);
   * maxExtent = sqrt(sum (high-low)^2); longestValidSegment = 0.01 * maxExtent. */
  const double ex = (m->pos_x + m->len_x) - (m->pos_x - m->len_x);
  const double ey = (m->pos_y + m->len_y) - (m->pos_y - m->len_y);
  double e = 0.0;
  e += ex * ex;
  e += ey * ey;
  e += z_extent * z_extent;
  const double seg_r3 = sqrt(e) * 0.01;
  double d2 = 0.0;
  for (int i = 0; i < 3; i++) {
    const double diff = a[i] - b[i];
    d2 += diff * diff;
  }
  const unsigned n_r3 = (unsigned)ceil(sqrt(d2) / seg_r3);
  /* SO3: maxExtent = pi/2 */
  const double seg_so3 = (0.5 * 3.14159265358979323846) * 0.01;
  const unsigned n_so3 = (unsigned)ceil(so3_arc_length(a + 3, b + 3) / seg_so3);
  return n_r3 > n_so3 ? n_r3 : n_so3;
}

int artp_oracle_check_motion(const artp_oracle_map* m, const artp_oracle_robot* r, double z_extent,
                             const double s1[7], const double s2[7], unsigned* n_checked) {
  unsigned checked = 1;
  int result = 1;
  if (!artp_oracle_state_valid(m, r, s2, NULL)) {
    if (n_checked) *n_checked = checked;
    return 0;
  }
  const int nd = (int)artp_oracle_valid_segment_count(m, z_extent, s1, s2);
  if (nd >= 2) {
    /* bisection queue of DiscreteMotionValidator::checkMotion */
    int (*queue)[2] = (int(*)[2])malloc(sizeof(int[2]) * (size_t)(nd + 2));
    int head = 0, tail = 0;
    queue[tail][0] = 1;
    queue[tail][1] = nd - 1;
    tail++;
    while (head < tail) {
      const int first = queue[head][0], second = queue[head][1];
      const int mid = (first + second) / 2;
      double test[7];
      artp_oracle_interpolate(s1, s2, (double)mid / (double)nd, test);
      checked++;
      if (!artp_oracle_state_valid(m, r, test, NULL)) {
        result = 0;
        break;
      }
      head++;
      if (first < mid) {
        queue[tail][0] = first;
        queue[tail][1] = mid - 1;
        tail++;
      }
      if (second > mid) {
        queue[tail][0] = mid + 1;
        queue[tail][1] = second;
        tail++;
      }
    }
    free(queue);
  }
  if (n_checked) *n_checked = checked;
  return result;
}

/* DiscreteMotionValidator::checkMotion(s1, s2, lastValid) of OMPL 1.4.2 (src/ompl/base/src/
 * DiscreteMotionValidator.cpp, second overload; pure virtual in ob::MotionValidator -- OMPL is not vendored,
 * restated from the published source): the interior states j = 1 .. nd-1 are tested IN ORDER, then s2;
 * lastValid.second = (j - 1) / nd at the first failing j, (nd - 1) / nd when only s2 fails, and
 * *lastValid.first = interpolate(s1, s2, lastValid.second). */
int artp_oracle_check_motion_last_valid(const artp_oracle_map* m, const artp_oracle_robot* r, double z_extent,
                                        const double s1[7], const double s2[7], double* last_valid_t,
                                        double last_valid_state[7]) {
  int result = 1;
  const int nd = (int)artp_oracle_valid_segment_count(m, z_extent, s1, s2);
  if (nd > 1) {
    for (int j = 1; j < nd; ++j) {
      double test[7];
      artp_oracle_interpolate(s1, s2, (double)j / (double)nd, test);
      if (!artp_oracle_state_valid(m, r, test, NULL)) {
        *last_valid_t = (double)(j - 1) / (double)nd;
        if (last_valid_state) artp_oracle_interpolate(s1, s2, *last_valid_t, last_valid_state);
        result = 0;
        break;
      }
    }
  }
  if (result) {
    if (!artp_oracle_state_valid(m, r, s2, NULL)) {
      *last_valid_t = (double)(nd - 1) / (double)nd;
      if (last_valid_state) artp_oracle_interpolate(s1, s2, *last_valid_t, last_valid_state);
      result = 0;
    }
  }
  return result;
}

int artp_oracle_edge_interp_valid(const artp_oracle_map* m, const artp_oracle_robot* r,
                                  const double s1[7], const double s2[7], unsigned* n_interp_out,
                                  double* interior, unsigned max_interior) {
  /* lateralDistance, utils.h:52-61; prm_motion_cost.cpp:340-377 */
  const double dx = s2[0] - s1[0];
  const double dy = s2[1] - s1[1];
  const double dist = sqrt(dx * dx + dy * dy);
  const double kMaxDist = 0.5;
  const unsigned n_interp = (unsigned)(dist / kMaxDist);
  if (n_interp_out) *n_interp_out = n_interp;
  int connection_valid = 1;
  if (n_interp > 0) {
    const double n_interp_div = 1.0 / (n_interp + 1);
    for (unsigned step = 1; step < n_interp + 1; ++step) {
      double st[7];
      artp_oracle_interpolate(s1, s2, step * n_interp_div, st);
      if (interior && step - 1 < max_interior) memcpy(interior + 7 * (step - 1), st, sizeof(st));
      connection_valid &= artp_oracle_state_valid(m, r, st, NULL);
      if (!connection_valid) break;
    }
  }
  return connection_valid;
}
