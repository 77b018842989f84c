// roadmap.h -- "next" row N1: the batched planner front end that turns the state / edge
// kernels into plans.  It follows the reference's PRM front ends
//   PRMMotionCostMaintainer::sampleGraph   art_planner/src/planners/prm_motion_cost.cpp:145-219
//   PRMMotionCost::addValidMilestone       prm_motion_cost.cpp:325-390  (k nearest, 0.5 m interpolation)
//   PRMMotionCost::constructSolution       prm_motion_cost.cpp:536-673  (A*, lazy checkMotion of the path)
//   PathLengthObjective                    art_planner/src/objectives/path_length_objective.cpp:26-70
// as ONE batch per stage instead of one milestone at a time:
//   1. milestones = the first n accepted states of the (seed, index) sample stream
//   2. k nearest neighbours of every vertex under OMPL's SE3 distance (|dp| + SO3 arc), k = the PRM* rule
//      ceil(e (1 + 1/6) ln n) of OMPL's KStarStrategy for a 6-dimensional space
//   3. candidate edges = the symmetrised k-NN pairs, validated by the 0.5 m interpolation rule of
//      addValidMilestone (the edge kernels of R7)
//   4. edge costs: the chain of interpolated sub-edges the reference would have put into its graph
//   5. host: CSR graph, A* (exact heap search), the final path's edges re-checked with the discrete motion
//      validator; an invalid one is removed and the search repeated (LazyPRM's loop)
// The reference inserts milestones one by one (each sees only its predecessors, interpolated states become
// vertices and neighbours themselves), so the batched graph differs by construction; what is kept is every predicate
// (validity, interpolation rule, costs, search).  artp_roadmap_params::construction selects the reference's own
// graphs instead: 1 = PRMMotionCost::addValidMilestone's insertion loop (IncrementalGraph below: sequential on the
// host, one small device batch of chain states per milestone), 2 = LazyPRMStarMinUpdate's predecessor-only direct
// edges (one device batch: knn_kernel with pred_only and a per-vertex k).  Both reproduce the edge SET of
// oracle/prm_incremental.py's literal restatement (tests/test_roadmap.py).  OMPL is not available here: parity for
// this row is unpinned, the tests check the stage results against brute force / the oracle / scipy.
#pragma once

#include <algorithm>
#include <cmath>
#include <queue>
#include <system_error>
#include <thread>
#include <tuple>
#include <vector>

// Tuning / comparison switches of the solve loop ($ARTP_SOLVE_TIMING, $ARTP_LAZY_INFORMED, $ARTP_SOLVE_SEQUENCE, $ARTP_LAZY_ROOT,
// $ARTP_SOLVE_ASTAR): read in the variants build only (make variants, -DARTP_VARIANTS); the product has no such switches.
static inline const char* variant_env(const char* name) {
#ifdef ARTP_VARIANTS
  return std::getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}

namespace artp {

// OMPL CompoundStateSpace::distance for SE3: RealVectorStateSpace L2 + SO3StateSpace::distance (arc length)
__device__ __forceinline__ double se3_distance(const double* a, const double* b) {
  const double dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
  return sqrt(dx * dx + dy * dy + dz * dz) + so3_arc_length(a + 3, b + 3);
}

// ---- k nearest neighbours on a uniform xy grid ----------------------------------------------------------
// Vertices are bucketed by the grid cell of their (x, y) (cell ids sorted with hipcub, vertex rows permuted
// into cell order).  One lane per query walks square rings of cells around its own cell; the k best live in
// LDS (entry e of lane l at [e * 64 + l]: conflict-free).  The SE3 distance is at least the planar
// distance, and every cell of ring r+1 is at least r cell sizes away in the plane, so the walk stops
// -- exactly -- once the list is full and r * h >= the current k-th distance (or the grid is exhausted).
// The R^3 term alone rejects most candidates before the arc length (an acos) is needed.
// Output: neighbours by ascending (distance, original index).
struct KnnGrid {
  double x0, y0, inv_h, h;
  int gx, gy;
};

__device__ __forceinline__ int knn_cell_coord(double v, double v0, double inv_h, int g) {
  int c = (int)floor((v - v0) * inv_h);
  return c < 0 ? 0 : (c >= g ? g - 1 : c);
}

__global__ void __launch_bounds__(256)
knn_cell_ids_kernel(const double* __restrict__ verts, int nv, KnnGrid g, uint32_t* __restrict__ cell,
                    uint32_t* __restrict__ ident) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nv) return;
  cell[i] = (uint32_t)(knn_cell_coord(verts[(size_t)i * 7 + 1], g.y0, g.inv_h, g.gy) * g.gx +
                       knn_cell_coord(verts[(size_t)i * 7 + 0], g.x0, g.inv_h, g.gx));
  ident[i] = (uint32_t)i;
}

// rows of verts in cell order + first sorted position of every cell (cell_start[ncell] = nv)
__global__ void __launch_bounds__(256)
knn_permute_kernel(const double* __restrict__ verts, const uint32_t* __restrict__ sorted_cell,
                   const uint32_t* __restrict__ sorted_id, int nv, int ncell, double* __restrict__ verts_sorted,
                   uint32_t* __restrict__ cell_start) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= nv) return;
  const uint32_t id = sorted_id[p];
#pragma unroll
  for (int c = 0; c < 7; ++c) verts_sorted[(size_t)p * 7 + c] = verts[(size_t)id * 7 + c];
  const uint32_t cc = sorted_cell[p];
  const uint32_t prev = p == 0 ? 0xffffffffu : sorted_cell[p - 1];
  if (p == 0 || prev != cc)
    for (uint32_t c2 = (p == 0 ? 0u : prev + 1u); c2 <= cc; ++c2) cell_start[c2] = (uint32_t)p;
  if (p == nv - 1)
    for (uint32_t c2 = cc + 1; c2 <= (uint32_t)ncell; ++c2) cell_start[c2] = (uint32_t)nv;
}

// k_of (may be NULL) = the list length of every query by ORIGINAL index (<= k_row, the row length of the outputs), and
// pred_only restricts a query's candidates to the vertices with a SMALLER original index: together they are the
// neighbour search of a planner that inserts its vertices one at a time (construction 2, roadmap_connect) --
// vertex i sees its predecessors only, with the k of the graph size at ITS insertion -- as one batch.
__global__ void __launch_bounds__(64)
knn_kernel(const double* __restrict__ verts_sorted, const uint32_t* __restrict__ sorted_id,
           const uint32_t* __restrict__ cell_start, KnnGrid g, int nv, int k_row, const int* __restrict__ k_of,
           int pred_only, uint32_t* __restrict__ out_idx, double* __restrict__ out_dist) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* bd = reinterpret_cast<double*>(smem);                          // [k_row][64]
  uint32_t* bi = reinterpret_cast<uint32_t*>(bd + (size_t)k_row * 64);  // [k_row][64] original indices
  const int lane = threadIdx.x;
  const int p = blockIdx.x * 64 + lane;  // query = sorted position p: neighbouring lanes share cells
  if (p >= nv) return;
  const uint32_t qi = sorted_id[p];
  const int k = k_of ? min(k_of[qi], k_row) : k_row;
  if (k <= 0) {
    for (int a = 0; a < k_row; ++a) {
      out_idx[(size_t)qi * k_row + a] = 0xffffffffu;
      out_dist[(size_t)qi * k_row + a] = INFINITY;
    }
    return;
  }
  double q[7];
#pragma unroll
  for (int c = 0; c < 7; ++c) q[c] = verts_sorted[(size_t)p * 7 + c];
  const int cx = knn_cell_coord(q[0], g.x0, g.inv_h, g.gx), cy = knn_cell_coord(q[1], g.y0, g.inv_h, g.gy);
  int count = 0;
  double thr = INFINITY;  // current k-th best distance once the list is full
  int arg = 0;            // its slot
  const int rmax = max(max(cx, g.gx - 1 - cx), max(cy, g.gy - 1 - cy));
  for (int r = 0; r <= rmax; ++r) {
    if (count == k && (double)(r - 1) * g.h >= thr) break;  // ring r is at least (r-1)*h away
    for (int yy = cy - r; yy <= cy + r; ++yy) {
      if (yy < 0 || yy >= g.gy) continue;
      const bool edge_row = (yy == cy - r) || (yy == cy + r);
      // edge rows: the whole run of cells cx-r .. cx+r (contiguous in the sorted order); inner rows: the two
      // end cells
      for (int part = 0; part < (edge_row || r == 0 ? 1 : 2); ++part) {
        int xa, xb;
        if (edge_row || r == 0) {
          xa = max(cx - r, 0);
          xb = min(cx + r, g.gx - 1);
        } else {
          xa = xb = part == 0 ? cx - r : cx + r;
          if (xa < 0 || xa >= g.gx) continue;
        }
        const uint32_t j0 = cell_start[yy * g.gx + xa], j1 = cell_start[yy * g.gx + xb + 1];
        for (uint32_t j = j0; j < j1; ++j) {
          if ((int)j == p) continue;
          if (pred_only && sorted_id[j] >= qi) continue;
          const double* c = verts_sorted + (size_t)j * 7;
          const double dx = q[0] - c[0], dy = q[1] - c[1], dz = q[2] - c[2];
          const double dp = sqrt(dx * dx + dy * dy + dz * dz);
          if (count == k && !(dp < thr)) continue;
          const double d = dp + so3_arc_length(q + 3, c + 3);
          if (count < k) {
            bd[count * 64 + lane] = d;
            bi[count * 64 + lane] = sorted_id[j];
            ++count;
            if (count == k) {
              thr = -1.0;
              for (int e = 0; e < k; ++e)
                if (bd[e * 64 + lane] > thr) {
                  thr = bd[e * 64 + lane];
                  arg = e;
                }
            }
          } else if (d < thr) {
            bd[arg * 64 + lane] = d;
            bi[arg * 64 + lane] = sorted_id[j];
            thr = -1.0;
            for (int e = 0; e < k; ++e)
              if (bd[e * 64 + lane] > thr) {
                thr = bd[e * 64 + lane];
                arg = e;
              }
          }
        }
      }
    }
  }
  // selection sort by (distance, index); unused slots (fewer than k candidates) are marked
  const size_t i = qi;
  for (int a = 0; a < k_row; ++a) {
    if (a >= count) {
      out_idx[i * k_row + a] = 0xffffffffu;
      out_dist[i * k_row + a] = INFINITY;
      continue;
    }
    int best = a;
    for (int e = a + 1; e < count; ++e) {
      const double de = bd[e * 64 + lane], db = bd[best * 64 + lane];
      if (de < db || (de == db && bi[e * 64 + lane] < bi[best * 64 + lane])) best = e;
    }
    const double dbest = bd[best * 64 + lane];
    const uint32_t ibest = bi[best * 64 + lane];
    bd[best * 64 + lane] = bd[a * 64 + lane];
    bi[best * 64 + lane] = bi[a * 64 + lane];
    bd[a * 64 + lane] = dbest;
    bi[a * 64 + lane] = ibest;
    out_idx[i * k_row + a] = ibest;
    out_dist[i * k_row + a] = dbest;
  }
}

// undirected candidate edge of (i, j in knn(i)) as the key min << 32 | max
__global__ void __launch_bounds__(256)
knn_edge_keys_kernel(const uint32_t* __restrict__ knn, int nv, int k, unsigned long long* __restrict__ keys) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)nv * k) return;
  const uint32_t i = (uint32_t)(t / k), j = knn[t];
  unsigned long long key = ~0ull;  // sorts last, dropped
  if (j != 0xffffffffu) key = ((unsigned long long)(i < j ? i : j) << 32) | (unsigned long long)(i < j ? j : i);
  keys[t] = key;
}

__global__ void __launch_bounds__(256)
gather_edge_states_kernel(const double* __restrict__ verts, const unsigned long long* __restrict__ keys, size_t ne,
                          double* __restrict__ s1, double* __restrict__ s2, int flip) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= ne) return;
  // flip: the edge's cost direction is larger id -> smaller id (construction 2: new vertex -> predecessor)
  const uint32_t a = (uint32_t)(keys[e] >> 32), b = (uint32_t)(keys[e] & 0xffffffffu);
  const uint32_t u = flip ? b : a, v = flip ? a : b;
#pragma unroll
  for (int c = 0; c < 7; ++c) {
    s1[e * 7 + c] = verts[(size_t)u * 7 + c];
    s2[e * 7 + c] = verts[(size_t)v * 7 + c];
  }
}

__global__ void __launch_bounds__(256)
gather_edge_states_uv_kernel(const double* __restrict__ verts, const uint32_t* __restrict__ eu,
                             const uint32_t* __restrict__ ev, size_t ne, double* __restrict__ s1,
                             double* __restrict__ s2) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= ne) return;
  const uint32_t u = eu[e], v = ev[e];
#pragma unroll
  for (int c = 0; c < 7; ++c) {
    s1[e * 7 + c] = verts[(size_t)u * 7 + c];
    s2[e * 7 + c] = verts[(size_t)v * 7 + c];
  }
}

struct PathLengthParams {  // Params::objectives.custom_path_length (params.h:69-73)
  int directional;
  double max_lon_vel, max_lat_vel, max_ang_vel;
};

// getYawFromSO3 (utils.h:78-86): Scalar = float result of the double atan2
__device__ __forceinline__ double yaw_from_quat(const double* q) {  // q = x y z w
  return (double)(float)atan2(2.0 * (q[3] * q[2] + q[0] * q[1]), 1.0 - 2.0 * (q[1] * q[1] + q[2] * q[2]));
}

// PathLengthObjective::motionCost / motionCostHeuristic (path_length_objective.cpp:26-70)
__device__ __forceinline__ double path_length_cost(const PathLengthParams& p, const double* a, const double* b) {
  const double x_dif = b[0] - a[0], y_dif = b[1] - a[1], z_dif = b[2] - a[2];
  if (!p.directional) return sqrt(x_dif * x_dif + y_dif * y_dif + z_dif * z_dif) / p.max_lon_vel;
  const double yaw1 = yaw_from_quat(a + 3), yaw2 = yaw_from_quat(b + 3);
  const double d = fabs(yaw1 - yaw2);
  const double yaw_dif = (d > 3.14159265358979323846) ? 2.0 * 3.14159265358979323846 - d : d;
  const double lon_dif = cos(yaw1) * x_dif + sin(yaw1) * y_dif;
  const double lat_dif = -sin(yaw1) * x_dif + cos(yaw1) * y_dif;
  const double t_yaw = fabs(yaw_dif) / p.max_ang_vel;
  const double t_lon = fabs(lon_dif) / p.max_lon_vel;
  const double t_lat = fabs(lat_dif) / p.max_lat_vel;
  return fmax(fmax(t_lon, t_lat), t_yaw);
}

// Cost of a roadmap edge = the sum over the chain the reference's addValidMilestone builds
// (prm_motion_cost.cpp:340-377): n_interp interior states at step / (n_interp + 1), sub-edges in order.
__global__ void __launch_bounds__(256)
edge_chain_cost_kernel(PathLengthParams p, const double* __restrict__ s1, const double* __restrict__ s2,
                       const uint32_t* __restrict__ n_interp, size_t ne, double* __restrict__ cost) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= ne) return;
  double a[7], b[7], prev[7], cur[7];
#pragma unroll
  for (int c = 0; c < 7; ++c) {
    a[c] = s1[e * 7 + c];
    b[c] = s2[e * 7 + c];
    prev[c] = a[c];
  }
  const unsigned ni = n_interp[e];
  const double div = 1.0 / (double)(ni + 1);
  double total = 0.0;
  for (unsigned step = 1; step <= ni; ++step) {
    se3_interpolate(a, b, (double)step * div, cur);
    total += path_length_cost(p, prev, cur);
#pragma unroll
    for (int c = 0; c < 7; ++c) prev[c] = cur[c];
  }
  total += path_length_cost(p, prev, b);
  cost[e] = total;
}

// Learned-cost objective (PRMMotionCostMaintainer::updateEdges, prm_motion_cost.cpp:27-73): every sub-edge of
// a chain is one row of the EdgeMatrix -- target (x, y, yaw) then start (x, y, yaw), as floats.
__global__ void __launch_bounds__(256)
chain_edge_matrix_kernel(const double* __restrict__ s1, const double* __restrict__ s2,
                         const uint32_t* __restrict__ n_interp, const uint32_t* __restrict__ row_off, size_t ne,
                         float* __restrict__ edge_matrix) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= ne) return;
  double a[7], b[7], prev[7], cur[7];
#pragma unroll
  for (int c = 0; c < 7; ++c) {
    a[c] = s1[e * 7 + c];
    b[c] = s2[e * 7 + c];
    prev[c] = a[c];
  }
  const unsigned ni = n_interp[e];
  const double div = 1.0 / (double)(ni + 1);
  float* row = edge_matrix + (size_t)row_off[e] * 6;
  for (unsigned step = 1; step <= ni + 1; ++step) {
    if (step <= ni) {
      se3_interpolate(a, b, (double)step * div, cur);
    } else {
#pragma unroll
      for (int c = 0; c < 7; ++c) cur[c] = b[c];
    }
    row[0] = (float)cur[0];
    row[1] = (float)cur[1];
    row[2] = (float)yaw_from_quat(cur + 3);
    row[3] = (float)prev[0];
    row[4] = (float)prev[1];
    row[5] = (float)yaw_from_quat(prev + 3);
    row += 6;
#pragma unroll
    for (int c = 0; c < 7; ++c) prev[c] = cur[c];
  }
}

// MotionCostObjective::motionCost's own split of a motion (motion_cost_objective.cpp:41-42):
// n_interp = (unsigned)(lateralDistance / max_query_edge_length)
__global__ void __launch_bounds__(256)
motion_cost_interp_kernel(const double* __restrict__ s1, const double* __restrict__ s2, size_t ne, double step,
                          uint32_t* __restrict__ n_interp) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= ne) return;
  const double dx = s2[e * 7] - s1[e * 7], dy = s2[e * 7 + 1] - s1[e * 7 + 1];
  const double q = sqrt(dx * dx + dy * dy) / step;
  n_interp[e] = q >= 0.0 && q < 4194304.0 ? (uint32_t)q : 0u;
}

__global__ void __launch_bounds__(256)
chain_rows_kernel(const uint32_t* __restrict__ n_interp, size_t ne, uint32_t* __restrict__ rows) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < ne) rows[e] = n_interp[e] + 1;
  if (e == ne) rows[e] = 0;
}

// MotionCostObjective::getCost / isFeasible (motion_cost_objective.h:54-66) summed over the chain; one
// infeasible sub-edge (risk above the threshold) makes the connection unusable (infinite cost).
__global__ void __launch_bounds__(256)
chain_motion_cost_kernel(const float* __restrict__ cost3, const uint32_t* __restrict__ row_off,
                         const uint32_t* __restrict__ n_interp, size_t ne, float w_energy, float w_time,
                         float w_risk, float risk_threshold, double* __restrict__ cost) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= ne) return;
  const float* c = cost3 + (size_t)row_off[e] * 3;
  double total = 0.0;
  bool feasible = true;
  for (unsigned s = 0; s <= n_interp[e]; ++s, c += 3) {
    const double en = c[0], ti = c[1], ri = c[2];
    feasible = feasible && (ri <= (double)risk_threshold);
    total += en * w_energy + ti * w_time + ri * w_risk;
  }
  cost[e] = feasible ? total : INFINITY;
}

// construction 1: the interior states of the 0.5 m chains of ONE milestone (tasks: a[7], b[7], t per row) -- the states
// themselves are needed on the host, where the valid ones become graph vertices (prm_motion_cost.cpp:353-366)
__global__ void __launch_bounds__(64)
chain_states_kernel(const double* __restrict__ tasks, int n, double* __restrict__ out) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  const double* t = tasks + (size_t)i * 15;
  double st[7];
  se3_interpolate(t, t + 7, t[14], st);
#pragma unroll
  for (int c = 0; c < 7; ++c) out[(size_t)i * 7 + c] = st[c];
}

}  // namespace artp

// -------------------------------------------------------------------------------------------------------
namespace artp {
// Label-correcting single-source shortest paths over the undirected edge list (one lane per edge, both
// directions): the search of constructSolution (boost::astar_search, prm_motion_cost.cpp:536-620) for roadmaps
// whose host A* would take tens of milliseconds.  Distances are non-negative doubles, so their bit patterns order
// like unsigned integers and atomicMin applies.  `rounds` sweeps per launch; *changed counts successful
// relaxations.  At the fixed point dist[] is exact (every sum is ONE double addition dist[u] + w, the same value the
// sequential search forms), and every reached vertex has an edge with dist[u] + w == dist[v] bit for bit.
__global__ void __launch_bounds__(256)
sssp_relax_kernel(const uint32_t* __restrict__ eu, const uint32_t* __restrict__ ev, const double* __restrict__ w,
                  size_t ne, unsigned long long* __restrict__ dist, unsigned* __restrict__ stamp, unsigned sweep,
                  unsigned* __restrict__ changed) {
  // stamp[v] = the last sweep that lowered dist[v].  An edge only has work when one of its ends moved in the
  // previous sweep (every edge of a vertex is relaxed in the sweep after each of its changes), so the sweeps
  // behind the wavefront cost two 4-byte reads per edge instead of two gathers and an atomic.
  unsigned local = 0;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < ne; e += (size_t)gridDim.x * blockDim.x) {
    const uint32_t u = eu[e], v = ev[e];
    if (stamp[u] + 1u != sweep && stamp[v] + 1u != sweep) continue;
    const double we = w[e];
    if (!(we < INFINITY)) continue;
    const double du = __longlong_as_double((long long)dist[u]), dv = __longlong_as_double((long long)dist[v]);
    if (du + we < dv) {
      const unsigned long long nd = (unsigned long long)__double_as_longlong(du + we);
      if (atomicMin(&dist[v], nd) > nd) {
        stamp[v] = sweep;
        local = 1;
      }
    } else if (dv + we < du) {
      const unsigned long long nd = (unsigned long long)__double_as_longlong(dv + we);
      if (atomicMin(&dist[u], nd) > nd) {
        stamp[u] = sweep;
        local = 1;
      }
    }
  }
  if (__any(local) && (threadIdx.x & 63) == 0) atomicAdd(changed, 1u);
}

// pred[v] = a neighbour u with dist[u] + w == dist[v] (the smallest such u: deterministic)
__global__ void __launch_bounds__(256)
sssp_pred_kernel(const uint32_t* __restrict__ eu, const uint32_t* __restrict__ ev, const double* __restrict__ w,
                 size_t ne, const unsigned long long* __restrict__ dist, uint32_t* __restrict__ pred) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < ne; e += (size_t)gridDim.x * blockDim.x) {
    const double we = w[e];
    if (!(we < INFINITY)) continue;
    const uint32_t u = eu[e], v = ev[e];
    const double du = __longlong_as_double((long long)dist[u]), dv = __longlong_as_double((long long)dist[v]);
    if (du < INFINITY && du + we == dv) atomicMin(&pred[v], u);
    if (dv < INFINITY && dv + we == du) atomicMin(&pred[u], v);
  }
}
}  // namespace artp

struct artp_roadmap {
  artp_ctx* ctx = nullptr;
  artp_roadmap_params params{};
  int k = 0;
  uint64_t samples_drawn = 0;
  artp_preprocess_params density_params_copy{};  // params.density_params points here (see roadmap_fix_params)
  uint64_t n_reweights = 0;        // density re-weightings during the build
  uint64_t budget_flags = 0;       // bit 0: max_sample_time ended the sampling, bit 1: max_n_edges cut the graph
  // device copies for the label-correcting search of large roadmaps (roadmap_sssp_dev)
  uint32_t* d_euv = nullptr;       // eu | ev
  double* d_w = nullptr;           // edge weight, +inf = not usable
  unsigned long long* d_dist = nullptr;
  uint32_t* d_pred = nullptr;
  size_t d_graph_ne = 0, d_graph_nv = 0;
  bool d_graph_dirty = true;
  std::vector<double> verts;       // nv x 7; vertex 0 = start, 1 = goal
  std::vector<uint32_t> knn;       // nv x k (0xffffffff = none)
  std::vector<double> knn_dist;    // nv x k
  std::vector<uint32_t> eu, ev;    // candidate edges (u < v), sorted by (u, v)
  std::vector<uint8_t> evalid;     // interpolation rule verdict
  std::vector<uint32_t> einterp;   // interior states of the chain
  std::vector<double> ecost;
  std::vector<uint8_t> eremoved;   // removed by the lazy path check
  // vertices the CURRENT map invalidated but that stay in `verts` (artp_roadmap_revalidate, growing a construction-1
  // graph): not neighbour targets for a new query (the reference's invalid vertices are not in nn_).  Empty = none.
  std::vector<uint8_t> vinvalid;
  // The reference computes an edge's weight ONCE, in the direction it was added to the undirected graph
  // (opt_->motionCost(m, n) new -> old in lazy_prm_star_min_update.cpp:436; source -> target of boost::add_edge(prev, new) /
  // (m, n) in PRMMotionCostMaintainer::updateEdges, prm_motion_cost.cpp:33-44) -- it matters for the directional and the
  // learned objective.  eflip[e] = 1: that direction is ev -> eu.  Empty = all 0 (construction 0: eu -> ev).
  std::vector<uint8_t> eflip;
  // CSR over the valid, not removed edges
  std::vector<uint32_t> row, adj, adj_edge;
  std::vector<double> adjw;   // weight per adjacency slot, +inf = not usable (removed / infeasible): what the tree search reads
  bool csr_dirty = true;
  // Verdicts of the discrete motion validator per edge AND direction ([2 e] = eu -> ev, [2 e + 1] = ev -> eu: the
  // validator interpolates from its first argument): 0 = not checked, 1 = valid, 2 = invalid.  The reference's lazy
  // planners keep the same knowledge in edgeValidityProperty_.  Valid for one map version and one edge list.
  std::vector<uint8_t> emotion;
  uint64_t emotion_map_version = ~0ull;
  bool emotion_dirty = true;
  // resident in HBM for artp_roadmap_revalidate: the endpoint states of all edges (s1 rows, then s2 rows), and
  // what they are gathered from after a re-query replaced the first edges (vertex states, edge end points)
  double* d_edge_states = nullptr;
  size_t d_edge_cap = 0;
  bool d_edge_states_stale = true;
  size_t nv() const { return verts.size() / 7; }
};

namespace {

// OMPL 1.4.2 KStarStrategy for SE3 (dimension 6): k = ceil(e (1 + 1/6) ln n), n = the number of graph vertices
inline int roadmap_kstar(size_t n) {
  return (int)std::ceil(2.718281828459045 * (1.0 + 1.0 / 6.0) * std::log((double)n));
}

// The roadmap owns a copy of the caller's preprocess parameters; call after every struct assignment / swap.
void roadmap_fix_params(artp_roadmap* rm) {
  if (rm->params.density_params && rm->params.density_params != &rm->density_params_copy)
    rm->density_params_copy = *rm->params.density_params;
  if (rm->params.density_params) rm->params.density_params = &rm->density_params_copy;
}

void roadmap_build_csr(artp_roadmap* rm) {
  const size_t nv = rm->nv(), ne = rm->eu.size();
  rm->row.assign(nv + 1, 0);
  for (size_t e = 0; e < ne; ++e)
    if (rm->evalid[e] && !rm->eremoved[e]) {
      ++rm->row[rm->eu[e] + 1];
      ++rm->row[rm->ev[e] + 1];
    }
  for (size_t v = 0; v < nv; ++v) rm->row[v + 1] += rm->row[v];
  rm->adj.assign(rm->row[nv], 0);
  rm->adj_edge.assign(rm->row[nv], 0);
  rm->adjw.assign(rm->row[nv], INFINITY);
  std::vector<uint32_t> fill(rm->row.begin(), rm->row.end() - 1);
  for (size_t e = 0; e < ne; ++e)
    if (rm->evalid[e] && !rm->eremoved[e]) {
      const uint32_t u = rm->eu[e], v = rm->ev[e];
      const double w = (std::isfinite(rm->ecost[e]) && rm->ecost[e] >= 0.0) ? rm->ecost[e] : INFINITY;
      rm->adj[fill[u]] = v;
      rm->adjw[fill[u]] = w;
      rm->adj_edge[fill[u]++] = (uint32_t)e;
      rm->adj[fill[v]] = u;
      rm->adjw[fill[v]] = w;
      rm->adj_edge[fill[v]++] = (uint32_t)e;
    }
  rm->csr_dirty = false;
}

// A* from vertex 0 to vertex 1 (boost::astar_search with PathLengthObjective::motionCostHeuristic for
// the Euclidean objective, which is consistent; zero heuristic otherwise).  Returns false if unreachable.
bool roadmap_astar(artp_roadmap* rm, std::vector<uint32_t>* path, double* cost) {
  if (rm->csr_dirty) roadmap_build_csr(rm);
  const size_t nv = rm->nv();
  const double* V = rm->verts.data();
  const bool use_h = rm->params.objective == 0;
  auto h = [&](uint32_t v) {
    if (!use_h) return 0.0;
    const double dx = V[7 + 0] - V[(size_t)v * 7 + 0], dy = V[7 + 1] - V[(size_t)v * 7 + 1],
                 dz = V[7 + 2] - V[(size_t)v * 7 + 2];
    return std::sqrt(dx * dx + dy * dy + dz * dz) / rm->params.max_lon_vel;
  };
  std::vector<double> g(nv, INFINITY);
  std::vector<uint32_t> prev(nv, 0xffffffffu);
  std::vector<uint8_t> closed(nv, 0);
  using Item = std::pair<double, uint32_t>;
  std::priority_queue<Item, std::vector<Item>, std::greater<Item>> open;
  g[0] = 0.0;
  open.push({h(0), 0});
  while (!open.empty()) {
    const uint32_t u = open.top().second;
    open.pop();
    if (closed[u]) continue;
    closed[u] = 1;
    if (u == 1) break;
    for (uint32_t a = rm->row[u]; a < rm->row[u + 1]; ++a) {
      const uint32_t v = rm->adj[a];
      const uint32_t e = rm->adj_edge[a];
      if (rm->eremoved[e]) continue;  // removed by the lazy path check since the CSR was built
      const double w = rm->ecost[e];
      if (!std::isfinite(w)) continue;
      const double ng = g[u] + w;
      if (ng < g[v]) {
        g[v] = ng;
        prev[v] = u;
        open.push({ng + h(v), v});
      }
    }
  }
  if (!closed[1]) return false;
  path->clear();
  for (uint32_t v = 1; v != 0xffffffffu; v = prev[v]) path->push_back(v);
  std::reverse(path->begin(), path->end());
  *cost = g[1];
  return true;
}

// The same search on the device for large roadmaps (see sssp_relax_kernel).  Same result as roadmap_astar up to the
// choice among equal-cost paths.
bool roadmap_sssp_dev(artp_roadmap* rm, std::vector<uint32_t>* path, double* cost, bool* ok) {
  artp_ctx* c = rm->ctx;
  const size_t nv = rm->nv(), ne = rm->eu.size();
  *ok = false;
  if (hipSetDevice(c->device) != hipSuccess) return false;
  hipStream_t st = c->stream;
  if (rm->d_graph_ne != ne || rm->d_graph_nv != nv || !rm->d_euv) {
    for (void* p : {(void*)rm->d_euv, (void*)rm->d_w, (void*)rm->d_dist, (void*)rm->d_pred})
      if (p) (void)hipFree(p);
    rm->d_euv = nullptr; rm->d_w = nullptr; rm->d_dist = nullptr; rm->d_pred = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&rm->d_euv), 2 * ne * sizeof(uint32_t)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&rm->d_w), ne * sizeof(double)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&rm->d_dist), nv * sizeof(unsigned long long) + 16) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&rm->d_pred), 2 * nv * sizeof(uint32_t)) != hipSuccess)  // pred | stamp
      return false;
    rm->d_graph_ne = ne;
    rm->d_graph_nv = nv;
    rm->d_graph_dirty = true;
    if (hipMemcpyAsync(rm->d_euv, rm->eu.data(), ne * 4, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemcpyAsync(rm->d_euv + ne, rm->ev.data(), ne * 4, hipMemcpyHostToDevice, st) != hipSuccess)
      return false;
  }
  if (rm->d_graph_dirty) {
    std::vector<double> w(ne);
    for (size_t e = 0; e < ne; ++e)
      // negative weights (a learned cost below 0) would break the atomicMin on the bit pattern of the distances, which
      // orders non-negative doubles only: such an edge is not traversable here (the reference's costs are >= 0)
      w[e] = (rm->evalid[e] && !rm->eremoved[e] && std::isfinite(rm->ecost[e]) && rm->ecost[e] >= 0.0) ? rm->ecost[e]
                                                                                                           : INFINITY;
    if (hipMemcpyAsync(rm->d_w, w.data(), ne * sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
      return false;
    rm->d_graph_dirty = false;
  }
  unsigned* d_changed = reinterpret_cast<unsigned*>(rm->d_dist + nv);
  // dist = +inf (0x7ff0...), dist[0] = 0
  std::vector<unsigned long long> init(nv, 0x7ff0000000000000ull);
  init[0] = 0ull;
  if (hipMemcpyAsync(rm->d_dist, init.data(), nv * 8, hipMemcpyHostToDevice, st) != hipSuccess) return false;
  // stamps: the source "moved" in sweep 0, nobody else yet (0xfefefefe + 1 never equals a sweep number)
  unsigned* d_stamp = rm->d_pred + nv;
  const unsigned zero = 0;
  if (hipMemsetAsync(d_stamp, 0xfe, nv * 4, st) != hipSuccess ||
      hipMemcpyAsync(d_stamp, &zero, 4, hipMemcpyHostToDevice, st) != hipSuccess)
    return false;
  size_t blocks = (ne + 255) / 256;
  if (blocks > (size_t)c->n_cus * 8) blocks = (size_t)c->n_cus * 8;
  for (unsigned sweep = 1; sweep < 1000000u; sweep += 16) {
    if (hipMemsetAsync(d_changed, 0, 4, st) != hipSuccess) return false;
    for (unsigned r = 0; r < 16; ++r)
      hipLaunchKernelGGL(artp::sssp_relax_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const uint32_t*)rm->d_euv,
                         (const uint32_t*)(rm->d_euv + ne), (const double*)rm->d_w, ne, rm->d_dist, d_stamp, sweep + r,
                         d_changed);
    unsigned changed = 0;
    if (hipMemcpyAsync(&changed, d_changed, 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
      return false;
    if (!changed) break;
  }
  // one more sweep group proved the fixed point (changed == 0 over 16 sweeps); predecessors
  if (hipMemsetAsync(rm->d_pred, 0xff, nv * 4, st) != hipSuccess) return false;
  hipLaunchKernelGGL(artp::sssp_pred_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const uint32_t*)rm->d_euv,
                     (const uint32_t*)(rm->d_euv + ne), (const double*)rm->d_w, ne,
                     (const unsigned long long*)rm->d_dist, rm->d_pred);
  std::vector<uint32_t> pred(nv);
  unsigned long long dgoal = 0;
  if (hipMemcpyAsync(pred.data(), rm->d_pred, nv * 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipMemcpyAsync(&dgoal, rm->d_dist + 1, 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess)
    return false;
  *ok = true;
  double dg;
  std::memcpy(&dg, &dgoal, 8);
  if (!(dg < INFINITY)) return false;
  path->clear();
  uint32_t v = 1;
  for (size_t guard = 0; guard <= nv; ++guard) {
    path->push_back(v);
    if (v == 0) break;
    v = pred[v];
    if (v == 0xffffffffu) return false;  // cannot happen at the fixed point
  }
  if (path->back() != 0) {
    // zero-weight edges (coincident vertices, a learned cost of 0) make dist[u] + 0 == dist[v] true in both
    // directions, so the predecessor chain can close a cycle that never reaches the start: no path from here --
    // the caller falls back to the host A*, which cannot cycle
    path->clear();
    *ok = false;
    return false;
  }
  std::reverse(path->begin(), path->end());
  *cost = dg;
  return true;
}

#define RM_TRY(expr)          \
  do {                        \
    const int rc_ = (expr);   \
    if (rc_ != ARTP_OK) {     \
      cleanup();              \
      return rc_;             \
    }                         \
  } while (0)
#define RM_HIP(expr)                                                                   \
  do {                                                                                 \
    const hipError_t e_ = (expr);                                                      \
    if (e_ != hipSuccess) {                                                            \
      c->last_error = std::string(#expr) + ": " + hipGetErrorString(e_);               \
      cleanup();                                                                       \
      return ARTP_ERR_HIP;                                                             \
    }                                                                                  \
  } while (0)

// Verdict (0.5 m interpolation rule), interior-state count and chain cost of ne edges whose endpoint states
// are on the device; results to the host arrays.
// direct: no interpolation rule -- every edge is one sub-edge of unknown validity (reported valid), the way the
// reference's planners put the edges of an already dense neighbourhood (n_interp == 0, prm_motion_cost.cpp:345-348)
// and LazyPRM*'s edges into their graphs.
// segment_cost_step > 0 (learned objective only): the edges are path SEGMENTS priced the way
// MotionCostObjective::motionCost prices a motion -- split by max_query_edge_length, not along the validity chain.
int roadmap_eval_edges_dev(artp_ctx* c, const artp_roadmap_params* prm, const double* d_s1, const double* d_s2,
                           size_t ne, uint8_t* evalid, uint32_t* einterp, double* ecost, bool direct = false,
                           double segment_cost_step = 0.0) {
  if (ne == 0) return ARTP_OK;
  hipStream_t st = c->stream;
  double* d_cost = nullptr;
  uint8_t* d_evalid = nullptr;
  uint32_t *d_einterp = nullptr, *d_rows = nullptr, *d_off = nullptr, *d_ncost = nullptr;
  float *d_em = nullptr, *d_c3 = nullptr;
  void* d_cub2 = nullptr;
  auto cleanup = [&]() {
    for (void* p : {(void*)d_cost, (void*)d_evalid, (void*)d_einterp, (void*)d_rows, (void*)d_off, (void*)d_em,
                    (void*)d_c3, d_cub2, (void*)d_ncost})
      if (p) (void)hipFree(p);
  };
  RM_HIP(hipMalloc(reinterpret_cast<void**>(&d_cost), ne * sizeof(double)));
  RM_HIP(hipMalloc(reinterpret_cast<void**>(&d_evalid), ne));
  RM_HIP(hipMalloc(reinterpret_cast<void**>(&d_einterp), ne * sizeof(uint32_t)));
  if (direct) {
    RM_HIP(hipMemsetAsync(d_evalid, 1, ne, st));
    RM_HIP(hipMemsetAsync(d_einterp, 0, ne * sizeof(uint32_t), st));
  } else {
    RM_TRY(artp_check_edges_interp_dev(c, d_s1, d_s2, ne, d_evalid, d_einterp));
  }
  if (prm->objective <= 1) {
    artp::PathLengthParams pl{prm->objective == 1, prm->max_lon_vel, prm->max_lat_vel, prm->max_ang_vel};
    hipLaunchKernelGGL(artp::edge_chain_cost_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, st, pl, d_s1,
                       d_s2, (const uint32_t*)d_einterp, ne, d_cost);
  } else {
    // learned cost: one EdgeMatrix row per sub-edge, one batched query, per-chain reduction
    const uint32_t* d_chain = d_einterp;  // sub-edges of the validity chain (what the reference's graph holds) ...
    if (segment_cost_step > 0.0) {        // ... or motionCost's own split of a path segment
      RM_HIP(hipMalloc(reinterpret_cast<void**>(&d_ncost), ne * sizeof(uint32_t)));
      hipLaunchKernelGGL(artp::motion_cost_interp_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, st, d_s1, d_s2, ne,
                         segment_cost_step, d_ncost);
      d_chain = d_ncost;
    }
    RM_HIP(hipMalloc(reinterpret_cast<void**>(&d_rows), (ne + 1) * 4));
    RM_HIP(hipMalloc(reinterpret_cast<void**>(&d_off), (ne + 1) * 4));
    size_t need = 0;
    uint32_t total = 0;
    hipLaunchKernelGGL(artp::chain_rows_kernel, dim3((unsigned)((ne + 256) / 256)), dim3(256), 0, st, d_chain, ne, d_rows);
    RM_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, need, d_rows, d_off, (int)(ne + 1), st));
    RM_HIP(hipMalloc(&d_cub2, need + 256));
    size_t cap2 = need + 256;
    RM_HIP(hipcub::DeviceScan::ExclusiveSum(d_cub2, cap2, d_rows, d_off, (int)(ne + 1), st));
    RM_HIP(hipMemcpyAsync(&total, d_off + ne, 4, hipMemcpyDeviceToHost, st));
    RM_HIP(hipStreamSynchronize(st));
    RM_HIP(hipMalloc(reinterpret_cast<void**>(&d_em), (size_t)total * 6 * 4));
    RM_HIP(hipMalloc(reinterpret_cast<void**>(&d_c3), (size_t)total * 3 * 4));
    hipLaunchKernelGGL(artp::chain_edge_matrix_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, st, d_s1, d_s2,
                       d_chain, (const uint32_t*)d_off, ne, d_em);
    RM_TRY(roadmap_cost_query_dev(c, d_em, total, d_c3));   // the caller's MotionCostFunc when one is installed
    hipLaunchKernelGGL(artp::chain_motion_cost_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, st,
                       (const float*)d_c3, (const uint32_t*)d_off, d_chain, ne, prm->w_energy,
                       prm->w_time, prm->w_risk, prm->risk_threshold, d_cost);
  }
  RM_HIP(hipGetLastError());
  RM_HIP(hipStreamSynchronize(st));
  RM_HIP(hipMemcpy(evalid, d_evalid, ne, hipMemcpyDeviceToHost));
  RM_HIP(hipMemcpy(einterp, d_einterp, ne * sizeof(uint32_t), hipMemcpyDeviceToHost));
  RM_HIP(hipMemcpy(ecost, d_cost, ne * sizeof(double), hipMemcpyDeviceToHost));
  RM_TRY(check_error_flag(c));
  cleanup();
  return ARTP_OK;
}

// same for edges (eu[e], ev[e]) of host vertices: gathers the endpoint states on the host first
int roadmap_eval_edges_host(artp_ctx* c, const artp_roadmap_params* prm, const std::vector<double>& verts,
                            const uint32_t* eu, const uint32_t* ev, size_t ne, uint8_t* evalid, uint32_t* einterp,
                            double* ecost, bool direct = false, const uint8_t* flip = nullptr,
                            double segment_cost_step = 0.0) {
  if (ne == 0) return ARTP_OK;
  std::vector<double> s(2 * ne * 7);
  for (size_t e = 0; e < ne; ++e) {
    const bool fl = flip && flip[e];
    std::memcpy(&s[e * 7], &verts[(size_t)(fl ? ev[e] : eu[e]) * 7], 7 * sizeof(double));
    std::memcpy(&s[(ne + e) * 7], &verts[(size_t)(fl ? eu[e] : ev[e]) * 7], 7 * sizeof(double));
  }
  double* d_s = nullptr;
  auto cleanup = [&]() {
    if (d_s) (void)hipFree(d_s);
  };
  RM_HIP(hipSetDevice(c->device));
  RM_HIP(hipMalloc(reinterpret_cast<void**>(&d_s), s.size() * sizeof(double)));
  RM_HIP(hipMemcpy(d_s, s.data(), s.size() * sizeof(double), hipMemcpyHostToDevice));
  RM_TRY(roadmap_eval_edges_dev(c, prm, d_s, d_s + ne * 7, ne, evalid, einterp, ecost, direct, segment_cost_step));
  cleanup();
  return ARTP_OK;
}

}  // namespace

extern "C" {

void artp_roadmap_params_defaults(artp_roadmap_params* p) {
  if (!p) return;
  std::memset(p, 0, sizeof(*p));
  p->seed = 42;
  p->first_index = 0;
  p->n_milestones = 10000;         // Params::planner.prm_motion_cost.max_n_vertices (params.h:51)
  p->k_neighbors = 0;              // PRM* rule
  p->objective = 0;                // use_directional_cost{false} (params.h:70)
  p->max_lon_vel = 0.5;            // params.h:71-73
  p->max_lat_vel = 0.1;
  p->max_ang_vel = 0.5;
  p->max_replans = 1000;
  p->w_energy = 0.0f;              // Params::planner.prm_motion_cost.cost_weights (params.h:58-62)
  p->w_time = 1.0f;
  p->w_risk = 5.0f;
  p->risk_threshold = 0.1f;        // params.h:55
  p->max_n_edges = 0;              // budgets and re-weighting are opt-in here (params.h:50-53 carry the
  p->recompute_density_after_n_samples = 0;  // reference's defaults: 50 000 / 1000 / 2.0 s)
  p->max_sample_time = 0.0;
  p->density_map = nullptr;
  p->density_params = nullptr;
  p->max_query_edge_length = 0.5;  // params.h:54
}

void artp_roadmap_destroy(artp_roadmap* rm) {
  if (!rm) return;
  if (rm->d_edge_states) (void)hipFree(rm->d_edge_states);
  for (void* p : {(void*)rm->d_euv, (void*)rm->d_w, (void*)rm->d_dist, (void*)rm->d_pred})
    if (p) (void)hipFree(p);
  delete rm;
}

// Connection + evaluation of a vertex set that is already in HBM (d_verts: nv x 7; vertex 0 = start, 1 = goal):
// k nearest neighbours, symmetrised unique candidate edges, the 0.5 m interpolation rule, chain costs.
static int roadmap_connect(artp_ctx* c, const artp_roadmap_params* prm, const double* d_verts, size_t nv,
                           artp_roadmap** out) {
  *out = nullptr;
  uint64_t* d_cnt = nullptr;
  uint32_t* d_knn = nullptr;
  double* d_knn_dist = nullptr;
  unsigned long long *d_keys = nullptr, *d_keys_sorted = nullptr, *d_keys_unique = nullptr;
  double *d_s1 = nullptr, *d_s2 = nullptr;
  void* d_cub = nullptr;
  auto cleanup = [&]() {
    for (void* p : {(void*)d_cnt, (void*)d_knn, (void*)d_knn_dist, (void*)d_keys, (void*)d_keys_sorted,
                    (void*)d_keys_unique, (void*)d_s1, d_cub})
      if (p) (void)hipFree(p);
  };
  hipStream_t st = c->stream;
  RM_HIP(hipMalloc(reinterpret_cast<void**>(&d_cnt), sizeof(uint64_t)));
  // 2. k nearest neighbours
  int k = (int)prm->k_neighbors;
  if (k <= 0) k = roadmap_kstar((size_t)nv);
  if (k > (int)nv - 1) k = (int)nv - 1;
  if (k < 1) k = 1;
  if (k > 128) k = 128;
  // construction 2 (LazyPRMStarMinUpdate::addValidMilestone, lazy_prm_star_min_update.cpp:424-446): vertex i is
  // connected to the k_i nearest of its PREDECESSORS, k_i = the rule at the graph size of its insertion (i + 1, itself
  // included; it enters the nearest-neighbour structure last)
  const bool pred_only = prm->construction == 2;
  int* d_k_of = nullptr;
  if (pred_only) {
    std::vector<int> k_of(nv);
    for (size_t i = 0; i < nv; ++i) {
      int ki = prm->k_neighbors ? (int)prm->k_neighbors : roadmap_kstar(i + 1);
      if (ki > (int)i) ki = (int)i;
      k_of[i] = ki > k ? k : ki;
    }
    if (hipMalloc(reinterpret_cast<void**>(&d_k_of), nv * sizeof(int)) != hipSuccess ||
        hipMemcpy(d_k_of, k_of.data(), nv * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
      if (d_k_of) (void)hipFree(d_k_of);
      c->last_error = "k table upload failed";
      cleanup();
      return ARTP_ERR_HIP;
    }
  }
  RM_HIP(hipMalloc(reinterpret_cast<void**>(&d_knn), nv * k * sizeof(uint32_t)));
  RM_HIP(hipMalloc(reinterpret_cast<void**>(&d_knn_dist), nv * k * sizeof(double)));
  {
    // grid over the map: about three vertices per cell
    artp::KnnGrid g;
    {
      std::lock_guard<std::recursive_mutex> lock(c->mu);
      g.x0 = c->geom.pos_x - 0.5 * c->geom.len_x;
      g.y0 = c->geom.pos_y - 0.5 * c->geom.len_y;
      const double area = c->geom.len_x * c->geom.len_y;
      g.h = std::sqrt(area * 3.0 / (double)nv);
      const double hmin = std::max(c->geom.len_x, c->geom.len_y) / 2048.0;
      if (g.h < hmin) g.h = hmin;
      g.inv_h = 1.0 / g.h;
      g.gx = std::max(1, (int)std::ceil(c->geom.len_x * g.inv_h));
      g.gy = std::max(1, (int)std::ceil(c->geom.len_y * g.inv_h));
    }
    const int ncell = g.gx * g.gy;
    uint32_t *d_cell = nullptr, *d_id = nullptr, *d_cell_s = nullptr, *d_id_s = nullptr, *d_start = nullptr;
    double* d_vs = nullptr;
    void* d_cub1 = nullptr;
    auto cleanup_knn = [&]() {
      for (void* p : {(void*)d_cell, (void*)d_id, (void*)d_cell_s, (void*)d_id_s, (void*)d_start, (void*)d_vs, d_cub1})
        if (p) (void)hipFree(p);
    };
    size_t need = 0;
    bool okk = hipMalloc(reinterpret_cast<void**>(&d_cell), nv * 4) == hipSuccess &&
               hipMalloc(reinterpret_cast<void**>(&d_id), nv * 4) == hipSuccess &&
               hipMalloc(reinterpret_cast<void**>(&d_cell_s), nv * 4) == hipSuccess &&
               hipMalloc(reinterpret_cast<void**>(&d_id_s), nv * 4) == hipSuccess &&
               hipMalloc(reinterpret_cast<void**>(&d_start), ((size_t)ncell + 1) * 4) == hipSuccess &&
               hipMalloc(reinterpret_cast<void**>(&d_vs), nv * 7 * sizeof(double)) == hipSuccess &&
               hipcub::DeviceRadixSort::SortPairs(nullptr, need, d_cell, d_cell_s, d_id, d_id_s, (int)nv, 0, 32, st) ==
                   hipSuccess &&
               hipMalloc(&d_cub1, need + 256) == hipSuccess;
    if (okk) {
      hipLaunchKernelGGL(artp::knn_cell_ids_kernel, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, st,
                         (const double*)d_verts, (int)nv, g, d_cell, d_id);
      size_t cap1 = need + 256;
      okk = hipcub::DeviceRadixSort::SortPairs(d_cub1, cap1, d_cell, d_cell_s, d_id, d_id_s, (int)nv, 0, 32, st) ==
            hipSuccess;
    }
    if (okk) {
      hipLaunchKernelGGL(artp::knn_permute_kernel, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, st,
                         (const double*)d_verts, (const uint32_t*)d_cell_s, (const uint32_t*)d_id_s, (int)nv, ncell,
                         d_vs, d_start);
      const size_t lds = (size_t)k * 64 * 12;
      okk = hipFuncSetAttribute(reinterpret_cast<const void*>(artp::knn_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
      if (okk) {
        hipLaunchKernelGGL(artp::knn_kernel, dim3((unsigned)((nv + 63) / 64)), dim3(64), lds, st, (const double*)d_vs,
                           (const uint32_t*)d_id_s, (const uint32_t*)d_start, g, (int)nv, k, (const int*)d_k_of,
                           pred_only ? 1 : 0, d_knn, d_knn_dist);
        okk = hipGetLastError() == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
      }
    }
    cleanup_knn();
    if (d_k_of) (void)hipFree(d_k_of);
    if (!okk) {
      c->last_error = "k-NN stage failed";
      cleanup();
      return ARTP_ERR_HIP;
    }
  }

  // 3. candidate edges: symmetrised, unique
  const size_t nk = nv * (size_t)k;
  RM_HIP(hipMalloc(reinterpret_cast<void**>(&d_keys), nk * 8));
  RM_HIP(hipMalloc(reinterpret_cast<void**>(&d_keys_sorted), nk * 8));
  RM_HIP(hipMalloc(reinterpret_cast<void**>(&d_keys_unique), nk * 8));
  hipLaunchKernelGGL(artp::knn_edge_keys_kernel, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, st,
                     (const uint32_t*)d_knn, (int)nv, k, d_keys);
  RM_HIP(hipGetLastError());
  size_t ne = 0;
  {
    size_t need1 = 0, need2 = 0;
    RM_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, need1, d_keys, d_keys_sorted, (int)nk, 0, 64, st));
    RM_HIP(hipcub::DeviceSelect::Unique(nullptr, need2, d_keys_sorted, d_keys_unique,
                                        reinterpret_cast<unsigned long long*>(d_cnt), (int)nk, st));
    const size_t need = std::max(need1, need2) + 256;
    RM_HIP(hipMalloc(&d_cub, need));
    size_t cap = need;
    RM_HIP(hipcub::DeviceRadixSort::SortKeys(d_cub, cap, d_keys, d_keys_sorted, (int)nk, 0, 64, st));
    cap = need;
    RM_HIP(hipcub::DeviceSelect::Unique(d_cub, cap, d_keys_sorted, d_keys_unique,
                                        reinterpret_cast<unsigned long long*>(d_cnt), (int)nk, st));
    uint64_t nu = 0;
    RM_HIP(hipMemcpyAsync(&nu, d_cnt, sizeof(nu), hipMemcpyDeviceToHost, st));
    RM_HIP(hipStreamSynchronize(st));
    ne = (size_t)nu;
    // the all-ones key (missing neighbour slots) sorts last
    unsigned long long last = 0;
    if (ne) {
      RM_HIP(hipMemcpy(&last, d_keys_unique + (ne - 1), 8, hipMemcpyDeviceToHost));
      if (last == ~0ull) --ne;
    }
  }

  auto rm = new artp_roadmap();
  rm->ctx = c;
  rm->params = *prm;
  roadmap_fix_params(rm);
  rm->k = k;
  rm->verts.resize(nv * 7);
  rm->knn.resize(nk);
  rm->knn_dist.resize(nk);
  auto fail = [&](int rc) {
    delete rm;
    cleanup();
    return rc;
  };
  if (hipMemcpy(rm->verts.data(), d_verts, nv * 7 * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemcpy(rm->knn.data(), d_knn, nk * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemcpy(rm->knn_dist.data(), d_knn_dist, nk * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
    return fail(ARTP_ERR_HIP);

  // 4. edge verdicts (0.5 m interpolation rule) and chain costs
  rm->eu.resize(ne);
  rm->ev.resize(ne);
  rm->evalid.assign(ne, 0);
  rm->einterp.assign(ne, 0);
  rm->ecost.assign(ne, 0.0);
  rm->eremoved.assign(ne, 0);
  if (ne) {
    if (hipMalloc(reinterpret_cast<void**>(&d_s1), 2 * ne * 7 * sizeof(double)) != hipSuccess) return fail(ARTP_ERR_HIP);
    d_s2 = d_s1 + ne * 7;
    hipLaunchKernelGGL(artp::gather_edge_states_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, st,
                       (const double*)d_verts, (const unsigned long long*)d_keys_unique, ne, d_s1, d_s2, pred_only ? 1 : 0);
    if (pred_only) rm->eflip.assign(ne, 1);
    // construction 2: the lazy planner puts DIRECT edges of unknown validity into its graph (no interpolation rule)
    const int rc = roadmap_eval_edges_dev(c, prm, d_s1, d_s2, ne, rm->evalid.data(), rm->einterp.data(),
                                          rm->ecost.data(), pred_only);
    if (rc != ARTP_OK) return fail(rc);
    std::vector<unsigned long long> keys(ne);
    if (hipMemcpy(keys.data(), d_keys_unique, ne * 8, hipMemcpyDeviceToHost) != hipSuccess) return fail(ARTP_ERR_HIP);
    for (size_t e = 0; e < ne; ++e) {
      rm->eu[e] = (uint32_t)(keys[e] >> 32);
      rm->ev[e] = (uint32_t)(keys[e] & 0xffffffffu);
    }
    rm->d_edge_states = d_s1;  // stays resident (freed by artp_roadmap_destroy)
    rm->d_edge_cap = ne;
    rm->d_edge_states_stale = false;
    d_s1 = d_s2 = nullptr;
  }
  cleanup();
  *out = rm;
  return ARTP_OK;
}


// Build over [start, goal, kept milestones (host, n_keep x 7, already known valid), n_new fresh accepted samples
// drawn from sample index first_new on].  artp_roadmap_build: no kept milestones; artp_roadmap_grow: the
// roadmap's own.
static int roadmap_build_impl(artp_ctx* c, const artp_roadmap_params* prm_in, const double* start7, const double* goal7,
                              const double* keep, size_t n_keep, size_t n_new, uint64_t first_new,
                              artp_roadmap** out) {
  artp_roadmap_params prm_local = *prm_in;
  prm_local.n_milestones = n_keep + n_new;
  const artp_roadmap_params* prm = &prm_local;
  *out = nullptr;
  const size_t nm = n_keep + n_new, nv = nm + 2;
  double* d_verts = nullptr;     // nv x 7
  double* d_batch = nullptr;     // sample batch
  uint8_t* d_valid = nullptr;
  double* d_compact = nullptr;
  uint64_t* d_cnt = nullptr;
  auto cleanup = [&]() {
    for (void* p : {(void*)d_verts, (void*)d_batch, (void*)d_valid, (void*)d_compact, (void*)d_cnt})
      if (p) (void)hipFree(p);
  };
  RM_HIP(hipSetDevice(c->device));
  hipStream_t st = c->stream;
  RM_HIP(hipMalloc(reinterpret_cast<void**>(&d_verts), nv * 7 * sizeof(double)));
  RM_HIP(hipMalloc(reinterpret_cast<void**>(&d_cnt), sizeof(uint64_t)));

  // start and goal must be valid states (baseSolve: INVALID_START / INVALID_GOAL, prm_motion_cost.cpp:452-476)
  {
    double sg[14];
    std::memcpy(sg, start7, 7 * sizeof(double));
    std::memcpy(sg + 7, goal7, 7 * sizeof(double));
    uint8_t ok[2] = {0, 0};
    RM_TRY(artp_validate_states(c, sg, 2, ok, nullptr));
    if (!ok[0] || !ok[1]) {
      c->last_error = !ok[0] ? "start state is not valid" : "goal state is not valid";
      cleanup();
      return ARTP_ERR_INVALID_ARG;
    }
    RM_HIP(hipMemcpyAsync(d_verts, sg, sizeof(sg), hipMemcpyHostToDevice, st));
  }

  // 1. milestones: the kept ones, then accepted states of the sample stream, in index order
  size_t have = n_keep;
  uint64_t next = first_new;
  if (n_keep)
    RM_HIP(hipMemcpyAsync(d_verts + 2 * 7, keep, n_keep * 7 * sizeof(double), hipMemcpyHostToDevice, st));
  uint64_t n_reweights = 0, budget_flags = 0;
  size_t nm_final = nm;  // fewer when the sampling-time budget ends the loop
  if (n_new) {
    // In-build re-weighting (prm_motion_cost.cpp:190-193): every R accepted vertices the sampling distribution is
    // recomputed from the inverse density of ALL vertices so far.  A round therefore ends at the next multiple of R;
    // the sample stream continues right behind the last sample the round consumed (nothing is drawn twice or
    // skipped, whatever the batch size).
    const size_t R = (prm->density_map && prm->density_params && prm->recompute_density_after_n_samples)
                         ? prm->recompute_density_after_n_samples : 0;
    size_t batch = std::max<size_t>(4 * (R ? std::min<size_t>(R, n_new) : n_new), 1u << 14);
    if (batch > (1u << 22)) batch = 1u << 22;
    // with a sampling-time budget the clock is read after every batch (the reference reads it every 100 samples)
    if (prm->max_sample_time > 0.0 && batch > (1u << 16)) batch = 1u << 16;
    uint32_t* d_idx = nullptr;
    RM_HIP(hipMalloc(reinterpret_cast<void**>(&d_batch), batch * 7 * sizeof(double)));
    RM_HIP(hipMalloc(reinterpret_cast<void**>(&d_compact), batch * 7 * sizeof(double) + batch * sizeof(uint32_t)));
    d_idx = reinterpret_cast<uint32_t*>(d_compact + batch * 7);
    RM_HIP(hipMalloc(reinterpret_cast<void**>(&d_valid), batch));
    const auto t_start = std::chrono::steady_clock::now();
    size_t n_proc = have / (R ? R : 1);  // the reference counts the graph's milestones (start / goal join later)
    int rounds = 0;
    while (have < nm) {
      RM_TRY(artp_sample_and_validate_dev(c, prm->seed, next, batch, d_batch, d_valid, nullptr));
      RM_TRY(artp_compact_valid_dev(c, d_batch, d_valid, batch, d_compact, d_cnt));
      uint64_t got = 0;
      RM_HIP(hipMemcpyAsync(&got, d_cnt, sizeof(got), hipMemcpyDeviceToHost, st));
      RM_HIP(hipStreamSynchronize(st));
      size_t want = nm - have;
      if (R) want = std::min(want, (n_proc + 1) * R - have);  // up to the next re-weighting point
      const size_t take = std::min<size_t>((size_t)got, want);
      RM_HIP(hipMemcpyAsync(d_verts + (2 + have) * 7, d_compact, take * 7 * sizeof(double), hipMemcpyDeviceToDevice, st));
      have += take;
      if (take < got) {
        // the round ended inside the batch: continue behind the sample that gave the last vertex taken
        uint32_t last = 0;
        RM_TRY(artp_compact_valid_indices_dev(c, d_valid, batch, d_idx, d_cnt));
        RM_HIP(hipMemcpyAsync(&last, d_idx + (take - 1), sizeof(last), hipMemcpyDeviceToHost, st));
        RM_HIP(hipStreamSynchronize(st));
        next += (uint64_t)last + 1;
      } else {
        next += batch;
      }
      if (R && have / R > n_proc) {
        // Map::reApplyPreprocessing(): the CDF follows the inverse vertex density from here on (also after the
        // last vertex, like the reference: the next growth samples from it)
        RM_TRY(artp_preprocessed_reweight_dev(c, prm->density_map, prm->density_params, d_verts + 14, have, 1));
        n_proc = have / R;
        ++n_reweights;
      }
      if (prm->max_sample_time > 0.0 &&
          std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count() > prm->max_sample_time) {
        if (have < nm) budget_flags |= 1u;  // "Reached sample timer limit." (prm_motion_cost.cpp:177-185)
        break;
      }
      if (++rounds > 100000 || (got == 0 && rounds > 8 && !R)) {
        c->last_error = "sampler produced too few valid states for the requested roadmap";
        cleanup();
        return ARTP_ERR_CAPACITY;
      }
    }
    nm_final = have;
    if (nm_final < 1) {
      c->last_error = "the sampling-time budget ended before the first milestone";
      cleanup();
      return ARTP_ERR_CAPACITY;
    }
  }

  // 2.-4. connect; PRMMotionCostMaintainer::sampleGraph also stops at max_n_edges (prm_motion_cost.cpp:171-172): the
  // vertex set is cut back to the longest prefix whose candidate edges stay within the budget and reconnected
  // (edge (u, v), u < v, exists among the first m vertices iff v < m; the counts per larger endpoint are a histogram)
  size_t nv_use = nm_final + 2;
  artp_roadmap* rm = nullptr;
  for (int attempt = 0;; ++attempt) {
    artp_roadmap_params p_use = *prm;
    p_use.n_milestones = (uint32_t)(nv_use - 2);
    const int rc = roadmap_connect(c, &p_use, d_verts, nv_use, &rm);
    if (rc != ARTP_OK) {
      cleanup();
      return rc;
    }
    if (!prm->max_n_edges || rm->eu.size() <= prm->max_n_edges || nv_use <= 3 || attempt >= 12) break;
    std::vector<uint32_t> per_v(nv_use, 0);
    for (uint32_t v : rm->ev) ++per_v[v];
    size_t m = 2, acc = 0;  // edges of THIS graph among its first m vertices
    while (m < nv_use && acc + per_v[m] <= prm->max_n_edges) acc += per_v[m++];
    // reconnecting the prefix gives its vertices nearer neighbours inside the prefix, i.e. more edges than `acc`:
    // from the second attempt on also shrink in proportion to the overshoot
    if (attempt > 0) m = std::min(m, (size_t)((double)(nv_use - 2) * prm->max_n_edges / (double)rm->eu.size() * 0.98) + 2);
    if (m >= nv_use) m = nv_use - 1;
    if (m < 3) m = 3;
    artp_roadmap_destroy(rm);
    rm = nullptr;
    nv_use = m;
    budget_flags |= 2u;
  }
  // the prefix search is bounded (12 reconnections): a graph still over the edge budget after that says so (bit 2)
  // instead of passing silently
  if (prm->max_n_edges && rm->eu.size() > prm->max_n_edges) budget_flags |= 4u;
  rm->samples_drawn = next - first_new;
  rm->n_reweights = n_reweights;
  rm->budget_flags = budget_flags;
  cleanup();
  *out = rm;
  return ARTP_OK;
}

// ---- construction 1: the reference's own insertion order (PRMMotionCost::addValidMilestone) ---------------------------
// The graph under construction.  Vertex ids: 0 = start, 1 = goal (their rows are reserved from the beginning, they are
// INSERTED last, like baseSolve does, prm_motion_cost.cpp:447-470), 2.. = insertion order (milestones and the chain
// vertices their connection attempts leave behind).
struct IncrementalGraph {
  artp_ctx* c = nullptr;
  artp_roadmap_params prm{};
  std::vector<double> verts;
  std::vector<uint8_t> inserted;            // vertex is in the graph (and, but for the milestone in flight, in nn_)
  std::vector<uint32_t> eu, ev;             // sub-edges, creation order
  size_t n_graph = 0;                       // boost::num_vertices(g_)
  std::vector<uint32_t> knn_rows;           // neighbour list of every MILESTONE (id, then its neighbours), for export
  int k_max = 0;
  uint64_t states_checked = 0;
  // nn_: uniform xy grid of vertex ids
  double x0 = 0, y0 = 0, h = 1, inv_h = 1;
  int gx = 1, gy = 1;
  std::vector<std::vector<uint32_t>> cells;
  // device staging of one milestone's chains
  double *d_tasks = nullptr, *d_states = nullptr;
  uint8_t* d_valid = nullptr;
  size_t cap = 0;
  std::vector<double> tasks, states;
  std::vector<uint8_t> valid;
  ~IncrementalGraph() {
    for (void* p : {(void*)d_tasks, (void*)d_states, (void*)d_valid})
      if (p) (void)hipFree(p);
  }
  size_t nv() const { return verts.size() / 7; }
  int cell_of(double v, double v0, int g) const {
    const int cc = (int)std::floor((v - v0) * inv_h);
    return cc < 0 ? 0 : (cc >= g ? g - 1 : cc);
  }
  void nn_add(uint32_t id) {
    const double* s = &verts[(size_t)id * 7];
    cells[(size_t)cell_of(s[1], y0, gy) * gx + cell_of(s[0], x0, gx)].push_back(id);
  }
  void setup_grid(size_t expected_vertices) {
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    x0 = c->geom.pos_x - 0.5 * c->geom.len_x;
    y0 = c->geom.pos_y - 0.5 * c->geom.len_y;
    h = std::sqrt(c->geom.len_x * c->geom.len_y * 3.0 / (double)std::max<size_t>(expected_vertices, 16));
    const double hmin = std::max(c->geom.len_x, c->geom.len_y) / 1024.0;
    if (h < hmin) h = hmin;
    inv_h = 1.0 / h;
    gx = std::max(1, (int)std::ceil(c->geom.len_x * inv_h));
    gy = std::max(1, (int)std::ceil(c->geom.len_y * inv_h));
    cells.assign((size_t)gx * gy, {});
  }
  // nn_->nearestK(q, k) under OMPL's SE3 distance, ascending (distance, id).  The distance is at least the planar one
  // and every cell of ring r is at least (r - 1) h away in the plane: the ring walk stops -- exactly -- once k
  // candidates are held and that bound reaches the k-th distance.
  void nn_nearest(const double* q, int k, std::vector<std::pair<double, uint32_t>>& out) const {
    out.clear();
    if (k <= 0) return;
    const int cx = cell_of(q[0], x0, gx), cy = cell_of(q[1], y0, gy);
    const int rmax = std::max(std::max(cx, gx - 1 - cx), std::max(cy, gy - 1 - cy));
    auto visit = [&](int xx, int yy) {
      for (uint32_t id : cells[(size_t)yy * gx + xx]) {
        const double* v = &verts[(size_t)id * 7];
        const double dx = v[0] - q[0], dy = v[1] - q[1], dz = v[2] - q[2];
        const double dp = std::sqrt(dx * dx + dy * dy + dz * dz);
        if ((int)out.size() == k && dp > out.front().first) continue;
        const double dq = std::fabs(v[3] * q[3] + v[4] * q[4] + v[5] * q[5] + v[6] * q[6]);
        const std::pair<double, uint32_t> cand{dp + (dq > 1.0 - 1e-9 ? 0.0 : std::acos(dq)), id};
        if ((int)out.size() < k) {
          out.push_back(cand);
          std::push_heap(out.begin(), out.end());
        } else if (cand < out.front()) {
          std::pop_heap(out.begin(), out.end());
          out.back() = cand;
          std::push_heap(out.begin(), out.end());
        }
      }
    };
    for (int r = 0; r <= rmax; ++r) {
      if ((int)out.size() == k && (double)(r - 1) * h >= out.front().first) break;
      for (int yy = cy - r; yy <= cy + r; ++yy) {
        if (yy < 0 || yy >= gy) continue;
        if (yy == cy - r || yy == cy + r) {
          for (int xx = std::max(cx - r, 0); xx <= std::min(cx + r, gx - 1); ++xx) visit(xx, yy);
        } else {
          if (cx - r >= 0) visit(cx - r, yy);
          if (r > 0 && cx + r < gx) visit(cx + r, yy);
        }
      }
    }
    std::sort_heap(out.begin(), out.end());
  }
  int eval_tasks(size_t n) {  // tasks -> states, valid
    hipStream_t st = c->stream;
    if (hipSetDevice(c->device) != hipSuccess) return ARTP_ERR_HIP;
    if (cap < n) {
      for (void* p : {(void*)d_tasks, (void*)d_states, (void*)d_valid})
        if (p) (void)hipFree(p);
      d_tasks = d_states = nullptr;
      d_valid = nullptr;
      cap = std::max<size_t>(2 * n, 1024);
      if (hipMalloc(reinterpret_cast<void**>(&d_tasks), cap * 15 * sizeof(double)) != hipSuccess ||
          hipMalloc(reinterpret_cast<void**>(&d_states), cap * 7 * sizeof(double)) != hipSuccess ||
          hipMalloc(reinterpret_cast<void**>(&d_valid), cap) != hipSuccess) {
        cap = 0;
        c->last_error = "chain staging allocation failed";
        return ARTP_ERR_HIP;
      }
    }
    states.resize(n * 7);
    valid.resize(n);
    if (hipMemcpyAsync(d_tasks, tasks.data(), n * 15 * sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess)
      return ARTP_ERR_HIP;
    hipLaunchKernelGGL(artp::chain_states_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st,
                       (const double*)d_tasks, (int)n, d_states);
    const int rc = artp_validate_states_dev(c, d_states, n, d_valid, nullptr);
    if (rc != ARTP_OK) return rc;
    if (hipMemcpyAsync(states.data(), d_states, n * 7 * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipMemcpyAsync(valid.data(), d_valid, n, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
      return ARTP_ERR_HIP;
    states_checked += n;
    return check_error_flag(c);
  }
  // PRMMotionCost::addValidMilestone (prm_motion_cost.cpp:325-390) for the vertex whose row is already in verts
  int add_valid_milestone(uint32_t m) {
    ++n_graph;  // :326 the milestone is a graph vertex first ...
    inserted[m] = 1;
    const int k = prm.k_neighbors ? (int)prm.k_neighbors : roadmap_kstar(n_graph);  // KStarStrategy at insertion time
    std::vector<std::pair<double, uint32_t>> nbrs;
    const double q[7] = {verts[(size_t)m * 7 + 0], verts[(size_t)m * 7 + 1], verts[(size_t)m * 7 + 2], verts[(size_t)m * 7 + 3],
                         verts[(size_t)m * 7 + 4], verts[(size_t)m * 7 + 5], verts[(size_t)m * 7 + 6]};
    nn_nearest(q, k, nbrs);  // :334 connectionStrategy_(m): m itself is not in nn_ yet
    if ((int)nbrs.size() > k_max) k_max = (int)nbrs.size();
    knn_rows.push_back(m);
    knn_rows.push_back((uint32_t)nbrs.size());
    for (const auto& nb : nbrs) knn_rows.push_back(nb.second);
    // :340-344 the interior states of every neighbour's chain, one device batch for the whole milestone
    std::vector<uint32_t> n_interp(nbrs.size());
    tasks.clear();
    for (size_t t = 0; t < nbrs.size(); ++t) {
      const double* b = &verts[(size_t)nbrs[t].second * 7];
      const double dx = b[0] - q[0], dy = b[1] - q[1];
      const double cnt = std::floor(std::sqrt(dx * dx + dy * dy) / 0.5);
      if (!(cnt >= 0.0) || cnt > 65536.0) {
        c->last_error = "a connection needs more than 65536 interpolation states (non-finite state?)";
        return ARTP_ERR_INVALID_ARG;
      }
      n_interp[t] = (uint32_t)cnt;
      const double div = 1.0 / (double)(n_interp[t] + 1);
      for (uint32_t step = 1; step <= n_interp[t]; ++step) {
        tasks.insert(tasks.end(), q, q + 7);
        tasks.insert(tasks.end(), b, b + 7);
        tasks.push_back((double)step * div);
      }
    }
    const size_t n_tasks = tasks.size() / 15;
    if (n_tasks) {
      const int rc = eval_tasks(n_tasks);
      if (rc != ARTP_OK) return rc;
    }
    size_t at = 0;
    for (size_t t = 0; t < nbrs.size(); ++t) {
      const uint32_t nb = nbrs[t].second;
      uint32_t prev = m;
      bool ok = true;
      for (uint32_t step = 0; step < n_interp[t]; ++step) {  // :353-371 the valid prefix stays in the graph
        if (ok && valid[at + step]) {
          const uint32_t v = (uint32_t)nv();
          verts.insert(verts.end(), states.begin() + (at + step) * 7, states.begin() + (at + step + 1) * 7);
          inserted.push_back(1);
          ++n_graph;
          eu.push_back(prev);
          ev.push_back(v);
          nn_add(v);  // :364 a neighbour target for every later milestone
          prev = v;
        } else {
          ok = false;
        }
      }
      at += n_interp[t];
      if (ok) {  // :345-348 direct edge, :373-377 last edge of a complete chain
        eu.push_back(prev);
        ev.push_back(nb);
      }
    }
    nn_add(m);  // :387 ... and a neighbour target last
    return ARTP_OK;
  }
};

// graph -> artp_roadmap: edges sorted by (u, v), u < v, one batched cost evaluation (updateEdges fills the weights of
// the reference's graph in one batch as well, prm_motion_cost.cpp:27-73); every sub-edge is direct
static int incremental_finish(IncrementalGraph& g, artp_roadmap* rm, const std::vector<uint8_t>* vertex_ok) {
  const size_t ne = g.eu.size(), nv = g.nv();
  // (min, max, creation direction is max -> min): an edge is created once, so (min, max) is unique
  std::vector<std::tuple<uint32_t, uint32_t, uint8_t>> e(ne);
  for (size_t i = 0; i < ne; ++i)
    e[i] = std::make_tuple(std::min(g.eu[i], g.ev[i]), std::max(g.eu[i], g.ev[i]), (uint8_t)(g.eu[i] > g.ev[i] ? 1 : 0));
  std::sort(e.begin(), e.end());
  e.erase(std::unique(e.begin(), e.end(), [](const auto& a, const auto& b) {
            return std::get<0>(a) == std::get<0>(b) && std::get<1>(a) == std::get<1>(b);
          }), e.end());
  rm->ctx = g.c;
  rm->params = g.prm;
  roadmap_fix_params(rm);
  rm->verts = g.verts;
  rm->k = std::max(g.k_max, 1);
  rm->knn.assign(nv * (size_t)rm->k, 0xffffffffu);
  rm->knn_dist.assign(nv * (size_t)rm->k, INFINITY);
  for (size_t at = 0; at < g.knn_rows.size();) {
    const uint32_t m = g.knn_rows[at], cnt = g.knn_rows[at + 1];
    for (uint32_t t = 0; t < cnt; ++t) {
      const uint32_t nb = g.knn_rows[at + 2 + t];
      rm->knn[(size_t)m * rm->k + t] = nb;
      const double *a = &g.verts[(size_t)m * 7], *b = &g.verts[(size_t)nb * 7];
      const double dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
      const double dq = std::fabs(a[3] * b[3] + a[4] * b[4] + a[5] * b[5] + a[6] * b[6]);
      rm->knn_dist[(size_t)m * rm->k + t] = std::sqrt(dx * dx + dy * dy + dz * dz) + (dq > 1.0 - 1e-9 ? 0.0 : std::acos(dq));
    }
    at += 2 + cnt;
  }
  const size_t n = e.size();
  rm->eu.resize(n);
  rm->ev.resize(n);
  rm->eflip.resize(n);
  for (size_t i = 0; i < n; ++i) {
    rm->eu[i] = std::get<0>(e[i]);
    rm->ev[i] = std::get<1>(e[i]);
    rm->eflip[i] = std::get<2>(e[i]);
  }
  rm->evalid.assign(n, 0);
  rm->einterp.assign(n, 0);
  rm->ecost.assign(n, 0.0);
  rm->eremoved.assign(n, 0);
  const int rc = roadmap_eval_edges_host(g.c, &rm->params, rm->verts, rm->eu.data(), rm->ev.data(), n, rm->evalid.data(),
                                         rm->einterp.data(), rm->ecost.data(), true, rm->eflip.data());
  if (rc != ARTP_OK) return rc;
  if (vertex_ok)
    for (size_t i = 0; i < n; ++i)
      if (!(*vertex_ok)[rm->eu[i]] || !(*vertex_ok)[rm->ev[i]]) rm->evalid[i] = 0;
  // the adjacency is part of the graph: built with it (OMPL's boost graph has it from the insertion on), not at the first
  // search -- 2 ms of the first solve at 10^4 vertices / 2.6 10^5 edges (round 5)
  roadmap_build_csr(rm);
  rm->emotion_dirty = true;
  rm->d_graph_dirty = true;
  rm->d_graph_ne = 0;
  rm->d_edge_states_stale = true;
  return ARTP_OK;
}

// sampleGraph's loop (prm_motion_cost.cpp:171-194) over the accepted states of the sample stream from index `first` on:
// milestones are added while the graph has fewer than max_vertices vertices (chain vertices count) and fewer than
// max_n_edges sub-edges.  The stream position behind the last sample consumed goes to *next_out.
static int incremental_sample_graph(IncrementalGraph& g, size_t max_vertices, uint64_t first, uint64_t* next_out,
                                    uint64_t* n_reweights, uint64_t* budget_flags) {
  artp_ctx* c = g.c;
  const artp_roadmap_params& prm = g.prm;
  hipStream_t st = c->stream;
  const size_t batch = 4096;
  double *d_batch = nullptr, *d_compact = nullptr, *d_verts = nullptr;
  uint8_t* d_valid = nullptr;
  uint32_t* d_idx = nullptr;
  uint64_t* d_cnt = nullptr;
  auto cleanup = [&]() {
    for (void* p : {(void*)d_batch, (void*)d_compact, (void*)d_valid, (void*)d_idx, (void*)d_cnt, (void*)d_verts})
      if (p) (void)hipFree(p);
  };
  RM_HIP(hipSetDevice(c->device));
  RM_HIP(hipMalloc(reinterpret_cast<void**>(&d_batch), batch * 7 * sizeof(double)));
  RM_HIP(hipMalloc(reinterpret_cast<void**>(&d_compact), batch * 7 * sizeof(double)));
  RM_HIP(hipMalloc(reinterpret_cast<void**>(&d_valid), batch));
  RM_HIP(hipMalloc(reinterpret_cast<void**>(&d_idx), batch * sizeof(uint32_t)));
  RM_HIP(hipMalloc(reinterpret_cast<void**>(&d_cnt), 2 * sizeof(uint64_t)));
  const size_t R = (prm.density_map && prm.density_params && prm.recompute_density_after_n_samples)
                       ? prm.recompute_density_after_n_samples : 0;
  std::vector<double> acc(batch * 7);
  std::vector<uint32_t> idx(batch);
  size_t q_n = 0, q_at = 0;
  uint64_t base = first, next = first;
  size_t n_proc = 0;  // counts from 0 in every sampleGraph call (:169)
  int empty_rounds = 0;
  const auto t_start = std::chrono::steady_clock::now();
  while (g.n_graph < max_vertices && (!prm.max_n_edges || g.eu.size() < prm.max_n_edges)) {
    if (q_at == q_n) {
      base = next;
      RM_TRY(artp_sample_and_validate_dev(c, prm.seed, base, batch, d_batch, d_valid, nullptr));
      RM_TRY(artp_compact_valid_dev(c, d_batch, d_valid, batch, d_compact, d_cnt));
      RM_TRY(artp_compact_valid_indices_dev(c, d_valid, batch, d_idx, d_cnt + 1));
      uint64_t got = 0;
      RM_HIP(hipMemcpyAsync(&got, d_cnt, sizeof(got), hipMemcpyDeviceToHost, st));
      RM_HIP(hipStreamSynchronize(st));
      if (got) {
        RM_HIP(hipMemcpyAsync(acc.data(), d_compact, (size_t)got * 7 * sizeof(double), hipMemcpyDeviceToHost, st));
        RM_HIP(hipMemcpyAsync(idx.data(), d_idx, (size_t)got * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        RM_HIP(hipStreamSynchronize(st));
      }
      q_n = (size_t)got;
      q_at = 0;
      next = base + batch;
      if (!got) {
        if (++empty_rounds > 64) {
          c->last_error = "sampler produced too few valid states for the requested roadmap";
          cleanup();
          return ARTP_ERR_CAPACITY;
        }
        continue;
      }
      empty_rounds = 0;
    }
    const uint32_t m = (uint32_t)g.nv();
    g.verts.insert(g.verts.end(), acc.begin() + q_at * 7, acc.begin() + (q_at + 1) * 7);
    g.inserted.push_back(0);
    const uint64_t consumed = base + idx[q_at];
    ++q_at;
    if (q_at == q_n) next = base + batch; else next = consumed + 1;
    RM_TRY(g.add_valid_milestone(m));
    if (R && g.n_graph / R > n_proc) {
      // :190-193 Map::reApplyPreprocessing(): the distribution follows the inverse density of ALL graph vertices; the
      // stream continues right behind the sample that gave this milestone
      const size_t nvx = g.nv() - 2;
      if (d_verts) (void)hipFree(d_verts);
      d_verts = nullptr;
      RM_HIP(hipMalloc(reinterpret_cast<void**>(&d_verts), nvx * 7 * sizeof(double)));
      RM_HIP(hipMemcpyAsync(d_verts, g.verts.data() + 14, nvx * 7 * sizeof(double), hipMemcpyHostToDevice, st));
      RM_TRY(artp_preprocessed_reweight_dev(c, prm.density_map, prm.density_params, d_verts, nvx, 1));
      ++n_proc;
      ++*n_reweights;
      next = consumed + 1;
      q_at = q_n = 0;
    }
    if (prm.max_sample_time > 0.0 &&
        std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count() > prm.max_sample_time) {
      *budget_flags |= 1u;  // "Reached sample timer limit." (:177-185)
      break;
    }
  }
  if (prm.max_n_edges && g.eu.size() >= prm.max_n_edges) *budget_flags |= 2u;
  *next_out = next;
  cleanup();
  return ARTP_OK;
}

static int roadmap_build_incremental(artp_ctx* c, const artp_roadmap_params* prm, const double* start7,
                                     const double* goal7, artp_roadmap** out) {
  *out = nullptr;
  auto cleanup = []() {};
  {
    double sg[14];
    std::memcpy(sg, start7, 7 * sizeof(double));
    std::memcpy(sg + 7, goal7, 7 * sizeof(double));
    uint8_t ok[2] = {0, 0};
    RM_TRY(artp_validate_states(c, sg, 2, ok, nullptr));
    RM_HIP(hipStreamSynchronize(c->stream));
    if (!ok[0] || !ok[1]) {
      c->last_error = !ok[0] ? "start state is not valid" : "goal state is not valid";
      return ARTP_ERR_INVALID_ARG;
    }
  }
  IncrementalGraph g;
  g.c = c;
  g.prm = *prm;
  g.verts.assign(start7, start7 + 7);
  g.verts.insert(g.verts.end(), goal7, goal7 + 7);
  g.inserted.assign(2, 0);
  g.setup_grid(prm->n_milestones);
  uint64_t next = prm->first_index, n_reweights = 0, budget_flags = 0;
  RM_TRY(incremental_sample_graph(g, prm->n_milestones, prm->first_index, &next, &n_reweights, &budget_flags));
  if (g.n_graph < 1) {
    c->last_error = "the sampling-time budget ended before the first milestone";
    return ARTP_ERR_CAPACITY;
  }
  // baseSolve: start and goal become milestones of the finished roadmap (:447-470)
  RM_TRY(g.add_valid_milestone(0));
  RM_TRY(g.add_valid_milestone(1));
  auto rm = new artp_roadmap();
  const int rc = incremental_finish(g, rm, nullptr);
  if (rc != ARTP_OK) {
    delete rm;
    return rc;
  }
  rm->samples_drawn = next - prm->first_index;
  rm->n_reweights = n_reweights;
  rm->budget_flags = budget_flags;
  *out = rm;
  return ARTP_OK;
}

// artp_roadmap_grow for construction 1: sampleGraph called again on the kept graph.  Vertices the current map
// invalidated stay in the vertex list (ids are stable) but leave the nearest-neighbour structure, and their edges are
// marked invalid; n_more = how many more graph vertices (chain vertices included) the budget allows.
static int roadmap_grow_incremental(artp_roadmap* rm, uint64_t n_more, uint64_t out[2]) {
  artp_ctx* c = rm->ctx;
  auto cleanup = []() {};
  const size_t nv = rm->nv();
  std::vector<uint8_t> vok(nv, 0);
  RM_TRY(artp_validate_states(c, rm->verts.data(), nv, vok.data(), nullptr));
  RM_HIP(hipStreamSynchronize(c->stream));
  if (!vok[0] || !vok[1]) {
    c->last_error = !vok[0] ? "start state is not valid" : "goal state is not valid";
    return ARTP_ERR_INVALID_ARG;
  }
  IncrementalGraph g;
  g.c = c;
  g.prm = rm->params;
  g.verts = rm->verts;
  g.inserted.assign(nv, 1);
  g.eu = rm->eu;
  g.ev = rm->ev;
  for (size_t e = 0; e < g.eu.size() && e < rm->eflip.size(); ++e)
    if (rm->eflip[e]) std::swap(g.eu[e], g.ev[e]);  // creation direction (incremental_finish sorts it out again)
  g.n_graph = nv;
  g.k_max = rm->k;
  g.setup_grid(nv + (size_t)n_more);
  size_t kept = 0;
  for (uint32_t v = 0; v < nv; ++v)
    if (vok[v]) {
      g.nn_add(v);
      ++kept;
    }
  for (uint32_t m = 0; m < nv; ++m) {  // the neighbour lists recorded so far
    uint32_t cnt = 0;
    while ((int)cnt < rm->k && rm->knn[(size_t)m * rm->k + cnt] != 0xffffffffu) ++cnt;
    if (!cnt) continue;
    g.knn_rows.push_back(m);
    g.knn_rows.push_back(cnt);
    for (uint32_t t = 0; t < cnt; ++t) g.knn_rows.push_back(rm->knn[(size_t)m * rm->k + t]);
  }
  uint64_t next = rm->params.first_index + rm->samples_drawn, n_reweights = 0, budget_flags = 0;
  RM_TRY(incremental_sample_graph(g, nv + (size_t)n_more, next, &next, &n_reweights, &budget_flags));
  vok.resize(g.nv(), 1);
  auto fresh = new artp_roadmap();
  const int rc = incremental_finish(g, fresh, &vok);
  if (rc != ARTP_OK) {
    delete fresh;
    return rc;
  }
  // The reference removes an edge the lazy path check rejected from g_ for good (prm_motion_cost.cpp:652-660): the
  // removals of the graph so far carry over.  Edge identity is stable -- (min, max) vertex ids, both lists sorted.
  {
    size_t o = 0;
    const size_t no = rm->eu.size();
    for (size_t e = 0; e < fresh->eu.size(); ++e) {
      while (o < no && (rm->eu[o] < fresh->eu[e] || (rm->eu[o] == fresh->eu[e] && rm->ev[o] < fresh->ev[e]))) ++o;
      if (o < no && rm->eu[o] == fresh->eu[e] && rm->ev[o] == fresh->ev[e] && rm->eremoved[o]) fresh->eremoved[e] = 1;
    }
  }
  fresh->vinvalid.assign(g.nv(), 0);
  for (size_t v = 0; v < g.nv(); ++v) fresh->vinvalid[v] = vok[v] ? 0 : 1;
  fresh->samples_drawn = next - rm->params.first_index;
  fresh->n_reweights = rm->n_reweights + n_reweights;
  fresh->budget_flags = budget_flags;
  fresh->params.n_milestones = (uint32_t)std::min<size_t>(nv + (size_t)n_more, 0xffffffffu);
  std::swap(*rm, *fresh);
  roadmap_fix_params(rm);
  roadmap_fix_params(fresh);
  artp_roadmap_destroy(fresh);
  if (out) {
    out[0] = kept;
    out[1] = nv - kept;
  }
  return ARTP_OK;
}

int artp_roadmap_build(artp_ctx* c, const artp_roadmap_params* prm, const double* start7, const double* goal7,
                       artp_roadmap** out) {
  if (!c || !prm || !start7 || !goal7 || !out || prm->n_milestones < 1 || prm->objective < 0 ||
      prm->objective > 2 || !(prm->max_lon_vel > 0) || !(prm->max_lat_vel > 0) || !(prm->max_ang_vel > 0) ||
      prm->construction < 0 || prm->construction > 2)
    return ARTP_ERR_INVALID_ARG;
  if (prm->construction == 1) return roadmap_build_incremental(c, prm, start7, goal7, out);
  return roadmap_build_impl(c, prm, start7, goal7, nullptr, 0, prm->n_milestones, prm->first_index, out);
}

// PRMMotionCostMaintainer::sampleGraph keeps adding milestones to the kept graph between queries
// (prm_motion_cost.cpp:145-219), LazyPRM* grows its roadmap for as long as it plans.  Batched: the milestones
// still valid on the CURRENT map stay, n_more new ones are drawn where the sample stream left off, and the
// connection rule (k grows with the vertex count, KStarStrategy) + every edge verdict are recomputed for the
// whole set -- 10^4 vertices / 10^5 edges are one small batch, cheaper than bookkeeping which edges survive.
int artp_roadmap_grow(artp_roadmap* rm, uint64_t n_more, uint64_t out[2]) {
  if (!rm) return ARTP_ERR_INVALID_ARG;
  if (rm->params.construction == 1) return roadmap_grow_incremental(rm, n_more, out);
  artp_ctx* c = rm->ctx;
  const size_t nv = rm->nv();
  std::vector<uint8_t> vok(nv, 0);
  int rc = artp_validate_states(c, rm->verts.data(), nv, vok.data(), nullptr);
  if (rc != ARTP_OK) return rc;
  if (!vok[0] || !vok[1]) {
    c->last_error = !vok[0] ? "start state is not valid" : "goal state is not valid";
    return ARTP_ERR_INVALID_ARG;
  }
  std::vector<double> keep;
  keep.reserve((nv - 2) * 7);
  for (size_t v = 2; v < nv; ++v)
    if (vok[v]) keep.insert(keep.end(), rm->verts.begin() + v * 7, rm->verts.begin() + (v + 1) * 7);
  const size_t n_keep = keep.size() / 7;
  if (n_keep + n_more < 1) return ARTP_ERR_INVALID_ARG;
  artp_roadmap* fresh = nullptr;
  const uint64_t drawn_before = rm->samples_drawn;
  rc = roadmap_build_impl(c, &rm->params, rm->verts.data(), rm->verts.data() + 7, keep.data(), n_keep, (size_t)n_more,
                          rm->params.first_index + drawn_before, &fresh);
  if (rc != ARTP_OK) return rc;
  fresh->samples_drawn += drawn_before;
  fresh->n_reweights += rm->n_reweights;
  std::swap(*rm, *fresh);
  roadmap_fix_params(rm);
  roadmap_fix_params(fresh);
  artp_roadmap_destroy(fresh);
  if (out) {
    out[0] = n_keep;
    out[1] = (nv - 2) - n_keep;  // milestones the current map invalidated
  }
  return ARTP_OK;
}

// LazyPRMStarMinUpdate keeps its roadmap across map updates and re-checks what changed
// (lazy_prm_star_min_update.cpp:18-217).  Batched: after the map changed, re-validate EVERY vertex and
// re-evaluate EVERY edge against the current layers (10^4 vertices + 10^5 edges are one small batch);
// edges with an invalid endpoint go, earlier lazy removals are forgotten.
int artp_roadmap_revalidate(artp_roadmap* rm, uint64_t out[4]) {
  if (!rm) return ARTP_ERR_INVALID_ARG;
  artp_ctx* c = rm->ctx;
  const size_t nv = rm->nv(), ne = rm->eu.size();
  std::vector<uint8_t> vok(nv, 0);
  int rc = artp_validate_states(c, rm->verts.data(), nv, vok.data(), nullptr);
  if (rc != ARTP_OK) return rc;
  uint64_t before = 0, after = 0, vbad = 0;
  for (size_t e = 0; e < ne; ++e) before += rm->evalid[e] ? 1 : 0;
  if (ne) {
    if (rm->d_edge_states_stale) {
      // the query vertices changed since the states were staged: vertex states and edge end points go up
      // (2 MB for 10^4 vertices), the 19 MB of endpoint states are gathered on the device
      double* d_v = nullptr;
      uint32_t* d_uv = nullptr;
      bool okk = hipSetDevice(c->device) == hipSuccess;
      if (okk && rm->d_edge_cap < ne) {
        if (rm->d_edge_states) (void)hipFree(rm->d_edge_states);
        rm->d_edge_states = nullptr;
        okk = hipMalloc(reinterpret_cast<void**>(&rm->d_edge_states), 2 * ne * 7 * sizeof(double)) == hipSuccess;
        rm->d_edge_cap = okk ? ne : 0;
      }
      okk = okk && hipMalloc(reinterpret_cast<void**>(&d_v), nv * 7 * sizeof(double)) == hipSuccess &&
            hipMalloc(reinterpret_cast<void**>(&d_uv), 2 * ne * sizeof(uint32_t)) == hipSuccess &&
            hipMemcpyAsync(d_v, rm->verts.data(), nv * 7 * sizeof(double), hipMemcpyHostToDevice, c->stream) == hipSuccess &&
            hipMemcpyAsync(d_uv, rm->eu.data(), ne * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream) == hipSuccess &&
            hipMemcpyAsync(d_uv + ne, rm->ev.data(), ne * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream) == hipSuccess;
      std::vector<uint32_t> src, dst;  // end points in cost direction (eflip)
      if (okk && !rm->eflip.empty()) {
        src = rm->eu;
        dst = rm->ev;
        for (size_t e = 0; e < ne; ++e)
          if (rm->eflip[e]) std::swap(src[e], dst[e]);
        okk = hipMemcpyAsync(d_uv, src.data(), ne * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream) == hipSuccess &&
              hipMemcpyAsync(d_uv + ne, dst.data(), ne * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream) == hipSuccess;
      }
      if (okk) {
        hipLaunchKernelGGL(artp::gather_edge_states_uv_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, c->stream,
                           (const double*)d_v, (const uint32_t*)d_uv, (const uint32_t*)(d_uv + ne), ne, rm->d_edge_states,
                           rm->d_edge_states + ne * 7);
        okk = hipGetLastError() == hipSuccess && hipStreamSynchronize(c->stream) == hipSuccess;
      }
      if (d_v) (void)hipFree(d_v);
      if (d_uv) (void)hipFree(d_uv);
      if (!okk) {
        c->last_error = "staging the edge states failed";
        return ARTP_ERR_HIP;
      }
      rm->d_edge_states_stale = false;
    }
    // construction 2 holds direct edges of unknown validity (the lazy path check's business); the sub-edges of
    // construction 1 are shorter than 0.5 m, so the rule passes them as they are and only applies to query edges
    rc = roadmap_eval_edges_dev(c, &rm->params, rm->d_edge_states, rm->d_edge_states + ne * 7, ne, rm->evalid.data(),
                                rm->einterp.data(), rm->ecost.data(), rm->params.construction == 2);
    if (rc != ARTP_OK) return rc;
  }
  for (size_t v = 0; v < nv; ++v) vbad += vok[v] ? 0 : 1;
  for (size_t e = 0; e < ne; ++e) {
    if (!vok[rm->eu[e]] || !vok[rm->ev[e]]) rm->evalid[e] = 0;
    after += rm->evalid[e] ? 1 : 0;
  }
  std::fill(rm->eremoved.begin(), rm->eremoved.end(), 0);
  rm->vinvalid.assign(nv, 0);
  for (size_t v = 0; v < nv; ++v) rm->vinvalid[v] = vok[v] ? 0 : 1;
  rm->csr_dirty = true;
  rm->emotion_dirty = true;
  rm->d_graph_dirty = true;
  if (out) {
    out[0] = vbad;
    out[1] = before;
    out[2] = after;
    out[3] = (uint64_t)((vok[0] ? 1 : 0) | (vok[1] ? 2 : 0));  // bit 0: start still valid, bit 1: goal
  }
  return ARTP_OK;
}

// New start / goal on the kept roadmap (every OMPL query adds its start and goal as milestones,
// prm_motion_cost.cpp:452-476): vertices 0 and 1 are replaced and connected to their k nearest vertices.
int artp_roadmap_set_query(artp_roadmap* rm, const double* start7, const double* goal7) {
  if (!rm || !start7 || !goal7) return ARTP_ERR_INVALID_ARG;
  artp_ctx* c = rm->ctx;
  double old_sg[14];
  {
    double sg[14];
    std::memcpy(sg, start7, 7 * sizeof(double));
    std::memcpy(sg + 7, goal7, 7 * sizeof(double));
    uint8_t ok[2] = {0, 0};
    const int rc = artp_validate_states(c, sg, 2, ok, nullptr);
    if (rc != ARTP_OK) return rc;
    if (!ok[0] || !ok[1]) {
      c->last_error = !ok[0] ? "start state is not valid" : "goal state is not valid";
      return ARTP_ERR_INVALID_ARG;
    }
    std::memcpy(old_sg, &rm->verts[0], sizeof(old_sg));
    std::memcpy(&rm->verts[0], sg, sizeof(sg));
  }
  const size_t nv = rm->nv();
  const int k = rm->k;
  // everything below either commits completely or leaves the roadmap as it was (a failing edge evaluation must
  // not leave new query states with the old edge prefix)
  const std::vector<uint32_t> old_knn(rm->knn.begin(), rm->knn.begin() + 2 * (size_t)k);
  const std::vector<double> old_knn_dist(rm->knn_dist.begin(), rm->knn_dist.begin() + 2 * (size_t)k);
  // edges are sorted by (u, v) with u < v: everything that touches vertex 0 or 1 is a prefix
  size_t first_keep = 0;
  while (first_keep < rm->eu.size() && rm->eu[first_keep] < 2) ++first_keep;
  auto arc = [](const double* a, const double* b) {
    const double dq = std::fabs(a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3]);
    return dq > 1.0 - 1e-9 ? 0.0 : std::acos(dq);
  };
  std::vector<uint32_t> nu, nvx;
  for (uint32_t q = 0; q < 2; ++q) {
    const double* a = &rm->verts[(size_t)q * 7];
    std::vector<std::pair<double, uint32_t>> cand;
    cand.reserve(nv);
    for (uint32_t j = 0; j < nv; ++j) {
      if (j == q) continue;
      if (j >= 2 && j < rm->vinvalid.size() && rm->vinvalid[j]) continue;  // invalidated by the current map
      const double* b = &rm->verts[(size_t)j * 7];
      const double dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
      cand.push_back({std::sqrt(dx * dx + dy * dy + dz * dz) + arc(a + 3, b + 3), j});
    }
    const size_t kk = std::min<size_t>((size_t)k, cand.size());
    std::partial_sort(cand.begin(), cand.begin() + kk, cand.end());
    std::vector<uint32_t> nb;
    for (size_t t = 0; t < (size_t)k; ++t) {
      rm->knn[(size_t)q * k + t] = t < kk ? cand[t].second : 0xffffffffu;
      rm->knn_dist[(size_t)q * k + t] = t < kk ? cand[t].first : INFINITY;
      if (t < kk) nb.push_back(cand[t].second);
    }
    std::sort(nb.begin(), nb.end());
    for (uint32_t j : nb) {
      nu.push_back(std::min(q, j));
      nvx.push_back(std::max(q, j));
    }
  }
  // (0, 1) belongs to the u = 0 block: re-sort the new prefix by (u, v) and drop duplicates
  std::vector<std::pair<uint32_t, uint32_t>> pre;
  for (size_t e = 0; e < nu.size(); ++e) pre.push_back({nu[e], nvx[e]});
  std::sort(pre.begin(), pre.end());
  pre.erase(std::unique(pre.begin(), pre.end()), pre.end());
  const size_t np = pre.size();
  std::vector<uint32_t> pu(np), pv(np), pinterp(np);
  std::vector<uint8_t> pvalid(np);
  std::vector<double> pcost(np);
  for (size_t e = 0; e < np; ++e) {
    pu[e] = pre[e].first;
    pv[e] = pre[e].second;
  }
  const int rc = roadmap_eval_edges_host(c, &rm->params, rm->verts, pu.data(), pv.data(), np, pvalid.data(),
                                         pinterp.data(), pcost.data(), rm->params.construction == 2);
  if (rc != ARTP_OK) {
    std::memcpy(&rm->verts[0], old_sg, sizeof(old_sg));
    std::copy(old_knn.begin(), old_knn.end(), rm->knn.begin());
    std::copy(old_knn_dist.begin(), old_knn_dist.end(), rm->knn_dist.begin());
    return rc;
  }
  auto splice = [&](auto& vec, const auto& head) {
    vec.erase(vec.begin(), vec.begin() + first_keep);
    vec.insert(vec.begin(), head.begin(), head.end());
  };
  splice(rm->eu, pu);
  splice(rm->ev, pv);
  splice(rm->evalid, pvalid);
  splice(rm->einterp, pinterp);
  splice(rm->ecost, pcost);
  std::vector<uint8_t> zeros(np, 0);
  splice(rm->eremoved, zeros);
  if (!rm->eflip.empty()) splice(rm->eflip, zeros);  // query vertex (the smaller id) -> neighbour: the order it is added in
  rm->csr_dirty = true;
  rm->emotion_dirty = true;
  rm->d_edge_states_stale = true;
  rm->d_graph_ne = 0;  // the edge list changed: the device copy is rebuilt at the next search
  return ARTP_OK;
}

// Planner::getSolutionPath simplifies the solution (params.planner.simplify_solution, planner.cpp:266-280, OMPL's
// randomised PathSimplifier::reduceVertices / shortcutPath).  Batched and deterministic instead: EVERY pair
// (i, j) of path states is tried as a shortcut in one batch -- the 0.5 m interpolation rule, the discrete
// motion validator and the objective's cost -- and the cheapest chain of valid shortcuts from the first to the
// last state is taken (a shortest path in a DAG).  Never worse than the input path under the objective.
int artp_roadmap_simplify_path(artp_roadmap* rm, const double* path_se3, size_t n, double* out_se3, size_t* n_out,
                               double* cost) {
  if (!rm || !path_se3 || !out_se3 || !n_out || n < 1) return ARTP_ERR_INVALID_ARG;
  artp_ctx* c = rm->ctx;
  if (n <= 2) {
    std::memcpy(out_se3, path_se3, n * 7 * sizeof(double));
    *n_out = n;
  }
  std::vector<double> verts(path_se3, path_se3 + n * 7);
  std::vector<uint32_t> eu, ev;
  for (uint32_t i = 0; i + 1 < n; ++i)
    for (uint32_t j = i + 1; j < n; ++j) {
      eu.push_back(i);
      ev.push_back(j);
    }
  const size_t ne = eu.size();
  std::vector<uint8_t> ok1(ne, 0), ok2(ne, 0);
  std::vector<uint32_t> ni(ne, 0);
  std::vector<double> ec(ne, 0.0);
  if (ne) {
    // shortcut candidates are path segments: the learned objective prices them the way motionCost does
    // (max_query_edge_length, motion_cost_objective.cpp:41-42); the 0.5 m rule still decides their validity
    const double mq = rm->params.max_query_edge_length > 0.0 ? rm->params.max_query_edge_length : 0.5;
    int rc = roadmap_eval_edges_host(c, &rm->params, verts, eu.data(), ev.data(), ne, ok1.data(), ni.data(), ec.data(), false,
                                     nullptr, rm->params.objective == 2 ? mq : 0.0);
    if (rc != ARTP_OK) return rc;
    std::vector<double> s1(ne * 7), s2(ne * 7);
    for (size_t e = 0; e < ne; ++e) {
      std::memcpy(&s1[e * 7], &verts[(size_t)eu[e] * 7], 7 * sizeof(double));
      std::memcpy(&s2[e * 7], &verts[(size_t)ev[e] * 7], 7 * sizeof(double));
    }
    rc = artp_check_motions(c, s1.data(), s2.data(), ne, ok2.data());
    if (rc != ARTP_OK) return rc;
    (void)hipStreamSynchronize(c->stream);
  }
  // DAG shortest path 0 -> n-1 (edges only go forward)
  std::vector<double> best(n, INFINITY);
  std::vector<uint32_t> from(n, 0xffffffffu);
  best[0] = 0.0;
  size_t e = 0;
  for (uint32_t i = 0; i + 1 < n; ++i)
    for (uint32_t j = i + 1; j < n; ++j, ++e) {
      if (!ok1[e] || !ok2[e] || !std::isfinite(ec[e]) || !std::isfinite(best[i])) continue;
      if (best[i] + ec[e] < best[j]) {
        best[j] = best[i] + ec[e];
        from[j] = i;
      }
    }
  if (n > 1 && !std::isfinite(best[n - 1])) {  // the input path itself does not pass: hand it back unchanged
    std::memcpy(out_se3, path_se3, n * 7 * sizeof(double));
    *n_out = n;
    if (cost) *cost = INFINITY;
    return ARTP_OK;
  }
  std::vector<uint32_t> keep;
  for (uint32_t v = (uint32_t)n - 1; v != 0xffffffffu; v = from[v]) keep.push_back(v);
  std::reverse(keep.begin(), keep.end());
  for (size_t k = 0; k < keep.size(); ++k) std::memcpy(out_se3 + k * 7, &verts[(size_t)keep[k] * 7], 7 * sizeof(double));
  *n_out = keep.size();
  if (cost) *cost = best[n - 1];
  return ARTP_OK;
}

int artp_roadmap_stats(const artp_roadmap* rm, uint64_t out[8]) {
  if (!rm || !out) return ARTP_ERR_INVALID_ARG;
  uint64_t nvalid = 0, nrem = 0;
  for (size_t e = 0; e < rm->eu.size(); ++e) {
    nvalid += rm->evalid[e] ? 1 : 0;
    nrem += rm->eremoved[e] ? 1 : 0;
  }
  out[0] = rm->nv();
  out[1] = rm->eu.size();
  out[2] = nvalid;
  out[3] = nrem;
  out[4] = (uint64_t)rm->k;
  out[5] = rm->samples_drawn;
  out[6] = rm->n_reweights;
  out[7] = rm->budget_flags;
  return ARTP_OK;
}

int artp_roadmap_export(const artp_roadmap* rm, double* verts, uint32_t* knn, double* knn_dist, uint32_t* edges_uv,
                        uint8_t* edge_valid, uint32_t* edge_interp, double* edge_cost, uint8_t* edge_removed) {
  if (!rm) return ARTP_ERR_INVALID_ARG;
  const size_t ne = rm->eu.size();
  if (verts) std::memcpy(verts, rm->verts.data(), rm->verts.size() * sizeof(double));
  if (knn) std::memcpy(knn, rm->knn.data(), rm->knn.size() * sizeof(uint32_t));
  if (knn_dist) std::memcpy(knn_dist, rm->knn_dist.data(), rm->knn_dist.size() * sizeof(double));
  if (edges_uv)
    for (size_t e = 0; e < ne; ++e) {
      edges_uv[2 * e] = rm->eu[e];
      edges_uv[2 * e + 1] = rm->ev[e];
    }
  if (edge_valid) std::memcpy(edge_valid, rm->evalid.data(), ne);
  if (edge_interp) std::memcpy(edge_interp, rm->einterp.data(), ne * sizeof(uint32_t));
  if (edge_cost) std::memcpy(edge_cost, rm->ecost.data(), ne * sizeof(double));
  if (edge_removed) std::memcpy(edge_removed, rm->eremoved.data(), ne);
  return ARTP_OK;
}

// from this vertex count on the search runs on the device (host A*: 1.3 ms at 10^4 vertices, 41 ms at 10^5)
#define ARTP_SSSP_MIN_VERTICES 30000

namespace {

// ---- the lazy path check of the reference planners, replayed from cached verdicts ----------------------------------
// constructSolution (prm_motion_cost.cpp:536-673, lazy_prm_star_min_update.cpp:619-747): search, walk the path from the
// goal backwards, checkMotion every edge not known to be valid, remove the FIRST invalid one, search again.  On the
// reference-order LazyPRM* graph (10^4 milestones, 265 000 direct edges of unknown validity) that is 216 rounds of a
// search and a validity batch: 181 ms when every round is a host A* and a device call.  Two things make it 10x faster
// with the SAME removals in the SAME order and the same path:
//  * the search is a shortest-path TREE from the start that is repaired, not recomputed, when a path edge goes: only
//    the subtree behind the removed edge is re-settled (Ramalingam-Reps deletion; the removed edge lies on the path to
//    the goal, usually close to it: the reference removes the invalid edge nearest the goal);
//  * a verdict is a property of (edge, direction, map): once a few rounds have shown that the graph needs its lazy
//    checks, EVERY usable edge is checked in both directions in ONE batch (2 x 265 000 motions = 26 M states: 6 ms on
//    this GPU) and the rounds after that read the table.  The removal rule itself is untouched: only the first invalid
//    edge of the CURRENT path is removed, whatever else the table knows.
struct LazyTree {
  artp_roadmap* rm;
  std::vector<double> dist;
  std::vector<uint32_t> pred, pred_edge;
  std::vector<uint8_t> in_s;
  std::vector<uint32_t> first_child, next_sib, prev_sib;  // the tree's child lists (0xffffffff = none)
  std::vector<uint32_t> sub;
  std::vector<uint8_t> rel;  // empty = every vertex; else the vertices that can still lie on a start-goal path (restrict)
  const double* to_other = nullptr;  // with rel: a lower bound of every vertex's distance to the terminal that is NOT the
                                     // root (consistent: exact distances of an earlier, larger graph), and
  double cost_max = INFINITY;        // the cost no start-goal path of this query will exceed (C* and a rounding margin)
  // The tree hangs from one terminal (0 = start, 1 = goal) and the path is read off at the other.  Which one is a matter
  // of cost only -- the removed edge's subtree is what a repair re-settles, and that is small when the edge lies FAR
  // from the root: edges that fail near the start want the root at the goal, and the other way round.
  uint32_t root = 0;
  using Item = std::pair<double, uint32_t>;
  // everything the searches touch per adjacency slot is contiguous (adj, adjw); dist / pred of 10^4 vertices stay in
  // the host's L1 / L2 (three random reads per slot into the per-edge arrays made a repair 10x slower)
  bool usable(uint32_t e) const {
    const double w = rm->ecost[e];
    return rm->evalid[e] && !rm->eremoved[e] && std::isfinite(w) && w >= 0.0;
  }
  void drop_edge(uint32_t e) {  // the slots of a removed edge stop being usable
    for (const uint32_t v : {rm->eu[e], rm->ev[e]})
      for (uint32_t a = rm->row[v]; a < rm->row[v + 1]; ++a)
        if (rm->adj_edge[a] == e) rm->adjw[a] = INFINITY;
  }
  void full(uint32_t new_root) {
    const size_t nv = rm->nv();
    root = new_root;
    dist.assign(nv, INFINITY);
    pred.assign(nv, 0xffffffffu);
    pred_edge.assign(nv, 0xffffffffu);
    in_s.assign(nv, 0);
    std::priority_queue<Item, std::vector<Item>, std::greater<Item>> open;
    dist[root] = 0.0;
    open.push({0.0, root});
    const uint32_t* adj = rm->adj.data();
    const double* adjw = rm->adjw.data();
    while (!open.empty()) {
      const Item it = open.top();
      open.pop();
      const uint32_t u = it.second;
      if (it.first > dist[u]) continue;
      const double du = dist[u];
      for (uint32_t a = rm->row[u], a1 = rm->row[u + 1]; a < a1; ++a) {
        const uint32_t v = adj[a];
        const double nd = du + adjw[a];  // +inf for an unusable slot: never an improvement
        if (nd < dist[v] && (!to_other || (rel[v] && nd + to_other[v] <= cost_max))) {
          dist[v] = nd;
          pred[v] = u;
          pred_edge[v] = rm->adj_edge[a];
          open.push({nd, v});
        }
      }
    }
    if (to_other)
      for (uint32_t v = 0; v < (uint32_t)nv; ++v)
        if (rel[v] && v != root && pred[v] == 0xffffffffu) rel[v] = 0;  // out of reach from this side as well
    first_child.assign(nv, 0xffffffffu);
    next_sib.assign(nv, 0xffffffffu);
    prev_sib.assign(nv, 0xffffffffu);
    for (uint32_t v = 0; v < (uint32_t)nv; ++v)
      if (pred[v] != 0xffffffffu) link(v);
  }
  // distances to `root` over the usable slots (the graph is undirected: this is also the distance FROM root)
  void distances_from(uint32_t root, std::vector<double>* out) const {
    const size_t nv = rm->nv();
    out->assign(nv, INFINITY);
    std::priority_queue<Item, std::vector<Item>, std::greater<Item>> open;
    (*out)[root] = 0.0;
    open.push({0.0, root});
    const uint32_t* adj = rm->adj.data();
    const double* adjw = rm->adjw.data();
    double* d = out->data();
    while (!open.empty()) {
      const Item it = open.top();
      open.pop();
      const uint32_t u = it.second;
      if (it.first > d[u]) continue;
      for (uint32_t a = rm->row[u], a1 = rm->row[u + 1]; a < a1; ++a) {
        const double nd = it.first + adjw[a];
        if (nd < d[adj[a]]) {
          d[adj[a]] = nd;
          open.push({nd, adj[a]});
        }
      }
    }
  }
  // Cost of the cheapest start-goal path over slots whose verdict is KNOWN VALID, among the vertices of `in` (A* with
  // the exact distance-to-goal of the larger graph as its heuristic: consistent).  +inf if there is none.
  double valid_only_cost(const std::vector<uint8_t>& in, const std::vector<double>& to_goal) const {
    const size_t nv = rm->nv();
    std::vector<double> g(nv, INFINITY);
    std::priority_queue<Item, std::vector<Item>, std::greater<Item>> open;
    g[0] = 0.0;
    open.push({to_goal[0], 0u});
    while (!open.empty()) {
      const Item it = open.top();
      open.pop();
      const uint32_t u = it.second;
      if (it.first > g[u] + to_goal[u]) continue;
      if (u == 1u) return g[1];
      for (uint32_t a = rm->row[u], a1 = rm->row[u + 1]; a < a1; ++a) {
        const uint32_t v = rm->adj[a], e = rm->adj_edge[a];
        if (!in[v] || !std::isfinite(rm->adjw[a])) continue;
        if (rm->emotion[2 * (size_t)e + (u == rm->eu[e] ? 0u : 1u)] != 1) continue;
        const double ng = g[u] + rm->adjw[a];
        if (ng < g[v]) {
          g[v] = ng;
          open.push({ng + to_goal[v], v});
        }
      }
    }
    return INFINITY;
  }
  // From now on the tree spans only the vertices of `keep` (closed under tree ancestors here): the others can no
  // longer lie on a start-goal path, their distances are never needed again and their subtrees never re-settled.
  void restrict(std::vector<uint8_t>&& keep, const double* other_bound, double cmax) {
    const size_t nv = rm->nv();
    rel = std::move(keep);
    to_other = other_bound;
    cost_max = cmax;
    for (uint32_t v = 0; v < (uint32_t)nv; ++v)
      if (rel[v])
        for (uint32_t u = pred[v]; u != 0xffffffffu && !rel[u]; u = pred[u]) rel[u] = 1;
    first_child.assign(nv, 0xffffffffu);
    next_sib.assign(nv, 0xffffffffu);
    prev_sib.assign(nv, 0xffffffffu);
    for (uint32_t v = 0; v < (uint32_t)nv; ++v) {
      if (!rel[v]) {
        dist[v] = INFINITY;
        pred[v] = pred_edge[v] = 0xffffffffu;
      } else if (pred[v] != 0xffffffffu) {
        link(v);
      }
    }
  }
  // Equal-cost ways into a vertex: a best-first search from scratch (roadmap_astar with the zero heuristic of the
  // directional and the learned objective: (cost, id) heap, strict '<' on relaxation) keeps the parent it settles
  // FIRST, the one with the smaller (distance, id).  The tree follows the same rule everywhere, so that it stays the
  // tree that search would build -- with the directional objective equal sums are common (a chain of edges priced by
  // their yaw differences costs exactly the difference of its end yaws whichever way it goes).
  bool before(uint32_t u, uint32_t p) const { return dist[u] < dist[p] || (dist[u] == dist[p] && u < p); }
  void link(uint32_t v) {  // v becomes the first child of pred[v]
    const uint32_t p = pred[v], f = first_child[p];
    next_sib[v] = f;
    prev_sib[v] = 0xffffffffu;
    if (f != 0xffffffffu) prev_sib[f] = v;
    first_child[p] = v;
  }
  void unlink(uint32_t v) {
    const uint32_t p = pred[v], n = next_sib[v], q = prev_sib[v];
    if (q != 0xffffffffu) next_sib[q] = n; else first_child[p] = n;
    if (n != 0xffffffffu) prev_sib[n] = q;
    next_sib[v] = prev_sib[v] = 0xffffffffu;
  }
  // the tree edge into `child` is gone (its slots already dropped): re-settle child's subtree
  void repair(uint32_t child) {
    const uint32_t* adj = rm->adj.data();
    const uint32_t* adje = rm->adj_edge.data();
    const double* adjw = rm->adjw.data();
    sub.clear();
    if (pred[child] != 0xffffffffu) unlink(child);
    sub.push_back(child);
    in_s[child] = 1;
    for (size_t i = 0; i < sub.size(); ++i)  // the subtree, by the child lists
      for (uint32_t x = first_child[sub[i]]; x != 0xffffffffu; x = next_sib[x]) {
        in_s[x] = 1;
        sub.push_back(x);
      }
    for (const uint32_t x : sub) first_child[x] = next_sib[x] = prev_sib[x] = 0xffffffffu;  // all relinked below
    std::priority_queue<Item, std::vector<Item>, std::greater<Item>> open;
    for (const uint32_t x : sub) {
      dist[x] = INFINITY;
      pred[x] = pred_edge[x] = 0xffffffffu;
    }
    for (size_t i = 0; i < sub.size(); ++i) {  // the best way in from outside the subtree (inside: dist = +inf for now)
      const uint32_t x = sub[i];
      if (i + 2 < sub.size()) {  // the rows are ~850-byte chunks scattered over 8 MB: start the row after next now
        const uint32_t r = rm->row[sub[i + 2]];
        __builtin_prefetch(adj + r);
        __builtin_prefetch(adjw + r);
        __builtin_prefetch(adjw + r + 8);
        __builtin_prefetch(adjw + r + 16);
        __builtin_prefetch(adjw + r + 24);
      }
      double best = INFINITY;
      uint32_t ba = 0xffffffffu;
      for (uint32_t a = rm->row[x], a1 = rm->row[x + 1]; a < a1; ++a) {
        const double nd = dist[adj[a]] + adjw[a];
        if (nd < best || (nd == best && nd < INFINITY && before(adj[a], adj[ba]))) {
          best = nd;
          ba = a;
        }
      }
      // a vertex that has moved out of reach (distance + what is left to the goal > the cost bound) stays out: every
      // later graph only has fewer edges, and no vertex still in reach has its shortest path through it (the bound
      // to the goal is consistent)
      if (ba != 0xffffffffu && (!to_other || best + to_other[x] <= cost_max)) {
        dist[x] = best;
        pred[x] = adj[ba];
        pred_edge[x] = adje[ba];
        open.push({best, x});
      }
    }
    while (!open.empty()) {
      const Item it = open.top();
      open.pop();
      const uint32_t u = it.second;
      if (it.first > dist[u]) continue;
      const double du = dist[u];
      for (uint32_t a = rm->row[u], a1 = rm->row[u + 1]; a < a1; ++a) {
        const uint32_t v = adj[a];
        const double nd = du + adjw[a];
        // only subtree vertices can improve: the rest is final
        if (nd < dist[v]) {
          if (!to_other || (rel[v] && nd + to_other[v] <= cost_max)) {
            dist[v] = nd;
            pred[v] = u;
            pred_edge[v] = adje[a];
            open.push({nd, v});
          }
        } else if (nd == dist[v] && pred[v] != u && pred[v] != 0xffffffffu && before(u, pred[v])) {
          // an equally short way in through a parent that a search from scratch settles earlier: that search would
          // have kept it (see before()).  v may lie outside the subtree -- then its own subtree moves with it
          if (!in_s[v]) {
            unlink(v);
            pred[v] = u;
            pred_edge[v] = adje[a];
            link(v);
          } else {
            pred[v] = u;
            pred_edge[v] = adje[a];
          }
        }
      }
    }
    for (const uint32_t x : sub) {
      in_s[x] = 0;
      if (pred[x] != 0xffffffffu) link(x);
      else if (to_other) rel[x] = 0;  // out of reach for good (or cut off)
    }
  }
};

// checkMotion of (edge, direction) items in one device batch: vertex states and the two index lists go up, the
// endpoint states are gathered on the device (56 bytes per state never cross PCIe), one verdict byte per item comes back.
int roadmap_check_motion_items(artp_roadmap* rm, const std::vector<uint32_t>& src, const std::vector<uint32_t>& dst,
                               std::vector<uint8_t>* ok) {
  artp_ctx* c = rm->ctx;
  const size_t n = src.size(), nv = rm->nv();
  ok->assign(n, 0);
  if (n == 0) return ARTP_OK;
  if (n <= 64) {  // a path's worth: the host entry point (one call, no allocations)
    std::vector<double> s1(n * 7), s2(n * 7);
    for (size_t i = 0; i < n; ++i) {
      std::memcpy(&s1[i * 7], &rm->verts[(size_t)src[i] * 7], 7 * sizeof(double));
      std::memcpy(&s2[i * 7], &rm->verts[(size_t)dst[i] * 7], 7 * sizeof(double));
    }
    const int rc = artp_check_motions(c, s1.data(), s2.data(), n, ok->data());
    if (rc != ARTP_OK) return rc;
    return hipStreamSynchronize(c->stream) == hipSuccess ? ARTP_OK : ARTP_ERR_HIP;
  }
  // the context's host-entry staging slots (tmp[0], tmp[1]: the device entry point below does not touch them) instead of
  // four hipMalloc / hipFree pairs per call -- a hipFree alone waits for the device
  auto cleanup = []() {};
  RM_HIP(hipSetDevice(c->device));
  RM_TRY(ensure_tmp(c, 0, nv * 7 * sizeof(double) + 2 * n * sizeof(uint32_t) + 64));
  RM_TRY(ensure_tmp(c, 1, 2 * n * 7 * sizeof(double) + n + 64));
  double* d_v = static_cast<double*>(c->tmp[0]);
  uint32_t* d_uv = reinterpret_cast<uint32_t*>(d_v + nv * 7);
  double* d_s = static_cast<double*>(c->tmp[1]);
  uint8_t* d_ok = reinterpret_cast<uint8_t*>(d_s + 2 * n * 7);
  RM_HIP(hipMemcpyAsync(d_v, rm->verts.data(), nv * 7 * sizeof(double), hipMemcpyHostToDevice, c->stream));
  RM_HIP(hipMemcpyAsync(d_uv, src.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
  RM_HIP(hipMemcpyAsync(d_uv + n, dst.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(artp::gather_edge_states_uv_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream,
                     (const double*)d_v, (const uint32_t*)d_uv, (const uint32_t*)(d_uv + n), n, d_s, d_s + n * 7);
  RM_HIP(hipGetLastError());
  // in chunks of 2^18 motions (~13 M interior states): the pipeline's scratch (PoseRecs, queues) is sized by the largest
  // batch a context has seen, and a 10^6-motion batch would make it allocate tens of GB once
  for (size_t at = 0; at < n; at += (size_t)1 << 18) {
    const size_t m = std::min(n - at, (size_t)1 << 18);
    RM_TRY(artp_check_motions_dev(c, d_s + at * 7, d_s + (n + at) * 7, m, d_ok + at));
  }
  RM_HIP(hipMemcpyAsync(ok->data(), d_ok, n, hipMemcpyDeviceToHost, c->stream));
  RM_HIP(hipStreamSynchronize(c->stream));
  cleanup();
  return ARTP_OK;
}

#define ARTP_LAZY_PRECHECK_AFTER 3          // lazy removals before the whole graph is checked in one batch
#define ARTP_LAZY_PRECHECK_MAX (1u << 22)   // ... unless that is more motions than this

int roadmap_solve_tree(artp_roadmap* rm, double* path_se3, size_t cap_states, size_t* n_path, double* cost,
                       int* n_replans) {
  artp_ctx* c = rm->ctx;
  const size_t ne = rm->eu.size();
  const bool timing = variant_env("ARTP_SOLVE_TIMING") != nullptr;
  auto now = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_csr = now(), t_full = 0, t_check = 0, t_repair = 0, t_pre = 0;
  size_t n_check_calls = 0, n_checked = 0, sub_total = 0, n_pre = 0;
  if (rm->csr_dirty) roadmap_build_csr(rm);
  t_csr = now() - t_csr;
  const uint64_t mv = artp_map_version(c);
  if (rm->emotion_dirty || rm->emotion.size() != 2 * ne || rm->emotion_map_version != mv) {
    rm->emotion.assign(2 * ne, 0);
    rm->emotion_map_version = mv;
    rm->emotion_dirty = false;
  }
  LazyTree t;
  t.rm = rm;
  // the first search is goal-directed (A*): a roadmap whose first path passes -- the batched default's usual case --
  // never pays for the whole tree; the tree is built when the first edge has to go
  bool have_tree = false;
  int replans = 0;
  const char* env_inf = variant_env("ARTP_LAZY_INFORMED");
  bool informed = env_inf ? std::atoi(env_inf) != 0 : true;
  const double informed_beta[5] = {1.06, 1.25, 1.6, 2.5, INFINITY};
  int informed_round = 0, wrong_side = 0, since_switch = 0, n_switch = 0, pinned_root = -1;
  double bound = INFINITY, c_star = INFINITY, c_pre = 0, t_informed = 0;
  size_t n_rel = 0;
  std::vector<double> to_other;
  std::vector<uint8_t> within;
  struct SubStat { uint32_t n, bad, np; float ms; };
  std::vector<SubStat> sub_sizes;
  auto report = [&]() {
    if (timing && !sub_sizes.empty()) {
      std::vector<SubStat> ss = sub_sizes;
      std::sort(ss.begin(), ss.end(), [](const SubStat& a, const SubStat& b) { return a.n > b.n; });
      std::fprintf(stderr, "[solve] largest repairs (vertices @ position/path states, ms):");
      for (size_t i = 0; i < ss.size() && i < 12; ++i) std::fprintf(stderr, " %u@%u/%u %.3f", ss[i].n, ss[i].bad, ss[i].np, ss[i].ms);
      std::fprintf(stderr, "; median %u\n", ss[ss.size() / 2].n);
      if (variant_env("ARTP_SOLVE_SEQUENCE")) {
        std::fprintf(stderr, "[solve] sequence (position/states:vertices):");
        for (const SubStat& q : sub_sizes) std::fprintf(stderr, " %u/%u:%u", q.bad, q.np, q.n);
        std::fprintf(stderr, "\n");
      }
    }
    if (timing)
      std::fprintf(stderr, "[solve] nv %zu ne %zu: csr %.2f ms, full tree %.2f ms, %zu check calls (%zu motions) %.2f ms of which "
                   "precheck %zu motions %.2f ms, %d repairs %.2f ms (subtree vertices %zu); informed set %zu vertices "
                   "(C* %.4f, cost at the precheck %.4f, shell %d) %.2f ms; root %u after %d switches\n", rm->nv(), ne, t_csr, t_full,
                   n_check_calls, n_checked, t_check, n_pre, t_pre, replans, t_repair, sub_total, n_rel, c_star, c_pre,
                   informed_round, t_informed, t.root, n_switch);
  };
  bool prechecked = false, precheck_batch = false;
  std::vector<uint32_t> path, pedge, src, dst, item;
  std::vector<uint8_t> ok;
  for (;;) {
    double path_cost = INFINITY;
    pedge.clear();
    if (!have_tree) {
      if (!roadmap_astar(rm, &path, &path_cost)) {
        if (n_replans) *n_replans = replans;
        return ARTP_OK;  // *n_path == 0: start and goal are not connected
      }
      for (size_t i = 0; i + 1 < path.size(); ++i) {  // the (unique) edge between two path vertices
        uint32_t e = 0xffffffffu;
        for (uint32_t a = rm->row[path[i]]; a < rm->row[path[i] + 1]; ++a)
          if (rm->adj[a] == path[i + 1] && std::isfinite(rm->adjw[a])) e = rm->adj_edge[a];
        pedge.push_back(e);
      }
    } else {
      const uint32_t far = t.root == 0u ? 1u : 0u;
      if (!std::isfinite(t.dist[far]) && !t.rel.empty()) {
        // The tree was restricted to the vertices within C*, the cost of the cheapest path whose FORWARD verdicts were all
        // valid.  Verdicts are cached per direction while the reference keeps one undirected VALIDITY_TRUE bit per edge
        // (lazy_prm_star_min_update.cpp:707-726): should the two directions of an edge ever disagree (floating point
        // only), a later path can remove an edge of the C* path and the restricted tree loses the goal where the
        // reference would still solve.  Before concluding "not connected": the whole roadmap again.
        t.rel.clear();
        t.to_other = nullptr;
        t.cost_max = INFINITY;
        informed = false;
        t.full(t.root);
      }
      if (!std::isfinite(t.dist[far])) {
        if (n_replans) *n_replans = replans;
        report();
        return ARTP_OK;
      }
      path.clear();
      for (uint32_t v = far; v != t.root; v = t.pred[v]) {
        path.push_back(v);
        pedge.push_back(t.pred_edge[v]);
      }
      path.push_back(t.root);
      if (t.root == 0u) {  // read from the goal back to the start
        std::reverse(path.begin(), path.end());
        std::reverse(pedge.begin(), pedge.end());
      }  // either way pedge[i] = edge path[i] -> path[i + 1]
      path_cost = t.dist[far];
    }
    const size_t np = path.size();
    auto slot = [&](size_t i) { return 2 * (size_t)pedge[i] + (path[i] == rm->eu[pedge[i]] ? 0u : 1u); };
    // verdicts the table does not hold yet: this path's, or -- once the graph has shown that it needs its lazy checks
    // -- every usable edge's, both directions, in one batch
    src.clear();
    dst.clear();
    item.clear();
    if (!prechecked && replans >= ARTP_LAZY_PRECHECK_AFTER) {
      // Every later path costs at least this one's cost c and at most C*, the cost of the cheapest path whose motions
      // are all valid; a vertex v with d(start, v) + d(v, goal) > C* (distances in TODAY's graph: they only grow as
      // edges go) can lie on none of them.  C* needs verdicts, so: guess C* <= beta c, check the motions among the
      // vertices within beta c in one batch, find C* among them, widen beta if it was not there.
      const double t0 = now();
      if (informed && to_other.empty()) t.distances_from(t.root == 0u ? 1u : 0u, &to_other);
      t_informed += now() - t0;
      double beta = informed ? informed_beta[informed_round] : INFINITY;
      bound = std::isfinite(beta) ? beta * path_cost * (1.0 + 1e-9) : INFINITY;
      c_pre = path_cost;
      within.assign(rm->nv(), 1);
      if (std::isfinite(bound))
        for (size_t v = 0; v < rm->nv(); ++v) within[v] = t.dist[v] + to_other[v] <= bound;
      for (size_t e = 0; e < ne && item.size() <= ARTP_LAZY_PRECHECK_MAX; ++e) {
        if (!within[rm->eu[e]] || !within[rm->ev[e]] || !t.usable((uint32_t)e)) continue;
        for (unsigned d = 0; d < 2; ++d)
          if (rm->emotion[2 * e + d] == 0) {
            src.push_back(d ? rm->ev[e] : rm->eu[e]);
            dst.push_back(d ? rm->eu[e] : rm->ev[e]);
            item.push_back((uint32_t)(2 * e + d));
          }
      }
      if (item.size() <= ARTP_LAZY_PRECHECK_MAX) {
        precheck_batch = true;
      } else {  // too many motions for one batch: the per-path checks go on, nothing is restricted
        informed = false;
        src.clear();
        dst.clear();
        item.clear();
      }
      prechecked = true;
    }
    if (item.empty())
      for (size_t i = 0; i + 1 < np; ++i)
        if (rm->emotion[slot(i)] == 0) {
          src.push_back(path[i]);
          dst.push_back(path[i + 1]);
          item.push_back((uint32_t)slot(i));
        }
    if (!item.empty()) {
      const double t0 = now();
      const int rc = roadmap_check_motion_items(rm, src, dst, &ok);
      if (rc != ARTP_OK) return rc;
      for (size_t k = 0; k < item.size(); ++k) rm->emotion[item[k]] = ok[k] ? 1 : 2;
      t_check += now() - t0;
      ++n_check_calls;
      n_checked += item.size();
      if (precheck_batch) {
        t_pre += now() - t0;
        n_pre += item.size();
      }
    }
    if (precheck_batch) {
      precheck_batch = false;
      if (informed && t.rel.empty()) {
        const double t0 = now();
        const double cstar = t.valid_only_cost(within, t.root == 1u ? t.dist : to_other);  // heuristic: distance to the goal
        if (cstar <= bound) {  // the true C*: a cheaper all-valid path would lie within the bound as well
          std::vector<uint8_t> keep(rm->nv());
          size_t kept = 0;
          for (size_t v = 0; v < rm->nv(); ++v) kept += keep[v] = t.dist[v] + to_other[v] <= cstar * (1.0 + 1e-9);
          t.restrict(std::move(keep), to_other.data(), cstar * (1.0 + 1e-9));
          n_rel = kept;
          c_star = cstar;
        } else if (std::isfinite(bound)) {  // not within beta c: the next shell (the verdicts so far stay)
          ++informed_round;
          prechecked = false;
        }
        t_informed += now() - t0;
      }
    }
    size_t bad = np;
    for (size_t i = np - 1; i-- > 0;)  // the reference walks from the goal backwards
      if (rm->emotion[slot(i)] == 2) {
        bad = i;
        break;
      }
    if (bad == np) {
      if (path_se3) {
        if (cap_states < np) {
          c->last_error = "path buffer too small";
          *n_path = np;
          return ARTP_ERR_CAPACITY;
        }
        for (size_t i = 0; i < np; ++i)
          std::memcpy(path_se3 + i * 7, &rm->verts[(size_t)path[i] * 7], 7 * sizeof(double));
      }
      *n_path = np;
      *cost = path_cost;
      if (n_replans) *n_replans = replans;
      report();
      return ARTP_OK;
    }
    rm->eremoved[pedge[bad]] = 1;
    rm->d_graph_dirty = true;
    t.drop_edge(pedge[bad]);
    // where along the path the edge failed: 0 = at the start, 1 = at the goal
    const double where = np > 2 ? (double)bad / (double)(np - 2) : 0.5;
    if (!have_tree) {
      const double t0 = now();
      // the order among equal-cost paths is that of a search from the START (LazyTree::before): with the directional
      // objective, where equal sums are real, the tree stays there
      if (rm->params.objective == 1) pinned_root = 0;
      const char* env_root = variant_env("ARTP_LAZY_ROOT");  // tuning aid (variants build): 0 / 1 pins the root
      if (env_root) pinned_root = std::atoi(env_root) != 0 ? 1 : 0;
      const uint32_t new_root = pinned_root >= 0 ? (uint32_t)pinned_root : (where < 0.5 ? 1u : 0u);  // the far end
      // the distances to the OTHER end (what the informed set of the precheck is made of) on a second host thread while
      // this one builds the tree: both only read the adjacency, and the thread is joined before the next edge goes
      std::thread side;
      if (informed) {
        try {
          side = std::thread([&]() { t.distances_from(new_root == 0u ? 1u : 0u, &to_other); });
        } catch (const std::system_error&) {  // no thread to be had: the precheck computes the distances itself
        }
      }
      t.full(new_root);
      if (side.joinable()) side.join();
      t_full = now() - t0;
      have_tree = true;
    } else {
      const double t0 = now();
      t.repair(t.root == 0u ? path[bad + 1] : path[bad]);  // the tree edge into that vertex is gone
      t_repair += now() - t0;
      sub_total += t.sub.size();
      if (timing) sub_sizes.push_back({(uint32_t)t.sub.size(), (uint32_t)bad, (uint32_t)np, (float)(now() - t0)});
      // the edges keep failing on the root's side of the path: hang the tree from the other end (a search over the
      // vertices still in reach; the old root's exact distances become the bound towards it)
      const bool root_side = t.root == 0u ? where < 0.35 : where > 0.65;
      wrong_side = root_side ? wrong_side + 1 : 0;
      if (++since_switch >= 12 && wrong_side >= 6 && t.to_other && pinned_root < 0) {
        const double t1 = now();
        if (t.to_other) {
          to_other = t.dist;
          t.to_other = to_other.data();
        }
        t.full(t.root == 0u ? 1u : 0u);
        t_full += now() - t1;
        ++n_switch;
        wrong_side = since_switch = 0;
      }
    }
    if (++replans > (int)rm->params.max_replans) {
      c->last_error = "too many lazy edge removals";
      if (n_replans) *n_replans = replans;
      return ARTP_ERR_CAPACITY;
    }
  }
}

}  // namespace

int artp_roadmap_solve(artp_roadmap* rm, double* path_se3, size_t cap_states, size_t* n_path, double* cost,
                       int* n_replans) {
  if (!rm || !n_path || !cost) return ARTP_ERR_INVALID_ARG;
  artp_ctx* c = rm->ctx;
  *n_path = 0;
  *cost = INFINITY;
  // roadmaps the host searches: the shortest-path tree with deletion repair and cached motion verdicts
  if (rm->nv() < ARTP_SSSP_MIN_VERTICES && !variant_env("ARTP_SOLVE_ASTAR"))
    return roadmap_solve_tree(rm, path_se3, cap_states, n_path, cost, n_replans);
  int replans = 0;
  std::vector<uint32_t> path;
  std::vector<double> s1, s2;
  std::vector<uint8_t> ok;
  for (;;) {
    double cst = INFINITY;
    bool found;
    const bool on_device = rm->nv() >= ARTP_SSSP_MIN_VERTICES;
    if (on_device) {
      bool dev_ok = false;
      found = roadmap_sssp_dev(rm, &path, &cst, &dev_ok);
      if (!dev_ok) found = roadmap_astar(rm, &path, &cst);
    } else {
      found = roadmap_astar(rm, &path, &cst);
    }
    if (!found) {
      if (n_replans) *n_replans = replans;
      return ARTP_OK;  // *n_path == 0: start and goal are not connected (PlannerStatus::TIMEOUT)
    }
    // discrete motion check of the path's edges (constructSolution, prm_motion_cost.cpp:628-661); the
    // first invalid edge is removed and the search repeated
    const size_t np = path.size();
    s1.resize((np - 1) * 7);
    s2.resize((np - 1) * 7);
    ok.assign(np - 1, 0);
    for (size_t i = 0; i + 1 < np; ++i) {
      std::memcpy(&s1[i * 7], &rm->verts[(size_t)path[i] * 7], 7 * sizeof(double));
      std::memcpy(&s2[i * 7], &rm->verts[(size_t)path[i + 1] * 7], 7 * sizeof(double));
    }
    if (np > 1) {
      const int rc = artp_check_motions(c, s1.data(), s2.data(), np - 1, ok.data());
      if (rc != ARTP_OK) return rc;
      (void)hipStreamSynchronize(c->stream);
    }
    size_t bad = np;
    for (size_t i = np - 1; i-- > 0;)  // the reference walks from the goal backwards
      if (!ok[i]) {
        bad = i;
        break;
      }
    if (bad == np) {
      if (path_se3) {
        if (cap_states < np) {
          c->last_error = "path buffer too small";
          *n_path = np;
          return ARTP_ERR_CAPACITY;
        }
        for (size_t i = 0; i < np; ++i)
          std::memcpy(path_se3 + i * 7, &rm->verts[(size_t)path[i] * 7], 7 * sizeof(double));
      }
      *n_path = np;
      *cost = cst;
      if (n_replans) *n_replans = replans;
      return ARTP_OK;
    }
    // remove edge (path[bad], path[bad+1])
    const uint32_t a = std::min(path[bad], path[bad + 1]), b = std::max(path[bad], path[bad + 1]);
    if (!rm->csr_dirty) {
      for (uint32_t t = rm->row[a]; t < rm->row[a + 1]; ++t)
        if (rm->adj[t] == b) rm->eremoved[rm->adj_edge[t]] = 1;  // the search skips removed edges
    } else {
      // no CSR (device search): the edge list is sorted by (u, v), u < v
      size_t lo = 0, hi = rm->eu.size();
      while (lo < hi) {
        const size_t mid = (lo + hi) / 2;
        if (rm->eu[mid] < a || (rm->eu[mid] == a && rm->ev[mid] < b)) lo = mid + 1; else hi = mid;
      }
      if (lo < rm->eu.size() && rm->eu[lo] == a && rm->ev[lo] == b) rm->eremoved[lo] = 1;
    }
    rm->d_graph_dirty = true;
    if (++replans > (int)rm->params.max_replans) {
      c->last_error = "too many lazy edge removals";
      if (n_replans) *n_replans = replans;
      return ARTP_ERR_CAPACITY;
    }
  }
}


int artp_roadmap_set_density_map(artp_roadmap* rm, artp_preprocessed* pp, const artp_preprocess_params* prm) {
  if (!rm || (pp && !prm)) return ARTP_ERR_INVALID_ARG;
  rm->params.density_map = pp;
  rm->params.density_params = pp ? prm : nullptr;
  roadmap_fix_params(rm);
  return ARTP_OK;
}

int artp_roadmap_solve_until(artp_roadmap* rm, double plan_time, uint32_t grow_step, double* path_se3,
                             size_t cap_states, size_t* n_path, double* cost, uint64_t stats[3]) {
  if (!rm || !n_path || !cost || !(plan_time >= 0.0)) return ARTP_ERR_INVALID_ARG;
  const auto t0 = std::chrono::steady_clock::now();
  auto elapsed = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
  std::vector<double> best, cur(cap_states ? cap_states * 7 : 7);
  double best_cost = INFINITY;
  uint64_t rounds = 0, improved = 0;
  *n_path = 0;
  *cost = INFINITY;
  for (;;) {
    size_t n = 0;
    double cst = INFINITY;
    int replans = 0;
    int rc = artp_roadmap_solve(rm, nullptr, 0, &n, &cst, &replans);  // length first
    if (rc != ARTP_OK) return rc;
    if (n > 0 && cst < best_cost) {  // opt_->isCostBetterThan(c, bestCost_)
      cur.resize(n * 7);
      rc = artp_roadmap_solve(rm, cur.data(), n, &n, &cst, &replans);
      if (rc != ARTP_OK) return rc;
      best.assign(cur.begin(), cur.begin() + n * 7);
      best_cost = cst;
      ++improved;
    }
    if (elapsed() >= plan_time || grow_step == 0) break;  // ptc
    uint64_t g[2];
    rc = artp_roadmap_grow(rm, grow_step, g);  // `do sampleUniform while !isValid; addValidMilestone` in batches
    if (rc != ARTP_OK) return rc;
    ++rounds;
  }
  if (stats) {
    stats[0] = rounds;
    stats[1] = rm->nv();
    stats[2] = improved;
  }
  const size_t np = best.size() / 7;
  if (np == 0) return ARTP_OK;
  *n_path = np;
  *cost = best_cost;
  if (path_se3) {
    if (cap_states < np) {
      rm->ctx->last_error = "path buffer too small";
      return ARTP_ERR_CAPACITY;
    }
    std::memcpy(path_se3, best.data(), np * 7 * sizeof(double));
  }
  return ARTP_OK;
}

}  // extern "C"
