// preprocess.h -- "next" row N2: the per-map preprocessing chain on the device.
// The reference runs it on the CPU with OpenCV on every map update (Map::setMap -> processors::Basic,
// art_planner/src/map/processors/basic.cpp:42-143; estimateNormals, art_planner/src/utils.cpp:213-326;
// computeCumulativeProbabilityDistribution, processors/probability_distribution.cpp:20-46); its outputs are
// exactly the hot path's inputs: elevation_masked, the normals + plane_fit_std_dev, the sampling CDFs.
// All layers are grid_map matrices: float32, column-major rows x cols, element (i, j) at i + j * rows.
//
// Stencils, one lane per cell: estimateNormals (cross products along the axes and the diagonals, in the
// reference's accumulation order), grey erosion / dilation with a disk footprint, the select chain of
// setMaskedElevationAndTraversability / setTraversabilityFilter, and the row / column scans of the CDF.
// OpenCV is not available here: the disk is the set x^2 + y^2 <= (size/2)^2 on a size x size window with
// replicated borders (cv::circle's rasterisation and cv::erode's border value may differ in single border
// pixels) and inpainting is left to the caller -- parity for this row is unpinned; the tests compare with
// the numpy restatement that also generates the benchmark maps (art_planner_amd/synthetic.py).
#pragma once
#include "telea.h"

namespace artp {

struct PreGeom {
  int rows, cols;
  float res;
  double pos_x, pos_y, len_x, len_y;
};

// grid_map getPosition as stored in the reference's float position matrix (utils.cpp:238-247)
__device__ __forceinline__ float pre_cell_x(const PreGeom& g, int i) {
  return (float)((g.pos_x + (0.5 * g.len_x - 0.5 * (double)g.res)) - (double)g.res * (double)i);
}
__device__ __forceinline__ float pre_cell_y(const PreGeom& g, int j) {
  return (float)((g.pos_y + (0.5 * g.len_y - 0.5 * (double)g.res)) - (double)g.res * (double)j);
}

struct NormalAcc {
  float sx, sy, sz, max_dz;
  int n;
};

// vec_sum += (a - c).cross(b - c).normalized(); max_z_diff; ++n_vec  (utils.cpp:262-266)
__device__ __forceinline__ void normal_accumulate(NormalAcc& acc, float cx, float cy, float cz, float ax, float ay,
                                                  float az, float bx, float by, float bz) {
  const float ux = ax - cx, uy = ay - cy, uz = az - cz;
  const float vx = bx - cx, vy = by - cy, vz = bz - cz;
  float tx = uy * vz - uz * vy, ty = uz * vx - ux * vz, tz = ux * vy - uy * vx;
  const float nrm = sqrtf(tx * tx + ty * ty + tz * tz);
  if (nrm > 0.f) {
    tx /= nrm;
    ty /= nrm;
    tz /= nrm;
  }
  acc.sx += tx;
  acc.sy += ty;
  acc.sz += tz;
  acc.max_dz = fmaxf(acc.max_dz, fmaxf(fabsf(uz), fabsf(vz)));
  ++acc.n;
}

__global__ void __launch_bounds__(256)
estimate_normals_kernel(const float* __restrict__ elev, PreGeom g, int n_r, int n_d, float* __restrict__ nx,
                        float* __restrict__ ny, float* __restrict__ nz, float* __restrict__ stdv) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= g.rows * g.cols) return;
  const int i = t % g.rows, j = t / g.rows;
  auto Z = [&](int a, int b) { return elev[a + (size_t)b * g.rows]; };
  const float cx = pre_cell_x(g, i), cy = pre_cell_y(g, j), cz = Z(i, j);
  NormalAcc acc{0.f, 0.f, 0.f, 0.f, 0};
  for (int o = 1; o < n_r; ++o) {  // +x / +y neighbours, both must exist
    if (i + o >= g.rows || j + o >= g.cols) continue;
    normal_accumulate(acc, cx, cy, cz, pre_cell_x(g, i + o), cy, Z(i + o, j), cx, pre_cell_y(g, j + o), Z(i, j + o));
  }
  for (int o = 1; o < n_r; ++o) {  // -x / -y
    if (i - o < 0 || j - o < 0) continue;
    normal_accumulate(acc, cx, cy, cz, pre_cell_x(g, i - o), cy, Z(i - o, j), cx, pre_cell_y(g, j - o), Z(i, j - o));
  }
  for (int o = 1; o < n_d; ++o) {  // diagonals (+,+) and (-,+)
    if (i + o >= g.rows || j + o >= g.cols || i - o < 0) continue;
    normal_accumulate(acc, cx, cy, cz, pre_cell_x(g, i + o), pre_cell_y(g, j + o), Z(i + o, j + o),
                      pre_cell_x(g, i - o), pre_cell_y(g, j + o), Z(i - o, j + o));
  }
  for (int o = 1; o < n_d; ++o) {  // diagonals (-,-) and (+,-)
    if (i - o < 0 || j - o < 0 || i + o >= g.rows) continue;
    normal_accumulate(acc, cx, cy, cz, pre_cell_x(g, i - o), pre_cell_y(g, j - o), Z(i - o, j - o),
                      pre_cell_x(g, i + o), pre_cell_y(g, j - o), Z(i + o, j - o));
  }
  float sx = acc.sx, sy = acc.sy, sz = acc.sz;
  if (acc.n > 0) {
    sx /= (float)acc.n;
    sy /= (float)acc.n;
    sz /= (float)acc.n;
  }
  const float nrm = sqrtf(sx * sx + sy * sy + sz * sz);
  if (nrm > 0.f) {
    sx /= nrm;
    sy /= nrm;
    sz /= nrm;
  }
  nx[t] = sx;
  ny[t] = sy;
  nz[t] = sz;
  stdv[t] = acc.max_dz;
}

// grey erosion (min) / dilation (max) with the footprint of getCircularKernel(size) (utils.cpp:114-119), anchored at
// (size/2, size/2) like cv::erode / cv::dilate; replicated borders (for these footprints the same as OpenCV's
// ignored border: clamping an offset moves it towards the anchor, where the footprint is at least as wide).
// fp: one 64-bit row mask per footprint row (bit x of row y = kernel(y, x)), `fsize` rows (<= 64).
template <bool DILATE>
__global__ void __launch_bounds__(256)
morph_kernel(const float* __restrict__ in, int rows, int cols, int fsize, const unsigned long long* __restrict__ fp,
             float* __restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= rows * cols) return;
  const int i = t % rows, j = t / rows;
  const int r = fsize / 2;
  float v = DILATE ? -INFINITY : INFINITY;
  for (int y = 0; y < fsize; ++y) {
    const unsigned long long m = fp[y];
    const int jj = min(max(j + y - r, 0), cols - 1);
    for (int x = 0; x < fsize; ++x) {
      if (!((m >> x) & 1ull)) continue;
      const int ii = min(max(i + x - r, 0), rows - 1);
      const float xv = in[ii + (size_t)jj * rows];
      v = DILATE ? fmaxf(v, xv) : fminf(v, xv);
    }
  }
  out[t] = v;
}

// the select chain of setMaskedElevationAndTraversability (basic.cpp:57-106), one elementwise stage each
__global__ void __launch_bounds__(256)
pre_threshold_kernel(const float* __restrict__ trav, int n, float thres, float* __restrict__ trav_filter) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) trav_filter[t] = (trav ? trav[t] : 1.0f) > thres ? 1.0f : 0.0f;
}
// safety = hole_mask ? trav_filter : closed;  then  wall_mask ? 1 : safety      (basic.cpp:75-88)
__global__ void __launch_bounds__(256)
pre_masks_kernel(const float* __restrict__ elev, const float* __restrict__ elev_eroded,
                 const float* __restrict__ elev_dilated, const float* __restrict__ trav_filter,
                 const float* __restrict__ closed, int n, float max_drop, float min_step, float* __restrict__ safety,
                 float* __restrict__ wall_mask) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const bool hole = (elev[t] - elev_eroded[t]) > max_drop;
  const bool wall = (elev_dilated[t] - elev[t]) > min_step;
  float s = hole ? trav_filter[t] : closed[t];
  s = wall ? 1.0f : s;
  safety[t] = s;
  wall_mask[t] = wall ? 1.0f : 0.0f;
}
// (trav_filter < 0.5 || wall) ? trav_filter : eroded      (basic.cpp:91-93)
__global__ void __launch_bounds__(256)
pre_keep_unsafe_kernel(const float* __restrict__ trav_filter, const float* __restrict__ wall_mask,
                       const float* __restrict__ eroded, int n, float* __restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) out[t] = (trav_filter[t] < 0.5f || (wall_mask && wall_mask[t] > 0.5f)) ? trav_filter[t] : eroded[t];
}
// elevation_masked = safety > 0.5 ? elevation : -inf      (basic.cpp:101-105)
__global__ void __launch_bounds__(256)
pre_masked_elevation_kernel(const float* __restrict__ elev, const float* __restrict__ safety, int n,
                            float* __restrict__ masked) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) masked[t] = safety[t] > 0.5f ? elev[t] : -INFINITY;
}

// CDF (probability_distribution.cpp:20-46): one lane per row walks its columns in order
__global__ void __launch_bounds__(64)
cdf_rows_kernel(const float* __restrict__ prob, int rows, int cols, float* __restrict__ cum_prob,
                float* __restrict__ row_sum) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows) return;
  // the additions stay in column order (the reference's float rounding); only the LOADS are taken eight at a time, in
  // front of the dependent chain -- a load per addition made every step wait for memory (0.15 ms for 400 columns)
  constexpr int U = 8;
  float s = 0.f;
  for (int j0 = 0; j0 < cols; j0 += U) {
    float v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = (j0 + u < cols) ? prob[i + (size_t)(j0 + u) * rows] : 0.f;
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (j0 + u < cols) s += v[u];
  }
  row_sum[i] = s;
  float c = 0.f;
  for (int j0 = 0; j0 < cols; j0 += U) {
    float v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = (j0 + u < cols) ? prob[i + (size_t)(j0 + u) * rows] : 0.f;
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (j0 + u < cols) {
        c += v[u] / s;
        cum_prob[i + (size_t)(j0 + u) * rows] = c;
      }
  }
}
// one workgroup: the row sums go through LDS, thread 0 adds them in row order
#define ARTP_CDF_ROWS_LDS 4096
__global__ void __launch_bounds__(256)
cdf_rowwise_kernel(const float* __restrict__ row_sum, int rows, float* __restrict__ cum_rowwise,
                   float* __restrict__ any_prob) {
  __shared__ float rs[ARTP_CDF_ROWS_LDS];
  if (blockIdx.x != 0) return;
  const bool in_lds = rows <= ARTP_CDF_ROWS_LDS;
  if (in_lds) {
    for (int i = threadIdx.x; i < rows; i += blockDim.x) rs[i] = row_sum[i];
    __syncthreads();
  }
  if (!in_lds) {  // maps with more rows than the LDS buffer holds: the plain serial form
    if (threadIdx.x != 0) return;
    float total = 0.f;
    for (int i = 0; i < rows; ++i) total += row_sum[i];
    float c = 0.f;
    for (int i = 0; i < rows; ++i) {
      c += row_sum[i] / total;
      cum_rowwise[i] = c;
    }
    *any_prob = total;
    return;
  }
  // the ORDERED parts (total, running sum) stay with thread 0, reads taken eight at a time in front of the additions;
  // the quotients, which do not depend on each other, are formed by everybody in between
  __shared__ float s_total;
  constexpr int U = 8;
  if (threadIdx.x == 0) {
    float total = 0.f;
    for (int i0 = 0; i0 < rows; i0 += U) {
      float v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = (i0 + u < rows) ? rs[i0 + u] : 0.f;
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (i0 + u < rows) total += v[u];
    }
    s_total = total;
    *any_prob = total;
  }
  __syncthreads();
  const float total = s_total;
  for (int i = threadIdx.x; i < rows; i += blockDim.x) rs[i] = rs[i] / total;
  __syncthreads();
  if (threadIdx.x == 0) {
    float c = 0.f;
    for (int i0 = 0; i0 < rows; i0 += U) {
      float v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = (i0 + u < rows) ? rs[i0 + u] : 0.f;
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (i0 + u < rows) {
          c += v[u];
          cum_rowwise[i0 + u] = c;
        }
    }
  }
}
__global__ void __launch_bounds__(256)
pre_fill_kernel(float* __restrict__ out, int n, float v) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) out[t] = v;
}

// unknown_space_untraversable: traversability = observed > 0.5 ? traversability : 0   (basic.cpp:49-54)
__global__ void __launch_bounds__(256)
pre_unknown_untraversable_kernel(const float* __restrict__ observed, int n, float* __restrict__ trav) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n && !(observed[t] > 0.5f)) trav[t] = 0.0f;
}

// computeInverseSampleDensity (sample_density.cpp:12-43): roadmap vertices per cell ...
__global__ void __launch_bounds__(256)
vertex_histogram_kernel(const double* __restrict__ se3, size_t nv, PreGeom g, float* __restrict__ counts) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nv) return;
  const double px = se3[7 * t + 0], py = se3[7 * t + 1];
  // grid_map getIndex: inside test, then index = (int)(-((p - L/2 - c) / res))
  const double tx = -((px - g.pos_x) - 0.5 * g.len_x), ty = -((py - g.pos_y) - 0.5 * g.len_y);
  if (!(tx >= 0.0 && ty >= 0.0 && tx < g.len_x && ty < g.len_y)) return;
  int i = (int)(tx / (double)g.res), j = (int)(ty / (double)g.res);
  i = min(i, g.rows - 1);
  j = min(j, g.cols - 1);
  atomicAdd(&counts[i + (size_t)j * g.rows], 1.0f);
}
// ... blurred with cv::GaussianBlur's separable kernel (k taps, BORDER_REFLECT_101), one pass per axis
template <bool ALONG_ROWS>
__global__ void __launch_bounds__(256)
gauss_pass_kernel(const float* __restrict__ in, int rows, int cols, const float* __restrict__ taps, int k,
                  float* __restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= rows * cols) return;
  const int i = t % rows, j = t / rows;
  const int r = k / 2, len = ALONG_ROWS ? rows : cols, at = ALONG_ROWS ? i : j;
  float acc = 0.f;
  for (int d = -r; d <= r; ++d) {
    int p = at + d;
    if (len == 1) p = 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * (len - 1) - p;  // gfedcb|abcdefgh|gfedcba
    acc += taps[d + r] * (ALONG_ROWS ? in[p + (size_t)j * rows] : in[i + (size_t)p * rows]);
  }
  out[t] = acc;
}
// max over a non-negative layer (float bits are monotone there)
__global__ void __launch_bounds__(256)
nonneg_max_kernel(const float* __restrict__ in, int n, unsigned* __restrict__ out_bits) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned v = (t < n) ? __float_as_uint(fmaxf(in[t], 0.0f)) : 0u;
  for (int off = 32; off > 0; off >>= 1) v = max(v, (unsigned)__shfl_xor((int)v, off, 64));
  if ((threadIdx.x & 63) == 0 && v) atomicMax(out_bits, v);
}
// sample_probability = max - blurred (if the blurred layer is not all zero), then * sample filter
// (applyBaseSampleDistribution, probability_distribution.cpp:9-16)
__global__ void __launch_bounds__(256)
base_distribution_kernel(const float* __restrict__ blurred, const unsigned* __restrict__ max_bits,
                         const float* __restrict__ filter, int n, float* __restrict__ prob) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  float p = 1.0f;
  if (blurred && *max_bits) p = __uint_as_float(*max_bits) - blurred[t];
  prob[t] = p * filter[t];
}
// applyMaxUnknownProbability (probability_distribution.cpp:50-90): probability mass of the observed and of
// the unobserved cells ...
// Summed in a FIXED order -- every workgroup its grid-stride share (thread order, then a shuffle tree, then the four
// wavefronts in order) into partial[2 b], partial[2 b + 1]; a second launch adds the partial sums in workgroup order:
// floating-point atomics from 2 500 wavefronts added the same numbers in a different order every run, and the cap's
// scale factor -- through it the CDF -- could differ in the last bit between two runs on the same map.
__global__ void __launch_bounds__(256)
known_unknown_mass_kernel(const float* __restrict__ prob, const float* __restrict__ observed, int n,
                          double* __restrict__ partial) {
  __shared__ double part[2][4];
  double known = 0.0, unknown = 0.0;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
    if (observed[t] > 0.0f) known += (double)prob[t];
    else unknown += (double)prob[t];
  }
  for (int off = 32; off > 0; off >>= 1) {
    known += __shfl_xor(known, off, 64);
    unknown += __shfl_xor(unknown, off, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    part[0][threadIdx.x >> 6] = known;
    part[1][threadIdx.x >> 6] = unknown;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[2 * blockIdx.x] = ((part[0][0] + part[0][1]) + part[0][2]) + part[0][3];
    partial[2 * blockIdx.x + 1] = ((part[1][0] + part[1][1]) + part[1][2]) + part[1][3];
  }
}
__global__ void known_unknown_mass_final_kernel(const double* __restrict__ partial, int n_part, double* __restrict__ mass) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  double k = 0.0, u = 0.0;
  for (int b = 0; b < n_part; ++b) {
    k += partial[2 * b];
    u += partial[2 * b + 1];
  }
  mass[0] = k;
  mass[1] = u;
}
// ... and the rescaling that caps the unobserved share at max_prob
__global__ void __launch_bounds__(256)
cap_unknown_kernel(const float* __restrict__ observed, const double* __restrict__ mass, double max_prob, int n,
                   float* __restrict__ prob) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const double known = mass[0], unknown = mass[1];
  const double base_unknown = unknown / (known + unknown);
  if (known > 0 && unknown > 0 && base_unknown > max_prob) {
    const float mult = (float)(observed[t] > 0.0f ? (1 - max_prob) / known : max_prob / unknown);
    prob[t] = prob[t] * mult;
  }
}
// computeChange (change.cpp:9-51) for two maps of equal size whose origins differ by (si, sj) cells:
// updated = 1 unless the cell exists in both, its height moved by <= thres and it did not turn untraversable
__global__ void __launch_bounds__(256)
change_kernel(const float* __restrict__ elev_new, const float* __restrict__ trav_new,
              const float* __restrict__ elev_old, const float* __restrict__ trav_old, int rows, int cols, int si,
              int sj, float thres, float* __restrict__ updated, int* __restrict__ rect, unsigned long long* __restrict__ count) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const bool in = t < rows * cols;
  const int i = in ? t % rows : 0, j = in ? t / rows : 0;
  const int io = i - si, jo = j - sj;
  float u = in ? 1.0f : 0.0f;
  if (in && io >= 0 && jo >= 0 && io < rows && jo < cols) {
    const size_t o = (size_t)io + (size_t)jo * rows;
    const bool height_changed = fabsf(elev_new[t] - elev_old[o]) > thres;
    const bool trav_changed = trav_old[o] - trav_new[t] > 0.5f;
    if (!height_changed && !trav_changed) u = 0.0f;
  }
  if (in) updated[t] = u;
  // bounding rectangle and count of the changed cells: reduced across the wavefront first (an atomic per changed cell
  // on five words: 8 000 cells of a 5 % update queued up behind each other)
  const bool ch = u != 0.0f;
  const unsigned long long bal = __ballot(ch);
  if (bal == 0ull) return;
  int i0 = ch ? i : 0x7fffffff, j0 = ch ? j : 0x7fffffff, i1 = ch ? i : -1, j1 = ch ? j : -1;
  for (int off = 32; off > 0; off >>= 1) {
    i0 = min(i0, __shfl_xor(i0, off, 64));
    j0 = min(j0, __shfl_xor(j0, off, 64));
    i1 = max(i1, __shfl_xor(i1, off, 64));
    j1 = max(j1, __shfl_xor(j1, off, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMin(&rect[0], i0);
    atomicMin(&rect[1], j0);
    atomicMax(&rect[2], i1);
    atomicMax(&rect[3], j1);
    atomicAdd(count, (unsigned long long)__popcll(bal));
  }
}


// ---- hole filling: inpaintMatrix (art_planner/src/utils.cpp:13-64) / the cost node's _elvMapProcess
// (art_planner_motion_cost/scripts/cost_query_server.py:92-111) ------------------------------------------------
// Both quantise the WHOLE layer to 8 bit over [min, max] of its valid cells, inpaint the holes on the 8-bit image
// (cv::inpaint, INPAINT_TELEA, radius 3) and scale back, so every cell -- hole or not -- comes out quantised.
// mode 0 (planner): cv::Mat::convertTo(CV_8U, 255/(max-min), -min*255/(max-min)) = saturate(cvRound(x*a + b)) in
//   float; back: q * ((max-min)/255) + min; then column 0 := column 1, row 0 := row 1 (utils.cpp:60-61).
// mode 1 (cost node): ((x - min) * 255 / (max - min)).astype(uint8) -- truncation; back: q * (max-min) / 255 + min.
// The quantisation and the scaling are restated exactly; the fill itself is NOT Telea's fast-marching method
// (OpenCV is not available here: parity of the hole cells is UNPINNED): holes are closed from their rim inwards,
// each pass giving every hole cell that has filled neighbours within radius 3 their 1/d^2-weighted mean.
__global__ void __launch_bounds__(256)
inpaint_quantise_kernel(const float* __restrict__ in, int n, int mode, float lo, float hi, unsigned char* __restrict__ q,
                        unsigned char* __restrict__ known, unsigned long long* __restrict__ n_holes) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned hole = 0;
  if (i < n) {
    const float x = in[i];
    const bool ok = mode == 0 ? !(x != x) : ((__float_as_uint(x) & 0x7f800000u) != 0x7f800000u);
    // planner mask = NaN cells only (cv::patchNaNs workaround, utils.cpp:27-32); the node masks every non-finite cell
    float v;
    if (mode == 0) {
      const float a = 255.0f / (hi - lo), b = -lo * 255.0f / (hi - lo);
      v = rintf(x * a + b);  // cvRound: round half to even
      if (!(v >= -2147483648.0f && v < 2147483648.0f)) v = 0.0f;  // cvtss2si of inf / NaN = INT_MIN -> saturates to 0
    } else {
      v = truncf((x - lo) * 255.0f / (hi - lo));
    }
    v = v < 0.0f ? 0.0f : (v > 255.0f ? 255.0f : v);  // saturate_cast<uchar>; NaN -> 0 (masked anyway)
    q[i] = ok ? (unsigned char)v : 0;
    known[i] = ok ? 1 : 0;
    hole = ok ? 0u : 1u;
  }
  const unsigned long long bal = __ballot(hole != 0);
  if ((threadIdx.x & 63) == 0 && bal) atomicAdd(n_holes, (unsigned long long)__popcll(bal));
}

// one pass: hole cells with filled cells within radius 3 get their weighted mean; reads (q, known), writes (q2, known2)
__global__ void __launch_bounds__(256)
inpaint_fill_pass_kernel(const unsigned char* __restrict__ q, const unsigned char* __restrict__ known, int rows, int cols,
                         unsigned char* __restrict__ q2, unsigned char* __restrict__ known2,
                         unsigned* __restrict__ n_left) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned left = 0;
  if (i < rows * cols) {
    unsigned char v = q[i], k = known[i];
    if (!k) {
      const int r = i % rows, c = i / rows;  // column-major rows x cols
      float sw = 0.0f, sv = 0.0f;
      for (int dc = -3; dc <= 3; ++dc)
        for (int dr = -3; dr <= 3; ++dr) {
          const int rr = r + dr, cc = c + dc, d2 = dr * dr + dc * dc;
          if (d2 == 0 || d2 > 9 || rr < 0 || cc < 0 || rr >= rows || cc >= cols) continue;
          const int j = rr + cc * rows;
          if (known[j]) {
            const float w = 1.0f / (float)d2;
            sw += w;
            sv += w * (float)q[j];
          }
        }
      if (sw > 0.0f) {
        v = (unsigned char)rintf(sv / sw);
        k = 1;
      } else {
        left = 1;
      }
    }
    q2[i] = v;
    known2[i] = k;
  }
  if (__any(left != 0) && (threadIdx.x & 63) == 0) atomicAdd(n_left, 1u);
}

__global__ void __launch_bounds__(256)
inpaint_dequantise_kernel(const unsigned char* __restrict__ q, int rows, int cols, int mode, float lo, float hi,
                          float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * cols) return;
  int r = i % rows, c = i / rows;
  if (mode == 0) {  // col(0) = col(1), then row(0) = row(1): cell (r, c) shows the value of (max(r,1), max(c,1))
    if (c == 0 && cols > 1) c = 1;
    if (r == 0 && rows > 1) r = 1;
  }
  const float v = (float)q[r + c * rows];
  out[i] = mode == 0 ? v * ((hi - lo) / 255.0f) + lo : v * (hi - lo) / 255.0f + lo;
}
}  // namespace artp

// -------------------------------------------------------------------------------------------------------
namespace {

enum PreLayer {
  PRE_ELEV = 0, PRE_TRAV, PRE_NX, PRE_NY, PRE_NZ, PRE_STD, PRE_TRAV_FILTER, PRE_SAFETY, PRE_MASKED, PRE_SAMPLE_PROB,
  PRE_CUM_PROB, PRE_OBSERVED, PRE_NSAMPLES, PRE_SAMPLE_FILTER, PRE_UPDATED, PRE_T0, PRE_T1, PRE_T2, PRE_T3, PRE_COUNT
};

const char* const kPreLayerNames[] = {"elevation", "traversability", "normal_x", "normal_y", "normal_z",
                                      "plane_fit_std_dev", "traversability_thresholded_no_safety",
                                      "traversability_thresholded", "elevation_masked", "sample_probability",
                                      "cum_prob", "observed", "n_samples", "traversability_sample_filter",
                                      "updated"};

}  // namespace

struct artp_preprocessed {
  int rows = 0, cols = 0;
  double len_x = 0, len_y = 0, pos_x = 0, pos_y = 0;
  float* buf = nullptr;          // PRE_COUNT layers + cum_prob_rowwise (rows) + 1 scalar
  float* layer(int k) const { return buf + (size_t)k * rows * cols; }
  float* rowwise() const { return buf + (size_t)PRE_COUNT * rows * cols; }
  // scalars behind the row CDF: [0] total probability, [1] max of the blurred density (bits), [2..5] two doubles
  float* scalars() const { return buf + ((((size_t)PRE_COUNT * rows * cols) + rows + 1) & ~(size_t)1); }
};


namespace {
// getCircularKernel(size) (utils.cpp:114-119): a size x size image, cv::circle(centre (size/2, size/2), radius size/2,
// filled).  The fill is the midpoint circle of OpenCV's drawing.cpp (Circle(): error term err / plus / minus, for
// every step the spans of rows centre -+ dy over [centre - dx, centre + dx] and of rows centre -+ dx over
// [centre - dy, centre + dy]), clipped to the image -- restated from the published OpenCV source, which is not
// installed here (parity UNPINNED).  size <= 0: cv::Mat() -> cv::erode / cv::dilate use a 3 x 3 rectangle.
std::vector<unsigned long long> circular_footprint(int size, int* fsize) {
  if (size <= 0) {
    *fsize = 3;
    return {7ull, 7ull, 7ull};
  }
  if (size > 64) size = 64;
  *fsize = size;
  std::vector<unsigned long long> rows(size, 0ull);
  const int radius = size / 2, cx = radius, cy = radius;
  auto hline = [&](int y, int x0, int x1) {
    if (y < 0 || y >= size) return;
    x0 = x0 < 0 ? 0 : x0;
    x1 = x1 >= size ? size - 1 : x1;
    for (int x = x0; x <= x1; ++x) rows[y] |= 1ull << x;
  };
  int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
  while (dx >= dy) {
    hline(cy - dy, cx - dx, cx + dx);
    hline(cy + dy, cx - dx, cx + dx);
    hline(cy - dx, cx - dy, cx + dy);
    hline(cy + dx, cx - dy, cx + dy);
    ++dy;
    err += plus;
    plus += 2;
    const int mask = (err <= 0) - 1;
    err -= minus & mask;
    dx += mask;
    minus -= mask & 2;
  }
  return rows;
}

// The sampling distribution of a preprocessed map (planner.cpp:43-56): [inverse vertex density] * sample filter
// [capped unknown share], then the CDF -- the part of the processor chain that depends on the roadmap's vertices
// and that Map::reApplyPreprocessing() (map.cpp:94-96) re-runs while PRMMotionCostMaintainer::sampleGraph grows
// the roadmap (prm_motion_cost.cpp:190-193).  d_verts: n_vertices x 7 doubles in HBM (may be null).  Asynchronous.
bool pre_sampling_distribution(artp_ctx* c, artp_preprocessed* pp, const artp_preprocess_params* prm,
                               const double* d_verts, size_t n_vertices) {
  const int rows = pp->rows, cols = pp->cols, n = rows * cols;
  const double res = pp->len_x / rows;
  hipStream_t st = c->stream;
  const dim3 grid((unsigned)((n + 255) / 256)), blk(256);
  auto L = [&](int k) { return pp->layer(k); };
  const artp::PreGeom geom{rows, cols, (float)res, pp->pos_x, pp->pos_y, pp->len_x, pp->len_y};
  float* d_taps = nullptr;
  unsigned* max_bits = reinterpret_cast<unsigned*>(pp->scalars() + 1);
  double* mass = reinterpret_cast<double*>(pp->scalars() + 2);  // 8-byte aligned (even offset, see scalars())
  bool ok = hipMemsetAsync(pp->scalars(), 0, 8 * sizeof(float), st) == hipSuccess;
  const bool density = prm->use_inverse_vertex_density && n_vertices > 0 && d_verts;
  if (density) {
    const double blur_radius = (c->params.torso_length + c->params.torso_width) * 0.25;  // planner.cpp:48
    int k = (int)(6 * blur_radius / res);
    const double sigma = blur_radius / res;
    if (k % 2 == 0) k += 1;
    // cv::getGaussianKernel: t_i = exp(-(i - (k-1)/2)^2 / (2 sigma^2)) as float, normalised by their sum
    std::vector<float> taps(k);
    double sum = 0.0;
    for (int i = 0; i < k; ++i) {
      const double x = i - (k - 1) * 0.5;
      taps[i] = (float)std::exp(-0.5 / (sigma * sigma) * x * x);
      sum += taps[i];
    }
    for (int i = 0; i < k; ++i) taps[i] = (float)(taps[i] * (1.0 / sum));
    ok = ok && hipMalloc(reinterpret_cast<void**>(&d_taps), k * sizeof(float)) == hipSuccess &&
         hipMemcpyAsync(d_taps, taps.data(), k * sizeof(float), hipMemcpyHostToDevice, st) == hipSuccess &&
         hipMemsetAsync(L(PRE_T0), 0, (size_t)n * 4, st) == hipSuccess;
    if (ok) {
      hipLaunchKernelGGL(artp::vertex_histogram_kernel, dim3((unsigned)((n_vertices + 255) / 256)), blk, 0, st,
                         d_verts, n_vertices, geom, L(PRE_T0));
      hipLaunchKernelGGL(artp::gauss_pass_kernel<true>, grid, blk, 0, st, (const float*)L(PRE_T0), rows, cols,
                         (const float*)d_taps, k, L(PRE_T1));
      hipLaunchKernelGGL(artp::gauss_pass_kernel<false>, grid, blk, 0, st, (const float*)L(PRE_T1), rows, cols,
                         (const float*)d_taps, k, L(PRE_NSAMPLES));
      hipLaunchKernelGGL(artp::nonneg_max_kernel, grid, blk, 0, st, (const float*)L(PRE_NSAMPLES), n, max_bits);
    }
  } else {
    ok = ok && hipMemsetAsync(L(PRE_NSAMPLES), 0, (size_t)n * 4, st) == hipSuccess;
  }
  hipLaunchKernelGGL(artp::base_distribution_kernel, grid, blk, 0, st,
                     density ? (const float*)L(PRE_NSAMPLES) : (const float*)nullptr, (const unsigned*)max_bits,
                     (const float*)L(PRE_SAMPLE_FILTER), n, L(PRE_SAMPLE_PROB));
  if (prm->use_max_prob_unknown_samples) {
    // the partial sums live in the scratch layer the row sums use further down (8-byte aligned inside it); a map too
    // small to hold them is summed by one workgroup straight into `mass`
    const int n_part = n >= 8192 ? 64 : 1;
    double* partial = n_part > 1 ? reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(L(PRE_T0)) + 7) & ~(uintptr_t)7) : mass;
    hipLaunchKernelGGL(artp::known_unknown_mass_kernel, dim3(n_part), dim3(256), 0, st, (const float*)L(PRE_SAMPLE_PROB),
                       (const float*)L(PRE_OBSERVED), n, partial);
    if (n_part > 1)
      hipLaunchKernelGGL(artp::known_unknown_mass_final_kernel, dim3(1), dim3(64), 0, st, (const double*)partial, n_part, mass);
    hipLaunchKernelGGL(artp::cap_unknown_kernel, grid, blk, 0, st, (const float*)L(PRE_OBSERVED), (const double*)mass,
                       prm->max_prob_unknown_samples, n, L(PRE_SAMPLE_PROB));
  }
  // computeCumulativeProbabilityDistribution                       probability_distribution.cpp:20-46
  float* row_sum = L(PRE_T0);
  float* total = pp->scalars();
  hipLaunchKernelGGL(artp::cdf_rows_kernel, dim3((unsigned)((rows + 63) / 64)), dim3(64), 0, st,
                     (const float*)L(PRE_SAMPLE_PROB), rows, cols, L(PRE_CUM_PROB), row_sum);
  hipLaunchKernelGGL(artp::cdf_rowwise_kernel, dim3(1), dim3(256), 0, st, (const float*)row_sum, rows, pp->rowwise(),
                     total);
  ok = ok && hipGetLastError() == hipSuccess;
  if (d_taps) {
    ok = hipStreamSynchronize(st) == hipSuccess && ok;  // the taps are read by the kernels above
    (void)hipFree(d_taps);
  }
  return ok;
}
}  // namespace

extern "C" {

void artp_preprocess_params_defaults(artp_preprocess_params* p) {
  if (!p) return;
  std::memset(p, 0, sizeof(*p));
  p->traversability_thres = 0.5f;  // Params::planner.traversability_thres (params.h:24)
  // Params::planner.safety defaults are all zero (params.h:27-34): no morphology
  p->use_inverse_vertex_density = 0;      // params.h:82-84
  p->use_max_prob_unknown_samples = 0;
  p->max_prob_unknown_samples = 0.1;
}

void artp_preprocess_params_yaml(artp_preprocess_params* p) {  // art_planner_ros/config/params.yaml
  artp_preprocess_params_defaults(p);
  if (!p) return;
  p->traversability_thres = 0.15f;
  p->foothold_margin = 0.3;
  p->foothold_margin_max_hole_size = 0.3;
  p->foothold_margin_max_drop = 0.3;
  p->foothold_margin_max_drop_search_radius = 0.16;
  p->foothold_margin_min_step = 0.3;
  p->foothold_size = 0.1;
  p->use_inverse_vertex_density = 1;      // params.yaml:49-51
  p->use_max_prob_unknown_samples = 1;
  p->max_prob_unknown_samples = 0.1;
}

void artp_preprocessed_destroy(artp_preprocessed* pp) {
  if (!pp) return;
  if (pp->buf) (void)hipFree(pp->buf);
  delete pp;
}

int artp_preprocess_map(artp_ctx* c, const float* elevation, const float* traversability, int rows, int cols,
                        double len_x, double len_y, double pos_x, double pos_y, const artp_preprocess_params* prm,
                        artp_preprocessed** out) {
  artp_preprocess_inputs in;
  std::memset(&in, 0, sizeof(in));
  in.elevation = elevation;
  in.traversability = traversability;
  in.rows = rows;
  in.cols = cols;
  in.len_x = len_x;
  in.len_y = len_y;
  in.pos_x = pos_x;
  in.pos_y = pos_y;
  return artp_preprocess_map_ex(c, &in, prm, out);
}

int artp_preprocess_map_ex(artp_ctx* c, const artp_preprocess_inputs* in, const artp_preprocess_params* prm,
                           artp_preprocessed** out) {
  if (!c || !in || !in->elevation || !prm || !out || in->rows < 2 || in->cols < 2 ||
      (in->n_vertices && !in->vertex_se3))
    return ARTP_ERR_INVALID_ARG;
  const float* elevation = in->elevation;
  const float* traversability = in->traversability;
  const int rows = in->rows, cols = in->cols;
  const double len_x = in->len_x, len_y = in->len_y, pos_x = in->pos_x, pos_y = in->pos_y;
  *out = nullptr;
  std::unique_lock<std::recursive_mutex> lock(c->mu);
  HIP_TRY(c, hipSetDevice(c->device));
  const int n = rows * cols;
  auto pp = new artp_preprocessed();
  pp->rows = rows;
  pp->cols = cols;
  pp->len_x = len_x;
  pp->len_y = len_y;
  pp->pos_x = pos_x;
  pp->pos_y = pos_y;
  if (hipMalloc(reinterpret_cast<void**>(&pp->buf), ((size_t)PRE_COUNT * n + rows + 2 + 16) * sizeof(float)) != hipSuccess) {
    delete pp;
    c->last_error = "hipMalloc failed in artp_preprocess_map";
    return ARTP_ERR_HIP;
  }
  hipStream_t st = c->stream;
  const double res = len_x / rows;
  const dim3 grid((unsigned)((n + 255) / 256)), blk(256);
  auto L = [&](int k) { return pp->layer(k); };
  bool ok = hipMemcpyAsync(L(PRE_ELEV), elevation, (size_t)n * 4, hipMemcpyHostToDevice, st) == hipSuccess;
  if (traversability)
    ok = ok && hipMemcpyAsync(L(PRE_TRAV), traversability, (size_t)n * 4, hipMemcpyHostToDevice, st) == hipSuccess;
  else
    hipLaunchKernelGGL(artp::pre_fill_kernel, grid, blk, 0, st, L(PRE_TRAV), n, 1.0f);  // checkTraversability

  // addKnownCells (basic.cpp:26-38): "observed" = the cells that were valid before inpainting
  if (in->observed)
    ok = ok && hipMemcpyAsync(L(PRE_OBSERVED), in->observed, (size_t)n * 4, hipMemcpyHostToDevice, st) == hipSuccess;
  else
    hipLaunchKernelGGL(artp::pre_fill_kernel, grid, blk, 0, st, L(PRE_OBSERVED), n, 1.0f);
  if (c->params.unknown_space_untraversable)
    hipLaunchKernelGGL(artp::pre_unknown_untraversable_kernel, grid, blk, 0, st, (const float*)L(PRE_OBSERVED), n,
                       L(PRE_TRAV));

  // estimateNormals(map, (torso.length + torso.width) * 0.25)      basic.cpp:47
  {
    const double radius = (c->params.torso_length + c->params.torso_width) * 0.25;
    artp::PreGeom g{rows, cols, (float)res, pos_x, pos_y, len_x, len_y};
    hipLaunchKernelGGL(artp::estimate_normals_kernel, grid, blk, 0, st, (const float*)L(PRE_ELEV), g,
                       (int)(radius / res), (int)(radius * 0.70710678118 / res), L(PRE_NX), L(PRE_NY), L(PRE_NZ),
                       L(PRE_STD));
  }
  // setMaskedElevationAndTraversability                            basic.cpp:57-106
  // footprints of this call's morphology sizes, uploaded once each (64 rows of 64 bits at most)
  unsigned long long* d_fp = nullptr;
  ok = ok && hipMalloc(reinterpret_cast<void**>(&d_fp), 16 * 64 * sizeof(unsigned long long)) == hipSuccess;
  int fp_sizes[16], fp_fsize[16], n_fp = 0;
  auto footprint = [&](int size, int* fsize) -> const unsigned long long* {
    for (int q = 0; q < n_fp; ++q)
      if (fp_sizes[q] == size) {
        *fsize = fp_fsize[q];
        return d_fp + 64 * q;
      }
    const std::vector<unsigned long long> rowsv = circular_footprint(size, fsize);
    const int q = n_fp < 16 ? n_fp++ : 15;
    fp_sizes[q] = size;
    fp_fsize[q] = *fsize;
    ok = ok && d_fp && hipMemcpyAsync(d_fp + 64 * q, rowsv.data(), rowsv.size() * 8, hipMemcpyHostToDevice, st) == hipSuccess &&
         hipStreamSynchronize(st) == hipSuccess;  // rowsv dies with this call
    return d_fp + 64 * q;
  };
  auto erode = [&](const float* in, int size, float* o) {
    int fs = 0;
    const unsigned long long* fp = footprint(size, &fs);
    if (ok) hipLaunchKernelGGL(artp::morph_kernel<false>, grid, blk, 0, st, in, rows, cols, fs, fp, o);
  };
  auto dilate = [&](const float* in, int size, float* o) {
    int fs = 0;
    const unsigned long long* fp = footprint(size, &fs);
    if (ok) hipLaunchKernelGGL(artp::morph_kernel<true>, grid, blk, 0, st, in, rows, cols, fs, fp, o);
  };
  const int fh = (int)std::ceil(prm->foothold_size / res);
  const int margin = (int)std::ceil(2 * prm->foothold_margin / res);
  const int hole = (int)std::floor(prm->foothold_margin_max_hole_size / res);
  const int search = (int)std::ceil(2 * prm->foothold_margin_max_drop_search_radius / res);
  hipLaunchKernelGGL(artp::pre_threshold_kernel, grid, blk, 0, st, (const float*)L(PRE_TRAV), n,
                     prm->traversability_thres, L(PRE_TRAV_FILTER));
  dilate(L(PRE_TRAV_FILTER), hole, L(PRE_T0));
  erode(L(PRE_T0), hole, L(PRE_T1));                 // T1 = closed holes
  erode(L(PRE_ELEV), search, L(PRE_T0));             // T0 = eroded elevation
  dilate(L(PRE_ELEV), margin, L(PRE_T2));            // T2 = dilated elevation
  hipLaunchKernelGGL(artp::pre_masks_kernel, grid, blk, 0, st, (const float*)L(PRE_ELEV), (const float*)L(PRE_T0),
                     (const float*)L(PRE_T2), (const float*)L(PRE_TRAV_FILTER), (const float*)L(PRE_T1), n,
                     (float)prm->foothold_margin_max_drop, (float)prm->foothold_margin_min_step, L(PRE_SAFETY),
                     L(PRE_T3));                     // T3 = wall mask
  erode(L(PRE_SAFETY), margin, L(PRE_T0));
  hipLaunchKernelGGL(artp::pre_keep_unsafe_kernel, grid, blk, 0, st, (const float*)L(PRE_TRAV_FILTER),
                     (const float*)L(PRE_T3), (const float*)L(PRE_T0), n, L(PRE_T1));
  erode(L(PRE_T1), fh, L(PRE_T0));
  dilate(L(PRE_T0), fh, L(PRE_T1));
  hipLaunchKernelGGL(artp::pre_keep_unsafe_kernel, grid, blk, 0, st, (const float*)L(PRE_TRAV_FILTER),
                     (const float*)nullptr, (const float*)L(PRE_T1), n, L(PRE_SAFETY));
  hipLaunchKernelGGL(artp::pre_masked_elevation_kernel, grid, blk, 0, st, (const float*)L(PRE_ELEV),
                     (const float*)L(PRE_SAFETY), n, L(PRE_MASKED));
  // setTraversabilityFilter                                        basic.cpp:110-125
  {
    const double total_reach = std::sqrt(c->params.reach_x * c->params.reach_x + c->params.reach_y * c->params.reach_y);
    const double min_wall = std::min((c->params.torso_length - c->params.reach_x) * 0.5,
                                     (c->params.torso_width - c->params.reach_y) * 0.5);
    dilate(L(PRE_SAFETY), (int)(total_reach / res), L(PRE_T0));
    erode(L(PRE_T0), (int)(total_reach / res), L(PRE_T1));
    erode(L(PRE_T1), (int)(min_wall / res), L(PRE_SAMPLE_FILTER));
  }
  // the sampling distribution (planner.cpp:43-56): [inverse vertex density] * sample filter [capped unknown share]
  double* d_verts = nullptr;
  if (prm->use_inverse_vertex_density && in->n_vertices > 0)
    ok = ok && hipMalloc(reinterpret_cast<void**>(&d_verts), in->n_vertices * 7 * sizeof(double)) == hipSuccess &&
         hipMemcpyAsync(d_verts, in->vertex_se3, in->n_vertices * 7 * sizeof(double), hipMemcpyHostToDevice, st) ==
             hipSuccess;
  ok = ok && pre_sampling_distribution(c, pp, prm, d_verts, d_verts ? in->n_vertices : 0);
  ok = ok && hipGetLastError() == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
  if (d_verts) (void)hipFree(d_verts);
  if (d_fp) (void)hipFree(d_fp);
  if (!ok) {
    c->last_error = "device preprocessing failed";
    lock.unlock();
    artp_preprocessed_destroy(pp);
    return ARTP_ERR_HIP;
  }
  *out = pp;
  return ARTP_OK;
}

int artp_preprocessed_get_layer(artp_ctx* c, const artp_preprocessed* pp, const char* name, float* out) {
  if (!c || !pp || !name || !out) return ARTP_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  HIP_TRY(c, hipSetDevice(c->device));
  if (std::strcmp(name, "cum_prob_rowwise") == 0) {
    HIP_TRY(c, hipMemcpy(out, pp->rowwise(), (size_t)pp->rows * 4, hipMemcpyDeviceToHost));
    return ARTP_OK;
  }
  for (int k = 0; k < (int)(sizeof(kPreLayerNames) / sizeof(kPreLayerNames[0])); ++k)
    if (std::strcmp(name, kPreLayerNames[k]) == 0) {
      HIP_TRY(c, hipMemcpy(out, pp->layer(k), (size_t)pp->rows * pp->cols * 4, hipMemcpyDeviceToHost));
      return ARTP_OK;
    }
  c->last_error = std::string("unknown layer ") + name;
  return ARTP_ERR_INVALID_ARG;
}

// computeChange (change.cpp:9-51, the "updated" layer of LazyPRMStarMinUpdate's maintenance) between two
// preprocessed maps of the same size and resolution; rect = {row0, col0, nrows, ncols} of the updated cells
// in the NEW map (nrows = 0 when nothing changed) -- what artp_update_layer_rect / a roadmap re-check need.
int artp_preprocessed_change(artp_ctx* c, artp_preprocessed* map_new, const artp_preprocessed* map_old,
                             float height_change_for_update, float* updated_out, int rect[4], uint64_t* n_updated) {
  if (!c || !map_new || !map_old || map_new->rows != map_old->rows || map_new->cols != map_old->cols)
    return ARTP_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  HIP_TRY(c, hipSetDevice(c->device));
  const int rows = map_new->rows, cols = map_new->cols, n = rows * cols;
  const double res = map_new->len_x / rows;
  // cell (i, j) of the new map lies over cell (i - si, j - sj) of the old one
  const int si = (int)std::lround((map_new->pos_x - map_old->pos_x) / res);
  const int sj = (int)std::lround((map_new->pos_y - map_old->pos_y) / res);
  int* d_rect = reinterpret_cast<int*>(map_new->scalars() + 8);
  unsigned long long* d_cnt = reinterpret_cast<unsigned long long*>(map_new->scalars() + 12);
  const int init[4] = {0x7fffffff, 0x7fffffff, -1, -1};
  HIP_TRY(c, hipMemcpyAsync(d_rect, init, sizeof(init), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemsetAsync(d_cnt, 0, 8, c->stream));
  hipLaunchKernelGGL(artp::change_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream,
                     (const float*)map_new->layer(PRE_ELEV), (const float*)map_new->layer(PRE_SAFETY),
                     (const float*)map_old->layer(PRE_ELEV), (const float*)map_old->layer(PRE_SAFETY), rows, cols, si, sj,
                     height_change_for_update, map_new->layer(PRE_UPDATED), d_rect, d_cnt);
  HIP_TRY(c, hipGetLastError());
  int r[4];
  unsigned long long cnt = 0;
  HIP_TRY(c, hipMemcpyAsync(r, d_rect, sizeof(r), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipMemcpyAsync(&cnt, d_cnt, 8, hipMemcpyDeviceToHost, c->stream));
  if (updated_out)
    HIP_TRY(c, hipMemcpyAsync(updated_out, map_new->layer(PRE_UPDATED), (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (rect) {
    rect[0] = cnt ? r[0] : 0;
    rect[1] = cnt ? r[1] : 0;
    rect[2] = cnt ? r[2] - r[0] + 1 : 0;
    rect[3] = cnt ? r[3] - r[1] + 1 : 0;
  }
  if (n_updated) *n_updated = cnt;
  return ARTP_OK;
}

// Planner::setMap (planner.cpp:135-163): make the preprocessed layers the context's current map -- both
// height fields (with their range / partner tables), the sampler layers and the z bounds.
int artp_preprocessed_install(artp_ctx* c, const artp_preprocessed* pp) {
  if (!c || !pp) return ARTP_ERR_INVALID_ARG;
  const size_t n = (size_t)pp->rows * pp->cols;
  float total = 0.f;
  {
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpy(&total, pp->scalars(), 4, hipMemcpyDeviceToHost));
  }
  if (!(total > 0.f)) {
    c->last_error = "sample_probability is zero everywhere: nothing can be sampled on this map";
    return ARTP_ERR_NO_MAP;
  }
  // everything stays in HBM: layout flips, copies and the z-bound reduction run on the device
  int rc = upload_layer_from_device(c, ARTP_SLOT_BODY, pp->layer(PRE_ELEV), pp->rows, pp->cols, pp->len_x, pp->len_y,
                                    pp->pos_x, pp->pos_y);
  if (rc) return rc;
  rc = upload_layer_from_device(c, ARTP_SLOT_FEET, pp->layer(PRE_MASKED), pp->rows, pp->cols, pp->len_x, pp->len_y,
                                pp->pos_x, pp->pos_y);
  if (rc) return rc;
  rc = upload_sampler_layers_from_device(c, pp->layer(PRE_CUM_PROB), pp->rowwise(), pp->layer(PRE_ELEV),
                                         pp->layer(PRE_NX), pp->layer(PRE_NY), pp->layer(PRE_NZ), pp->layer(PRE_STD),
                                         pp->rows, pp->cols, pp->len_x, pp->len_y, pp->pos_x, pp->pos_y);
  if (rc) return rc;
  // bounds.low[2] / high[2] = min / max finite elevation -/+ reach.z / 2 (planner.cpp:146-156)
  float lo = 0.f, hi = 0.f;
  bool any = false;
  rc = finite_min_max_dev(c, pp->layer(PRE_ELEV), n, &lo, &hi, &any);
  if (rc) return rc;
  if (!any) lo = hi = 0.f;
  return artp_set_z_bounds(c, (double)lo - c->params.reach_z / 2, (double)hi + c->params.reach_z / 2);
}


// Map::reApplyPreprocessing (map.cpp:94-96) for the part of the chain that can change on an unchanged map: the
// sampling distribution, re-weighted by the inverse density of the given roadmap vertices (device pointer, n x 7
// doubles; null / 0 = no density term), then the CDF.  With install_sampler the context's sampler switches to the
// new CDF at once (the height fields are untouched).
int artp_preprocessed_reweight_dev(artp_ctx* c, artp_preprocessed* pp, const artp_preprocess_params* prm,
                                   const double* vertex_se3_dev, size_t n_vertices, int install_sampler) {
  if (!c || !pp || !prm || (n_vertices && !vertex_se3_dev)) return ARTP_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  HIP_TRY(c, hipSetDevice(c->device));
  if (!pre_sampling_distribution(c, pp, prm, vertex_se3_dev, n_vertices)) {
    c->last_error = "re-weighting the sampling distribution failed";
    return ARTP_ERR_HIP;
  }
  if (!install_sampler) return ARTP_OK;
  float total = 0.f;
  HIP_TRY(c, hipMemcpyAsync(&total, pp->scalars(), 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (!(total > 0.f)) {
    c->last_error = "sample_probability is zero everywhere: nothing can be sampled on this map";
    return ARTP_ERR_NO_MAP;
  }
  return upload_sampler_layers_from_device(c, pp->layer(PRE_CUM_PROB), pp->rowwise(), pp->layer(PRE_ELEV), pp->layer(PRE_NX),
                                           pp->layer(PRE_NY), pp->layer(PRE_NZ), pp->layer(PRE_STD), pp->rows, pp->cols,
                                           pp->len_x, pp->len_y, pp->pos_x, pp->pos_y);
}


// d_in -> d_out (both rows x cols column-major floats in HBM; may alias).  *n_holes = masked cells.  No holes: a
// plain copy (both reference functions are only called when the layer has invalid cells).
static int inpaint_dev(artp_ctx* c, const float* d_in, int rows, int cols, int mode, float* d_out, uint64_t* n_holes) {
  const size_t n = (size_t)rows * cols;
  float lo = 0.f, hi = 0.f;
  bool any = false;
  int rc = finite_min_max_dev(c, d_in, n, &lo, &hi, &any);
  if (rc) return rc;
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  HIP_TRY(c, hipSetDevice(c->device));
  hipStream_t st = c->stream;
  unsigned char* buf = nullptr;
  HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&buf), 4 * n + 64));
  unsigned char *q = buf, *known = buf + n, *q2 = buf + 2 * n, *known2 = buf + 3 * n;
  unsigned long long* d_holes = reinterpret_cast<unsigned long long*>(c->d_count);
  unsigned* d_left = reinterpret_cast<unsigned*>(buf + 4 * n + (8 - (4 * n) % 8) % 8);
  const dim3 grid((unsigned)((n + 255) / 256)), blk(256);
  auto fail = [&](int code) {
    (void)hipFree(buf);
    return code;
  };
  if (hipMemsetAsync(d_holes, 0, 8, st) != hipSuccess) return fail(ARTP_ERR_HIP);
  if (!any || !(hi > lo)) {
    mode &= 1;
    // no valid cell at all, or a constant layer (the reference divides by max - min = 0 here): holes take the
    // constant, nothing is quantised
    hipLaunchKernelGGL(artp::inpaint_quantise_kernel, grid, blk, 0, st, d_in, (int)n, mode, 0.0f, 1.0f, q, known, d_holes);
    unsigned long long h = 0;
    if (hipMemcpyAsync(&h, d_holes, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
      return fail(ARTP_ERR_HIP);
    if (n_holes) *n_holes = h;
    std::vector<float> tmp(n);
    if (hipMemcpy(tmp.data(), d_in, n * 4, hipMemcpyDeviceToHost) != hipSuccess) return fail(ARTP_ERR_HIP);
    for (float& v : tmp)
      if (!std::isfinite(v)) v = any ? lo : 0.0f;
    if (hipMemcpy(d_out, tmp.data(), n * 4, hipMemcpyHostToDevice) != hipSuccess) return fail(ARTP_ERR_HIP);
    return fail(ARTP_OK);
  }
  // Telea's march reads a 3 x 3 neighbourhood around every band pixel with cv::inpaint's border index rule: it needs at
  // least two rows and two columns.  A one-row / one-column layer takes the device fill instead.
  const bool telea = (mode & ARTP_INPAINT_TELEA) != 0 && rows >= 2 && cols >= 2;
  mode &= 1;
  hipLaunchKernelGGL(artp::inpaint_quantise_kernel, grid, blk, 0, st, d_in, (int)n, mode, lo, hi, q, known, d_holes);
  unsigned long long h = 0;
  if (hipMemcpyAsync(&h, d_holes, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
    return fail(ARTP_ERR_HIP);
  if (n_holes) *n_holes = h;
  if (h == 0) {
    if (d_out != d_in && hipMemcpyAsync(d_out, d_in, n * 4, hipMemcpyDeviceToDevice, st) != hipSuccess)
      return fail(ARTP_ERR_HIP);
    if (hipStreamSynchronize(st) != hipSuccess) return fail(ARTP_ERR_HIP);
    return fail(ARTP_OK);
  }
  if (telea) {
    // Telea's fast-marching fill (telea.h) on the host, on the image as the reference hands it to cv::inpaint: the planner
    // wraps the column-major layer as a cv::Mat(cols, rows) (utils.cpp:16-19: the buffer read row-major), the cost node
    // works on a[r][c] = layer(rows - 1 - r, cols - 1 - c) (cost_query_server.py:74)
    std::vector<unsigned char> hq(n), hk(n), img(n), msk(n);
    if (hipMemcpyAsync(hq.data(), q, n, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipMemcpyAsync(hk.data(), known, n, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
      return fail(ARTP_ERR_HIP);
    auto src_of = [&](size_t i) {   // image index -> layer index
      if (mode == 0) return i;
      const size_t r = i / (size_t)cols, cc = i % (size_t)cols;
      return ((size_t)rows - 1 - r) + ((size_t)cols - 1 - cc) * (size_t)rows;
    };
    for (size_t i = 0; i < n; ++i) {
      img[i] = hq[src_of(i)];
      msk[i] = hk[src_of(i)] ? 0 : 1;
    }
    if (mode == 0) artp_telea::inpaint_u8(cols, rows, img.data(), msk.data(), 3);
    else artp_telea::inpaint_u8(rows, cols, img.data(), msk.data(), 3);
    for (size_t i = 0; i < n; ++i) hq[src_of(i)] = img[i];
    if (hipMemcpyAsync(q, hq.data(), n, hipMemcpyHostToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
      return fail(ARTP_ERR_HIP);
  }
  for (int group = 0; !telea && group < (rows + cols) / 3 + 2; ++group) {
    if (hipMemsetAsync(d_left, 0, 4, st) != hipSuccess) return fail(ARTP_ERR_HIP);
    for (int p = 0; p < 4; ++p) {  // an even number of passes: the result is back in (q, known)
      hipLaunchKernelGGL(artp::inpaint_fill_pass_kernel, grid, blk, 0, st, (const unsigned char*)q,
                         (const unsigned char*)known, rows, cols, q2, known2, d_left);
      hipLaunchKernelGGL(artp::inpaint_fill_pass_kernel, grid, blk, 0, st, (const unsigned char*)q2,
                         (const unsigned char*)known2, rows, cols, q, known, d_left);
    }
    unsigned left = 0;
    if (hipMemcpyAsync(&left, d_left, 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
      return fail(ARTP_ERR_HIP);
    if (!left) break;
  }
  hipLaunchKernelGGL(artp::inpaint_dequantise_kernel, grid, blk, 0, st, (const unsigned char*)q, rows, cols, mode, lo, hi,
                     d_out);
  if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return fail(ARTP_ERR_HIP);
  return fail(ARTP_OK);
}

int artp_inpaint_layer(artp_ctx* c, const float* layer, int rows, int cols, int mode, float* out, uint64_t* n_holes) {
  if (!c || !layer || !out || rows < 1 || cols < 1 || mode < 0 || mode > 3) return ARTP_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  HIP_TRY(c, hipSetDevice(c->device));
  const size_t n = (size_t)rows * cols;
  float* d = nullptr;
  HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&d), n * sizeof(float)));
  int rc = hipMemcpy(d, layer, n * 4, hipMemcpyHostToDevice) == hipSuccess ? ARTP_OK : ARTP_ERR_HIP;
  if (rc == ARTP_OK) rc = inpaint_dev(c, d, rows, cols, mode, d, n_holes);
  if (rc == ARTP_OK && hipMemcpy(out, d, n * 4, hipMemcpyDeviceToHost) != hipSuccess) rc = ARTP_ERR_HIP;
  (void)hipFree(d);
  return rc;
}

int artp_telea_inpaint_u8(const uint8_t* img, const uint8_t* mask, int h, int w, int range, uint8_t* out) {
  if (!img || !mask || !out || h < 2 || w < 2 || range < 1 || range > 16) return ARTP_ERR_INVALID_ARG;
  if (out != img) std::memcpy(out, img, (size_t)h * w);
  artp_telea::inpaint_u8(h, w, out, mask, range);
  return ARTP_OK;
}

int artp_cost_set_hole_filling(artp_ctx* c, int enabled) {
  if (!c) return ARTP_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  c->cost_fill_holes = enabled != 0;      // 2: with Telea's fill (ARTP_INPAINT_TELEA)
  c->cost_fill_telea = enabled == 2;
  return ARTP_OK;
}

// artp_cost_update_map_layer for a layer with holes when hole filling is on: the node's _elvMapProcess on the device
int cost_update_map_layer_filled(artp_ctx* c, const float* layer, int rows, int cols, double res, double len_x,
                                 double len_y, double pos_x, double pos_y) {
  std::vector<float> filled((size_t)rows * cols);
  uint64_t holes = 0;
  const int rc = artp_inpaint_layer(c, layer, rows, cols, ARTP_INPAINT_COST_NODE | (c->cost_fill_telea ? ARTP_INPAINT_TELEA : 0),
                                    filled.data(), &holes);
  if (rc != ARTP_OK) return rc;
  c->cost_fill_holes = false;  // the filled layer has no holes; avoid recursion
  const int rc2 = artp_cost_update_map_layer(c, filled.data(), rows, cols, res, len_x, len_y, pos_x, pos_y);
  c->cost_fill_holes = true;
  return rc2;
}

}  // extern "C"
