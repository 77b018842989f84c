// telea.h -- Telea's fast-marching inpainting (A. Telea, "An image inpainting technique based on the fast marching method",
// J. Graphics Tools 9(1), 2004) with the conventions of OpenCV's cv::inpaint(..., cv::INPAINT_TELEA) for single-channel
// 8-bit images: the fill behind art_planner's inpaintMatrix (art_planner/src/utils.cpp:44-48, radius 3) and the cost
// node's _elvMapProcess (art_planner_motion_cost/scripts/cost_query_server.py:107).  Host code (a sequential march over
// a priority queue; a 400 x 400 layer with 10 % holes takes ~10 ms), selected by ARTP_INPAINT_TELEA in artp_inpaint_layer.
//
// OpenCV is not installed in this image and its sources are not under /root/reference: this is a restatement of the
// published algorithm with OpenCV's (opencv/modules/photo/src/inpaint.cpp, 3.x / 4.x) documented particulars --
//   * flags KNOWN / BAND / INSIDE on an image with a one-pixel border, T = 1e6 inside, 0 on the band (known pixels
//     4-adjacent to the mask),
//   * T of the known pixels within the (2 r + 1)^2 box dilation of the mask from an outward march, negated,
//   * the march's four-neighbour order (i-1, j), (i, j-1), (i+1, j), (i, j+1), first-in-first-out among equal T,
//   * weights dst = 1 / |r|^3, lev = 1 / (1 + |T(k,l) - T(i,j)|), dir = r . gradT (|dir| <= 0.01 -> 1e-6),
//   * OpenCV's image-gradient quirks (one-sided differences unscaled, central ones times 2) and its normalised
//     first-order term (Jx + Jy) / (sqrt(Jx^2 + Jy^2) + 1e-20), + 0.5, saturate_cast<uchar>
// -- written from the paper and from memory of that file; UNPINNED (nothing here can run OpenCV to compare).
#pragma once

#include <cmath>
#include <cstdint>
#include <queue>
#include <vector>

namespace artp_telea {

enum : uint8_t { KNOWN = 0, BAND = 1, INSIDE = 2, CHANGE = 3 };

struct HeapItem {
  float T;
  uint64_t seq;
  int i, j;
};
struct HeapCmp {
  bool operator()(const HeapItem& a, const HeapItem& b) const { return a.T != b.T ? a.T > b.T : a.seq > b.seq; }
};
using Heap = std::priority_queue<HeapItem, std::vector<HeapItem>, HeapCmp>;

struct Grid {   // (H + 2) x (W + 2) with a one-pixel border, row-major
  int er, ec;
  std::vector<uint8_t> f;
  std::vector<float> t;
  uint8_t& F(int i, int j) { return f[(size_t)i * ec + j]; }
  float& T(int i, int j) { return t[(size_t)i * ec + j]; }
};

inline float fm_solve(Grid& g, int i1, int j1, int i2, int j2) {
  const double a11 = g.T(i1, j1), a22 = g.T(i2, j2), m12 = a11 < a22 ? a11 : a22;
  double sol;
  if (g.F(i1, j1) != INSIDE) {
    if (g.F(i2, j2) != INSIDE) {
      if (std::fabs(a11 - a22) >= 1.0) sol = 1 + m12;
      else sol = (a11 + a22 + std::sqrt(2 - (a11 - a22) * (a11 - a22))) * 0.5;
    } else {
      sol = 1 + a11;
    }
  } else if (g.F(i2, j2) != INSIDE) {
    sol = 1 + a22;
  } else {
    sol = 1 + m12;
  }
  return (float)sol;
}
inline float fm_min4(Grid& g, int i, int j) {
  const float a = fm_solve(g, i - 1, j, i, j - 1), b = fm_solve(g, i + 1, j, i, j - 1), c = fm_solve(g, i - 1, j, i, j + 1),
              d = fm_solve(g, i + 1, j, i, j + 1);
  const float ab = a < b ? a : b, cd = c < d ? c : d;
  return ab < cd ? ab : cd;
}

// img: H x W row-major 8-bit, mask != 0 marks the pixels to fill (their values are ignored); filled in place.
inline void inpaint_u8(int H, int W, uint8_t* img, const uint8_t* mask, int range) {
  const int er = H + 2, ec = W + 2;
  Grid g{er, ec, std::vector<uint8_t>((size_t)er * ec, KNOWN), std::vector<float>((size_t)er * ec, 1.0e6f)};
  std::vector<uint8_t> m((size_t)er * ec, 0), band((size_t)er * ec, 0);
  auto M = [&](int i, int j) -> uint8_t& { return m[(size_t)i * ec + j]; };
  bool any = false;
  for (int i = 0; i < H; ++i)
    for (int j = 0; j < W; ++j)
      if (mask[(size_t)i * W + j]) {
        M(i + 1, j + 1) = 1;
        any = true;
      }
  if (!any) return;
  // band = cross-dilation of the mask minus the mask, border cleared
  for (int i = 1; i <= H; ++i)
    for (int j = 1; j <= W; ++j)
      if (!M(i, j) && (M(i - 1, j) || M(i + 1, j) || M(i, j - 1) || M(i, j + 1))) band[(size_t)i * ec + j] = 1;
  Heap heap, outheap;
  uint64_t seq = 0;
  for (int i = 0; i < er; ++i)
    for (int j = 0; j < ec; ++j) {
      if (band[(size_t)i * ec + j]) {
        g.F(i, j) = BAND;
        g.T(i, j) = 0.0f;
        heap.push({0.0f, seq, i, j});
        outheap.push({0.0f, seq, i, j});
        ++seq;
      } else if (m[(size_t)i * ec + j]) {
        g.F(i, j) = INSIDE;
      }
    }
  // T outside the mask (negated afterwards): march over the box dilation of the mask minus mask and band
  {
    Grid o{er, ec, std::vector<uint8_t>((size_t)er * ec, KNOWN), std::vector<float>()};
    for (int i = 1; i <= H; ++i)
      for (int j = 1; j <= W; ++j) {
        if (M(i, j) || band[(size_t)i * ec + j]) continue;
        bool near = false;
        for (int k = i - range; k <= i + range && !near; ++k)
          for (int l = j - range; l <= j + range; ++l)
            if (k >= 0 && l >= 0 && k < er && l < ec && m[(size_t)k * ec + l]) {
              near = true;
              break;
            }
        if (near) o.F(i, j) = INSIDE;
      }
    o.t.swap(g.t);   // the march reads and writes the shared T
    while (!outheap.empty()) {
      const HeapItem it = outheap.top();
      outheap.pop();
      const int ii = it.i, jj = it.j;
      o.F(ii, jj) = CHANGE;
      const int di[4] = {-1, 0, 1, 0}, dj[4] = {0, -1, 0, 1};
      for (int q = 0; q < 4; ++q) {
        const int i = ii + di[q], j = jj + dj[q];
        if (i <= 0 || j <= 0 || i > er - 1 || j > ec - 1) continue;
        if (i >= er - 1 || j >= ec - 1) continue;
        if (o.F(i, j) == INSIDE) {
          const float dist = fm_min4(o, i, j);
          o.T(i, j) = dist;
          o.F(i, j) = BAND;
          outheap.push({dist, seq++, i, j});
        }
      }
    }
    for (int i = 0; i < er; ++i)
      for (int j = 0; j < ec; ++j)
        if (o.F(i, j) == CHANGE && !band[(size_t)i * ec + j]) o.T(i, j) = -o.T(i, j);
    g.t.swap(o.t);
  }
  auto OUT = [&](int r, int c) -> float { return (float)img[(size_t)r * W + c]; };
  const int di[4] = {-1, 0, 1, 0}, dj[4] = {0, -1, 0, 1};
  while (!heap.empty()) {
    const HeapItem it = heap.top();
    heap.pop();
    const int ii = it.i, jj = it.j;
    g.F(ii, jj) = KNOWN;
    for (int q = 0; q < 4; ++q) {
      const int i = ii + di[q], j = jj + dj[q];
      if (i <= 0 || j <= 0 || i > er - 2 || j > ec - 2) continue;
      if (g.F(i, j) != INSIDE) continue;
      const float dist = fm_min4(g, i, j);
      g.T(i, j) = dist;
      float gTx, gTy;
      if (g.F(i, j + 1) != INSIDE) gTx = g.F(i, j - 1) != INSIDE ? (g.T(i, j + 1) - g.T(i, j - 1)) * 0.5f : (g.T(i, j + 1) - g.T(i, j));
      else gTx = g.F(i, j - 1) != INSIDE ? (g.T(i, j) - g.T(i, j - 1)) : 0.0f;
      if (g.F(i + 1, j) != INSIDE) gTy = g.F(i - 1, j) != INSIDE ? (g.T(i + 1, j) - g.T(i - 1, j)) * 0.5f : (g.T(i + 1, j) - g.T(i, j));
      else gTy = g.F(i - 1, j) != INSIDE ? (g.T(i, j) - g.T(i - 1, j)) : 0.0f;
      float Ia = 0, Jx = 0, Jy = 0, s = 1.0e-20f;
      for (int k = i - range; k <= i + range; ++k) {
        const int km = k - 1 + (k == 1), kp = k - 1 - (k == er - 2);
        for (int l = j - range; l <= j + range; ++l) {
          const int lm = l - 1 + (l == 1), lp = l - 1 - (l == ec - 2);
          if (!(k > 0 && l > 0 && k < er - 1 && l < ec - 1)) continue;
          if (g.F(k, l) == INSIDE || (l - j) * (l - j) + (k - i) * (k - i) > range * range) continue;
          const float ry = (float)(i - k), rx = (float)(j - l);
          const float len2 = rx * rx + ry * ry;
          const float dst = (float)(1.0 / (len2 * std::sqrt((double)len2)));
          const float lev = (float)(1.0 / (1 + std::fabs(g.T(k, l) - g.T(i, j))));
          float dir = rx * gTx + ry * gTy;
          if (std::fabs(dir) <= 0.01f) dir = 0.000001f;
          const float w = (float)std::fabs(dst * lev * dir);
          float gIx, gIy;
          if (g.F(k, l + 1) != INSIDE) gIx = g.F(k, l - 1) != INSIDE ? (OUT(km, lp + 1) - OUT(km, lm - 1)) * 2.0f : (OUT(km, lp + 1) - OUT(km, lm));
          else gIx = g.F(k, l - 1) != INSIDE ? (OUT(km, lp) - OUT(km, lm - 1)) : 0.0f;
          if (g.F(k + 1, l) != INSIDE) gIy = g.F(k - 1, l) != INSIDE ? (OUT(kp + 1, lm) - OUT(km - 1, lm)) * 2.0f : (OUT(kp + 1, lm) - OUT(km, lm));
          else gIy = g.F(k - 1, l) != INSIDE ? (OUT(kp, lm) - OUT(km - 1, lm)) : 0.0f;
          Ia += w * OUT(km, lm);
          Jx -= w * (gIx * rx);
          Jy -= w * (gIy * ry);
          s += w;
        }
      }
      const float sat = Ia / s + (Jx + Jy) / (std::sqrt(Jx * Jx + Jy * Jy) + 1.0e-20f) + 0.5f;
      const int v = (int)std::lrintf(sat);   // saturate_cast<uchar>(float): round to nearest, clamp
      img[(size_t)(i - 1) * W + (j - 1)] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
      g.F(i, j) = BAND;
      heap.push({dist, seq++, i, j});
    }
  }
}

}  // namespace artp_telea
