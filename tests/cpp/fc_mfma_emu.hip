// Host-side check of the MFMA form of the cost-query MLP (cost_kernels.h fc_cost_mfma_kernel): the REAL fragment blob
// (fc_mfma_pack, artp_capi.hip) is run through a literal emulation of the kernel's tile arithmetic -- v_mfma operand and
// accumulator lane layouts, the hi / lo half-float pairs, the hidden-unit permutation that makes the first GEMM's
// accumulators the second GEMM's operand -- and compared with the plain MLP in double.  No GPU involved: it runs in the
// CPU test stage (tests/test_fc_mfma_pack.py) and is what separated "the packing / indexing is right" from "the device
// does something else" when the kernel was brought up.
#include "artp_capi.hip"
#include <random>
static float h2f(uint16_t h) {
  const uint32_t sign = (h >> 15) & 1u, ex = (h >> 10) & 31u, man = h & 1023u;
  double r;
  if (ex == 0) r = std::ldexp((double)man, -24);
  else r = std::ldexp((double)(man | 1024u), (int)ex - 25);
  return (float)(sign ? -r : r);
}
// D[i][j] += sum_k A[i][k] B[k][j]; A frag: lane l: i = l&15, k = KPL*(l>>4)+e ; B frag: lane l: j = l&15, same k; D: lane l: j = l&15, i = 4*(l>>4)+r
template <int KPL>
static void mfma(const float (*A)[8], const float (*Bf)[8], float (*D)[4]) {
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      double s = 0;
      for (int g = 0; g < 4; ++g)
        for (int e = 0; e < KPL; ++e) s += (double)A[16 * g + i][e] * (double)Bf[16 * g + j][e];
      D[16 * (i / 4) + j][i % 4] += (float)s;
    }
}
int main() {
  std::mt19937 gen(5);
  std::normal_distribution<float> nd(0.f, 0.3f);
  std::vector<float> w(FcWeights::TOTAL);
  for (auto& v : w) v = nd(gen);
  std::vector<unsigned char> blob;
  fc_mfma_pack(w.data(), &blob);
  auto frag8 = [&](size_t off, float (*out)[8]) { for (int l = 0; l < 64; ++l) for (int e = 0; e < 8; ++e) { uint16_t b; std::memcpy(&b, blob.data() + off + l * 16 + e * 2, 2); out[l][e] = h2f(b); } };
  auto frag4 = [&](size_t off, float (*out)[8]) { for (int l = 0; l < 64; ++l) for (int e = 0; e < 8; ++e) { out[l][e] = 0; if (e < 4) { uint16_t b; std::memcpy(&b, blob.data() + off + l * 8 + e * 2, 2); out[l][e] = h2f(b); } } };
  // 16 edges: features (exact halfs), t inputs
  float feat[16][48], t[16][16];
  for (int n = 0; n < 16; ++n) { for (int k = 0; k < 48; ++k) feat[n][k] = h2f(f32_to_f16_bits(nd(gen) * 5)); for (int k = 0; k < 16; ++k) t[n][k] = k < 10 ? nd(gen) * 4 : (k == 10 ? 1.f : 0.f); }
  // reference
  double ref[16][3];
  for (int n = 0; n < 16; ++n) {
    double x[64], h[48];
    for (int k = 0; k < 48; ++k) x[k] = feat[n][k];
    for (int o = 0; o < 16; ++o) { double a = w[FcWeights::TAR0_B + o]; for (int k = 0; k < 10; ++k) a += (double)t[n][k] * w[FcWeights::TAR0_W + o * 10 + k]; x[48 + o] = a; }
    for (int o = 0; o < 48; ++o) { double a = w[FcWeights::OUT0_B + o]; for (int k = 0; k < 64; ++k) a += x[k] * w[FcWeights::OUT0_W + o * 64 + k]; h[o] = a > 0 ? a : 0.3 * a; }
    double p = w[FcWeights::O1_B], q = w[FcWeights::O2_B], r = w[FcWeights::O3_B];
    for (int o = 0; o < 24; ++o) { double a = w[FcWeights::H1_B + o], c = w[FcWeights::H2_B + o]; for (int k = 0; k < 48; ++k) { a += h[k] * w[FcWeights::H1_W + o * 48 + k]; c += h[k] * w[FcWeights::H2_W + o * 48 + k]; } a = a > 0 ? a : 0.3 * a; c = c > 0 ? c : 0.3 * c; p += a * w[FcWeights::O1_W + o]; q += c * w[FcWeights::O2_W + o]; }
    for (int o = 0; o < 36; ++o) { double a = w[FcWeights::H3_B + o]; for (int k = 0; k < 48; ++k) a += h[k] * w[FcWeights::H3_W + o * 48 + k]; a = a > 0 ? a : 0.3 * a; r += a * w[FcWeights::O3_W + o]; }
    ref[n][0] = p; ref[n][1] = q; ref[n][2] = r;
  }
  // the kernel's tile, literally
  static float x0[64][8], x1[64][8], x1lo[64][8], W[64][8];
  for (int l = 0; l < 64; ++l) {
    const int li = l & 15, kg = l >> 4;
    for (int e = 0; e < 8; ++e) {
      x0[l][e] = feat[li][8 * kg + e];
      if (kg < 2) { x1[l][e] = feat[li][32 + 8 * kg + e]; x1lo[l][e] = 0; }
      else { const float v = t[li][8 * (kg - 2) + e]; const float hi = h2f(f32_to_f16_bits(v)); x1[l][e] = hi; x1lo[l][e] = h2f(f32_to_f16_bits(v - hi)); }
    }
  }
  static float a1[3][64][4];
  for (int tt = 0; tt < 3; ++tt) {
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) a1[tt][l][r] = 0;
    const float (*ops[5])[8] = {x0, x1, x1lo, x0, x1};
    const int sidx[5] = {0, 1, 1, 0, 1}, pidx[5] = {1, 1, 0, 0, 0};
    for (int m = 0; m < 5; ++m) { frag8(FcMfma::G1 + (size_t)((sidx[m] * 3 + tt) * 2 + pidx[m]) * 1024, W); mfma<8>(W, ops[m], a1[tt]); }
  }
  static float hA[64][8], hAlo[64][8], hB[64][8], hBlo[64][8];
  for (int l = 0; l < 64; ++l)
    for (int j = 0; j < 4; ++j) {
      float v0 = a1[0][l][j], v1 = a1[1][l][j], v2 = a1[2][l][j];
      v0 = v0 > 0 ? v0 : 0.3f * v0; v1 = v1 > 0 ? v1 : 0.3f * v1; v2 = v2 > 0 ? v2 : 0.3f * v2;
      hA[l][j] = h2f(f32_to_f16_bits(v0)); hAlo[l][j] = h2f(f32_to_f16_bits(v0 - hA[l][j]));
      hA[l][4 + j] = h2f(f32_to_f16_bits(v1)); hAlo[l][4 + j] = h2f(f32_to_f16_bits(v1 - hA[l][4 + j]));
      hB[l][j] = h2f(f32_to_f16_bits(v2)); hBlo[l][j] = h2f(f32_to_f16_bits(v2 - hB[l][j])); hB[l][4 + j] = hBlo[l][4 + j] = 0;
    }
  const float* bias2 = reinterpret_cast<const float*>(blob.data() + FcMfma::BIAS2);
  const float* outw = reinterpret_cast<const float*>(blob.data() + FcMfma::OUT);
  const float* ob = reinterpret_cast<const float*>(blob.data() + FcMfma::OB);
  static float acc[64][4];
  float p[64] = {0}, q[64] = {0}, r[64] = {0};
  for (int tt = 0; tt < 6; ++tt) {
    for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) acc[l][i] = bias2[16 * tt + 4 * (l >> 4) + i];
    static float Wa[64][8];
    frag8(FcMfma::G2A + (size_t)(tt * 2 + 1) * 1024, Wa); mfma<8>(Wa, hA, acc);
    frag8(FcMfma::G2A + (size_t)(tt * 2 + 0) * 1024, Wa); mfma<8>(Wa, hAlo, acc); mfma<8>(Wa, hA, acc);
    frag4(FcMfma::G2B + (size_t)(tt * 2 + 1) * 512, Wa); mfma<4>(Wa, hB, acc);
    frag4(FcMfma::G2B + (size_t)(tt * 2 + 0) * 512, Wa); mfma<4>(Wa, hBlo, acc); mfma<4>(Wa, hB, acc);
    for (int l = 0; l < 64; ++l)
      for (int i = 0; i < 4; ++i) {
        float a = acc[l][i]; a = a > 0 ? a : 0.3f * a;
        const int u = 16 * tt + 4 * (l >> 4) + i;
        if (tt <= 1) p[l] += a * outw[0 * 96 + u];
        if (tt >= 1 && tt <= 2) q[l] += a * outw[1 * 96 + u];
        if (tt >= 3) r[l] += a * outw[2 * 96 + u];
      }
  }
  double worst = 0;
  for (int n = 0; n < 16; ++n) {
    const double pp = p[n] + p[n + 16] + p[n + 32] + p[n + 48] + ob[0], qq = q[n] + q[n + 16] + q[n + 32] + q[n + 48] + ob[1], rr = r[n] + r[n + 16] + r[n + 32] + r[n + 48] + ob[2];
    worst = std::max(worst, std::max(std::fabs(pp - ref[n][0]), std::max(std::fabs(qq - ref[n][1]), std::fabs(rr - ref[n][2]))));
    if (n < 3) std::printf("edge %d: %.5f %.5f %.5f  ref %.5f %.5f %.5f\n", n, pp, qq, rr, ref[n][0], ref[n][1], ref[n][2]);
  }
  std::printf("worst |emulated kernel - reference| = %.3g\n", worst);
  return worst < 2e-5 ? 0 : 1;
}
