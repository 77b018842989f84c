// mfma_hazard_probe.hip -- measures, on the device, how many wait states gfx950 needs between a
// v_mfma_f32_16x16x32_f16 and the instructions that may collide with it.  Test infrastructure (tests/test_mfma_hazard.py):
// DESIGN 4.4 claims that ROCm 7.2 pads some of these hazards too short for the 8-pass 16x16x32 MFMAs; the probes are
// written in inline assembly with fixed registers, so the distance K is exactly what is on the page (the compiler's
// hazard recogniser does not look inside an asm block).
//
//   A  srcC write-after-read:   MFMA(dst D, srcC C != D); K wait states; VALU overwrites C      (the round-4 bug)
//   B  dependent MFMA, other shape: 16x16x32 -> K -> 16x16x16 accumulating into the same registers
//   B3 dependent MFMA, other shape: 16x16x16 -> K -> 16x16x32
//   B2 dependent MFMA, same shape:  16x16x32 -> K -> 16x16x32
//   C  VALU reads the MFMA's result after K wait states
//   E  VALU overwrites the MFMA's destination after K wait states (the MFMA's own write must not land later)
// Each probe runs with PRE = 0 and PRE = four independent MFMAs issued right in front (a busy matrix pipe delays the
// probed instruction's passes).  Reference = the same sequence with 64 wait states.  Output: one line per probe and PRE with
// the result for K = 0 .. 15 ('.' = exact, 'X' = wrong) and "min_safe=<K>".
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

#define NOPS16 "s_nop 15\n"
#define LONG_WAIT NOPS16 NOPS16 NOPS16 NOPS16
#define PRE_MFMA                                                        \
  "v_mfma_f32_16x16x32_f16 v[108:111], %[a], %[b], v[108:111]\n"        \
  "v_mfma_f32_16x16x32_f16 v[112:115], %[a], %[b], v[112:115]\n"        \
  "v_mfma_f32_16x16x32_f16 v[116:119], %[a], %[b], v[116:119]\n"        \
  "v_mfma_f32_16x16x32_f16 v[120:123], %[a], %[b], v[120:123]\n"
#define CLOB "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", \
             "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123"
#define SETUP                                                                                   \
  "v_mov_b32 v100, %[c0]\n v_mov_b32 v101, %[c1]\n v_mov_b32 v102, %[c2]\n v_mov_b32 v103, %[c3]\n" \
  "v_mov_b32 v108, 0\n v_mov_b32 v109, 0\n v_mov_b32 v110, 0\n v_mov_b32 v111, 0\n"                 \
  "v_mov_b32 v112, 0\n v_mov_b32 v113, 0\n v_mov_b32 v114, 0\n v_mov_b32 v115, 0\n"                 \
  "v_mov_b32 v116, 0\n v_mov_b32 v117, 0\n v_mov_b32 v118, 0\n v_mov_b32 v119, 0\n"                 \
  "v_mov_b32 v120, 0\n v_mov_b32 v121, 0\n v_mov_b32 v122, 0\n v_mov_b32 v123, 0\n" NOPS16
#define RESULT "v_mov_b32 %[d0], v104\n v_mov_b32 %[d1], v105\n v_mov_b32 %[d2], v106\n v_mov_b32 %[d3], v107\n"
#define OUTS [d0] "=&v"(d[0]), [d1] "=&v"(d[1]), [d2] "=&v"(d[2]), [d3] "=&v"(d[3])
#define INS [a] "v"(av), [b] "v"(bv), [a4] "v"(a4), [b4] "v"(b4), [c0] "v"(cv[0]), [c1] "v"(cv[1]), [c2] "v"(cv[2]), [c3] "v"(cv[3])

// K wait states = K - 1 as the operand of one s_nop (K = 0: nothing); K = 64: the reference
#define GAP(K) "s_nop " #K "\n"

template <int PROBE, int K, int PRE>
__global__ void probe_kernel(const half8* __restrict__ a, const half8* __restrict__ b, const floatx4* __restrict__ c,
                             floatx4* __restrict__ out) {
  const int l = threadIdx.x;
  const half8 av = a[l], bv = b[l];
  const half4 a4 = {av[0], av[1], av[2], av[3]}, b4 = {bv[0], bv[1], bv[2], bv[3]};
  const floatx4 cv = c[l];
  float d[4];
#define BODY(GAPSTR)                                                                                                        \
  if (PROBE == 0) {                                                                                                         \
    if (PRE)                                                                                                                \
      asm volatile(SETUP PRE_MFMA "v_mfma_f32_16x16x32_f16 v[104:107], %[a], %[b], v[100:103]\n" \
                   GAPSTR "v_mov_b32 v100, 0x7fc00000\n v_mov_b32 v101, 0x7fc00000\n v_mov_b32 v102, 0x7fc00000\n"          \
                          "v_mov_b32 v103, 0x7fc00000\n" LONG_WAIT RESULT : OUTS : INS : CLOB);                             \
    else                                                                                                                    \
      asm volatile(SETUP "v_mfma_f32_16x16x32_f16 v[104:107], %[a], %[b], v[100:103]\n"                                     \
                   GAPSTR "v_mov_b32 v100, 0x7fc00000\n v_mov_b32 v101, 0x7fc00000\n v_mov_b32 v102, 0x7fc00000\n"          \
                          "v_mov_b32 v103, 0x7fc00000\n" LONG_WAIT RESULT : OUTS : INS : CLOB);                             \
  } else if (PROBE == 1) {                                                                                                  \
    if (PRE)                                                                                                                \
      asm volatile(SETUP PRE_MFMA "v_mfma_f32_16x16x32_f16 v[104:107], %[a], %[b], v[100:103]\n" \
                   GAPSTR "v_mfma_f32_16x16x16_f16 v[104:107], %[a4], %[b4], v[104:107]\n" LONG_WAIT RESULT : OUTS : INS : CLOB); \
    else                                                                                                                    \
      asm volatile(SETUP "v_mfma_f32_16x16x32_f16 v[104:107], %[a], %[b], v[100:103]\n"                                     \
                   GAPSTR "v_mfma_f32_16x16x16_f16 v[104:107], %[a4], %[b4], v[104:107]\n" LONG_WAIT RESULT : OUTS : INS : CLOB); \
  } else if (PROBE == 2) {                                                                                                  \
    if (PRE)                                                                                                                \
      asm volatile(SETUP PRE_MFMA "v_mfma_f32_16x16x32_f16 v[104:107], %[a], %[b], v[100:103]\n" \
                   GAPSTR "v_mfma_f32_16x16x32_f16 v[104:107], %[b], %[a], v[104:107]\n" LONG_WAIT RESULT : OUTS : INS : CLOB); \
    else                                                                                                                    \
      asm volatile(SETUP "v_mfma_f32_16x16x32_f16 v[104:107], %[a], %[b], v[100:103]\n"                                     \
                   GAPSTR "v_mfma_f32_16x16x32_f16 v[104:107], %[b], %[a], v[104:107]\n" LONG_WAIT RESULT : OUTS : INS : CLOB); \
  } else if (PROBE == 3) {                                                                                                  \
    if (PRE)                                                                                                                \
      asm volatile(SETUP "v_mov_b32 v104, 0\n v_mov_b32 v105, 0\n v_mov_b32 v106, 0\n v_mov_b32 v107, 0\n" NOPS16           \
                   PRE_MFMA "v_mfma_f32_16x16x32_f16 v[104:107], %[a], %[b], v[100:103]\n"       \
                   GAPSTR RESULT LONG_WAIT : OUTS : INS : CLOB);                                                            \
    else                                                                                                                    \
      asm volatile(SETUP "v_mov_b32 v104, 0\n v_mov_b32 v105, 0\n v_mov_b32 v106, 0\n v_mov_b32 v107, 0\n" NOPS16           \
                   "v_mfma_f32_16x16x32_f16 v[104:107], %[a], %[b], v[100:103]\n" GAPSTR RESULT LONG_WAIT : OUTS : INS : CLOB); \
  } else if (PROBE == 5) {                                                                                                  \
    if (PRE)                                                                                                                \
      asm volatile(SETUP PRE_MFMA "v_mfma_f32_16x16x16_f16 v[104:107], %[a4], %[b4], v[100:103]\n"                          \
                   GAPSTR "v_mfma_f32_16x16x32_f16 v[104:107], %[a], %[b], v[104:107]\n" LONG_WAIT RESULT : OUTS : INS : CLOB); \
    else                                                                                                                    \
      asm volatile(SETUP "v_mfma_f32_16x16x16_f16 v[104:107], %[a4], %[b4], v[100:103]\n"                                   \
                   GAPSTR "v_mfma_f32_16x16x32_f16 v[104:107], %[a], %[b], v[104:107]\n" LONG_WAIT RESULT : OUTS : INS : CLOB); \
  } else {                                                                                                                  \
    if (PRE)                                                                                                                \
      asm volatile(SETUP PRE_MFMA "v_mfma_f32_16x16x32_f16 v[104:107], %[a], %[b], v[100:103]\n" \
                   GAPSTR "v_mov_b32 v104, 1.0\n v_mov_b32 v105, 1.0\n v_mov_b32 v106, 1.0\n v_mov_b32 v107, 1.0\n"         \
                   LONG_WAIT RESULT : OUTS : INS : CLOB);                                                                   \
    else                                                                                                                    \
      asm volatile(SETUP "v_mfma_f32_16x16x32_f16 v[104:107], %[a], %[b], v[100:103]\n"                                     \
                   GAPSTR "v_mov_b32 v104, 1.0\n v_mov_b32 v105, 1.0\n v_mov_b32 v106, 1.0\n v_mov_b32 v107, 1.0\n"         \
                   LONG_WAIT RESULT : OUTS : INS : CLOB);                                                                   \
  }
  if (K == 0) {
    BODY("")
  } else if (K == 1) {
    BODY(GAP(0))
  } else if (K == 2) {
    BODY(GAP(1))
  } else if (K == 3) {
    BODY(GAP(2))
  } else if (K == 4) {
    BODY(GAP(3))
  } else if (K == 5) {
    BODY(GAP(4))
  } else if (K == 6) {
    BODY(GAP(5))
  } else if (K == 7) {
    BODY(GAP(6))
  } else if (K == 8) {
    BODY(GAP(7))
  } else if (K == 9) {
    BODY(GAP(8))
  } else if (K == 10) {
    BODY(GAP(9))
  } else if (K == 11) {
    BODY(GAP(10))
  } else if (K == 12) {
    BODY(GAP(11))
  } else if (K == 13) {
    BODY(GAP(12))
  } else if (K == 14) {
    BODY(GAP(13))
  } else if (K == 15) {
    BODY(GAP(14))
  } else {
    BODY(NOPS16 NOPS16 NOPS16 NOPS16)
  }
#undef BODY
  out[l] = floatx4{d[0], d[1], d[2], d[3]};
}

#define HIPCHK(x)                                                                          \
  do {                                                                                     \
    hipError_t e_ = (x);                                                                   \
    if (e_ != hipSuccess) {                                                                \
      std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                         \
      return 2;                                                                            \
    }                                                                                      \
  } while (0)

struct Bufs {
  half8 *a, *b;
  floatx4 *c, *out;
};

template <int PROBE, int K, int PRE>
static int run_one(const Bufs& d, std::vector<float>* host) {
  hipLaunchKernelGGL((probe_kernel<PROBE, K, PRE>), dim3(1), dim3(64), 0, 0, d.a, d.b, d.c, d.out);
  HIPCHK(hipGetLastError());
  HIPCHK(hipDeviceSynchronize());
  host->resize(256);
  HIPCHK(hipMemcpy(host->data(), d.out, 256 * sizeof(float), hipMemcpyDeviceToHost));
  return 0;
}

template <int PROBE, int PRE, int K>
static int sweep(const Bufs& d, const std::vector<float>& ref, char* marks, int* min_safe) {
  if constexpr (K <= 15) {
    std::vector<float> h;
    if (int rc = run_one<PROBE, K, PRE>(d, &h)) return rc;
    // three repetitions: a hazard can be timing dependent
    bool ok = std::memcmp(h.data(), ref.data(), 256 * sizeof(float)) == 0;
    for (int rep = 0; rep < 2 && ok; ++rep) {
      if (int rc = run_one<PROBE, K, PRE>(d, &h)) return rc;
      ok = std::memcmp(h.data(), ref.data(), 256 * sizeof(float)) == 0;
    }
    marks[K] = ok ? '.' : 'X';
    if (!ok) *min_safe = K + 1;
    return sweep<PROBE, PRE, K + 1>(d, ref, marks, min_safe);
  } else {
    return 0;
  }
}

template <int PROBE, int PRE>
static int probe(const Bufs& d, const char* name) {
  std::vector<float> ref;
  if (int rc = run_one<PROBE, 64, PRE>(d, &ref)) return rc;
  char marks[17] = {0};
  int min_safe = 0;
  if (int rc = sweep<PROBE, PRE, 0>(d, ref, marks, &min_safe)) return rc;
  std::printf("probe %-26s pre_mfma=%d  K=0..15 %s  min_safe=%d\n", name, PRE, marks, min_safe);
  return 0;
}

int main() {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n == 0) {
    std::printf("no GPU\n");
    return 3;
  }
  std::vector<_Float16> ha(512), hb(512);
  std::vector<float> hc(256);
  unsigned s = 12345u;
  auto rnd = [&]() {
    s = s * 1664525u + 1013904223u;
    return (float)((s >> 8) & 0xffff) / 65536.0f - 0.5f;
  };
  for (auto& v : ha) v = (_Float16)rnd();
  for (auto& v : hb) v = (_Float16)rnd();
  for (auto& v : hc) v = 100.0f * rnd();
  Bufs d;
  HIPCHK(hipMalloc((void**)&d.a, 1024));
  HIPCHK(hipMalloc((void**)&d.b, 1024));
  HIPCHK(hipMalloc((void**)&d.c, 1024));
  HIPCHK(hipMalloc((void**)&d.out, 1024));
  HIPCHK(hipMemcpy(d.a, ha.data(), 1024, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d.b, hb.data(), 1024, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d.c, hc.data(), 1024, hipMemcpyHostToDevice));
  hipDeviceProp_t p;
  HIPCHK(hipGetDeviceProperties(&p, 0));
  std::printf("device %s\n", p.gcnArchName);
  int rc = 0;
  rc |= probe<0, 0>(d, "A srcC-WAR (VALU write)");
  rc |= probe<0, 1>(d, "A srcC-WAR (VALU write)");
  rc |= probe<1, 0>(d, "B mfma32 -> mfma16 same acc");
  rc |= probe<1, 1>(d, "B mfma32 -> mfma16 same acc");
  rc |= probe<5, 0>(d, "B3 mfma16 -> mfma32 same acc");
  rc |= probe<5, 1>(d, "B3 mfma16 -> mfma32 same acc");
  rc |= probe<2, 0>(d, "B2 mfma32 -> mfma32 same acc");
  rc |= probe<2, 1>(d, "B2 mfma32 -> mfma32 same acc");
  rc |= probe<3, 0>(d, "C VALU reads result");
  rc |= probe<3, 1>(d, "C VALU reads result");
  rc |= probe<4, 0>(d, "E VALU overwrites dst");
  rc |= probe<4, 1>(d, "E VALU overwrites dst");
  return rc;
}
