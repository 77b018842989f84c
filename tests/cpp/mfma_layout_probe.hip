// Operand / result layout of v_mfma_f32_32x32x16_f16 on gfx950, checked against the layout cost_kernels.h assumes:
//   first operand  (M x K = 32 x 16): lane l holds A[m = l % 32][k = 8 (l / 32) + i], i = 0..7
//   second operand (K x N = 16 x 32): lane l holds B[k = 8 (l / 32) + i][n = l % 32]
//   result (32 x 32): lane l, register v holds D[m = 8 (v / 4) + 4 (l / 32) + v % 4][n = l % 32]
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma_layout tests/cpp/mfma_layout_probe.hip && /tmp/mfma_layout
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

__global__ void probe(const float* A, const float* B, float* D) {  // A [32][16], B [16][32], D [32][32] row-major
  const int l = threadIdx.x;
  half8 a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = (_Float16)A[(l % 32) * 16 + 8 * (l / 32) + i];
    b[i] = (_Float16)B[(8 * (l / 32) + i) * 32 + l % 32];
  }
  floatx16 acc;
  for (int v = 0; v < 16; ++v) acc[v] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  for (int v = 0; v < 16; ++v) D[(8 * (v / 4) + 4 * (l / 32) + v % 4) * 32 + l % 32] = acc[v];
}

int main() {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n < 1) {
    fprintf(stderr, "mfma_layout_probe: no HIP device\n");
    return 1;
  }
  std::vector<float> A(32 * 16), B(16 * 32), D(32 * 32), R(32 * 32, 0.f);
  for (int i = 0; i < 32 * 16; ++i) A[i] = (float)((i * 7 + 3) % 13 - 6);
  for (int i = 0; i < 16 * 32; ++i) B[i] = (float)((i * 5 + 1) % 11 - 5);
  for (int m = 0; m < 32; ++m)
    for (int nn = 0; nn < 32; ++nn)
      for (int k = 0; k < 16; ++k) R[m * 32 + nn] += A[m * 16 + k] * B[k * 32 + nn];
  float *dA, *dB, *dD;
  (void)hipMalloc(&dA, A.size() * 4);
  (void)hipMalloc(&dB, B.size() * 4);
  (void)hipMalloc(&dD, D.size() * 4);
  (void)hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  (void)hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 32 * 32; ++i) bad += std::fabs(D[i] - R[i]) > 1e-3f;
  printf("v_mfma_f32_32x32x16_f16 layout: %d of 1024 elements differ from A x B\n", bad);
  return bad ? 2 : 0;
}
