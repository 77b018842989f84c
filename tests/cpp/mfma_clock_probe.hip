// What the matrix cores of THIS part sustain: every SIMD of the chip issues independent v_mfma_f32_16x16x32_f16 back to back
// (no memory traffic at all) for tens of microseconds, like the 15 x 15 layer's main loop does.  Prints the rate by HIP events
// and the shader clock the wavefronts saw (s_memtime cycles / s_memrealtime): the nominal dense f16 peak (2.5 PFLOP/s) is
// 256 CUs x 4 SIMDs x 1024 FLOP/clk at 2.4 GHz; under a chip-wide MFMA load the clock does not stay there.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma_clock tests/cpp/mfma_clock_probe.hip && /tmp/mfma_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void __launch_bounds__(256) mfma_loop(int iters, float* out, unsigned long long* stamps) {
  half8 a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = (_Float16)(0.001f * (threadIdx.x & 7));
    b[j] = (_Float16)(0.002f * (threadIdx.x & 3));
  }
  floatx4 acc[NACC];
  for (int k = 0; k < NACC; ++k) acc[k] = floatx4{0.f, 0.f, 0.f, 0.f};
  const unsigned long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < NACC; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[k], 0, 0, 0);
  }
  const unsigned long long c1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
  for (int k = 0; k < NACC; ++k) s += acc[k][0] + acc[k][3];
  if (s == 12345.678f) out[0] = s;  // keeps the loop
  if (threadIdx.x == 0 && blockIdx.x < 1024) {
    stamps[2 * blockIdx.x] = c1 - c0;
    stamps[2 * blockIdx.x + 1] = w1 - w0;
  }
}

// the other f16 shape of gfx950: v_mfma_f32_32x32x16_f16 -- 32 768 FLOP in 8 passes, half the A / B operand reads per FLOP of
// 16x16x32 (MI355X_MICROARCH.md measured its 2 495 TFLOP/s with this one)
typedef float floatx16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ void __launch_bounds__(256) mfma_loop_32(int iters, float* out, unsigned long long* stamps) {
  half8 a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = (_Float16)(0.001f * (threadIdx.x & 7));
    b[j] = (_Float16)(0.002f * (threadIdx.x & 3));
  }
  floatx16 acc[NACC];
  for (int k = 0; k < NACC; ++k)
    for (int j = 0; j < 16; ++j) acc[k][j] = 0.f;
  const unsigned long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < NACC; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[k], 0, 0, 0);
  }
  const unsigned long long c1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
  for (int k = 0; k < NACC; ++k) s += acc[k][0] + acc[k][15];
  if (s == 12345.678f) out[0] = s;  // keeps the loop
  if (threadIdx.x == 0 && blockIdx.x < 1024) {
    stamps[2 * blockIdx.x] = c1 - c0;
    stamps[2 * blockIdx.x + 1] = w1 - w0;
  }
}

int main() {
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, 0) != hipSuccess || p.multiProcessorCount < 1) {
    fprintf(stderr, "mfma_clock_probe: no HIP device\n");
    return 1;
  }
  const int cus = p.multiProcessorCount;
  printf("device %s, %d CUs, clockRate %d kHz\n", p.gcnArchName, cus, p.clockRate);
  float* d_out;
  unsigned long long* d_st;
  (void)hipMalloc(&d_out, 4);
  (void)hipMalloc(&d_st, 2048 * sizeof(unsigned long long));
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  constexpr int NACC = 24;
  double last_tflops[3] = {0, 0, 0}, last_cpm[3] = {0, 0, 0}, last_ghz[3] = {0, 0, 0};
  for (int wgs_per_cu : {1, 2}) {
    for (int iters : {200, 1000, 5000}) {
      const int grid = cus * wgs_per_cu;
      hipLaunchKernelGGL(mfma_loop<NACC>, dim3(grid), dim3(256), 0, 0, iters, d_out, d_st);  // warm-up
      (void)hipDeviceSynchronize();
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(mfma_loop<NACC>, dim3(grid), dim3(256), 0, 0, iters, d_out, d_st);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms = 0;
      (void)hipEventElapsedTime(&ms, e0, e1);
      std::vector<unsigned long long> st(2048);
      (void)hipMemcpy(st.data(), d_st, 2048 * 8, hipMemcpyDeviceToHost);
      double cyc = 0, wall = 0;
      const int nrec = grid < 1024 ? grid : 1024;
      for (int i = 0; i < nrec; ++i) {
        cyc += (double)st[2 * i];
        wall += (double)st[2 * i + 1];
      }
      cyc /= nrec;
      wall /= nrec;
      const double mfmas_per_wave = (double)iters * NACC;
      const double flop = mfmas_per_wave * 16384.0 * 4.0 * grid;  // 4 wavefronts per workgroup
      printf("wavefronts/SIMD %d  %5d x %d MFMAs per wavefront: %8.1f us by events = %7.1f TFLOP/s = %.3f of 2500;  in the loop: %.0f cycles = "
             "%.2f cycles per MFMA and SIMD, %.2f us -> shader clock %.2f GHz\n",
             wgs_per_cu, iters, NACC, ms * 1e3, flop / (ms * 1e-3) / 1e12, flop / (ms * 1e-3) / 1e12 / 2500.0, cyc,
             cyc / (mfmas_per_wave * wgs_per_cu), wall * 0.01, cyc / (wall * 0.01) / 1e3);
      last_tflops[wgs_per_cu] = flop / (ms * 1e-3) / 1e12;
      last_cpm[wgs_per_cu] = cyc / (mfmas_per_wave * wgs_per_cu);
      last_ghz[wgs_per_cu] = cyc / (wall * 0.01) / 1e3;
    }
  }
  // ---- v_mfma_f32_32x32x16_f16, same protocol (8 independent accumulators of 16 registers) ----
  constexpr int NACC32 = 8;
  double t32[3] = {0, 0, 0}, cpm32[3] = {0, 0, 0}, ghz32[3] = {0, 0, 0};
  for (int wgs_per_cu : {1, 2}) {
    for (int iters : {300, 1500, 7500}) {
      const int grid = cus * wgs_per_cu;
      hipLaunchKernelGGL(mfma_loop_32<NACC32>, dim3(grid), dim3(256), 0, 0, iters, d_out, d_st);  // warm-up
      (void)hipDeviceSynchronize();
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(mfma_loop_32<NACC32>, dim3(grid), dim3(256), 0, 0, iters, d_out, d_st);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms = 0;
      (void)hipEventElapsedTime(&ms, e0, e1);
      std::vector<unsigned long long> st(2048);
      (void)hipMemcpy(st.data(), d_st, 2048 * 8, hipMemcpyDeviceToHost);
      double cyc = 0, wall = 0;
      const int nrec = grid < 1024 ? grid : 1024;
      for (int i = 0; i < nrec; ++i) {
        cyc += (double)st[2 * i];
        wall += (double)st[2 * i + 1];
      }
      cyc /= nrec;
      wall /= nrec;
      const double mfmas_per_wave = (double)iters * NACC32;
      const double flop = mfmas_per_wave * 32768.0 * 4.0 * grid;
      printf("32x32x16: wavefronts/SIMD %d  %5d x %d MFMAs per wavefront: %8.1f us by events = %7.1f TFLOP/s = %.3f of 2500;  in the loop: "
             "%.0f cycles = %.2f cycles per MFMA and SIMD, %.2f us -> shader clock %.2f GHz\n",
             wgs_per_cu, iters, NACC32, ms * 1e3, flop / (ms * 1e-3) / 1e12, flop / (ms * 1e-3) / 1e12 / 2500.0, cyc,
             cyc / (mfmas_per_wave * wgs_per_cu), wall * 0.01, cyc / (wall * 0.01) / 1e3);
      t32[wgs_per_cu] = flop / (ms * 1e-3) / 1e12;
      cpm32[wgs_per_cu] = cyc / (mfmas_per_wave * wgs_per_cu);
      ghz32[wgs_per_cu] = cyc / (wall * 0.01) / 1e3;
    }
  }
  printf("{\"shape\": \"32x32x16\", \"one_per_simd_tflops\": %.1f, \"one_per_simd_cycles_per_mfma\": %.2f, \"one_per_simd_clock_ghz\": %.3f, "
         "\"two_per_simd_tflops\": %.1f, \"two_per_simd_cycles_per_mfma\": %.2f, \"two_per_simd_clock_ghz\": %.3f}\n",
         t32[1], cpm32[1], ghz32[1], t32[2], cpm32[2], ghz32[2]);
  // the longest runs, machine readable (bench.py, tests/test_mfma_hazard.py): the LAST line, the 16x16x32 figures
  printf("{\"one_per_simd_tflops\": %.1f, \"one_per_simd_cycles_per_mfma\": %.2f, \"one_per_simd_clock_ghz\": %.3f, "
         "\"two_per_simd_tflops\": %.1f, \"two_per_simd_cycles_per_mfma\": %.2f, \"two_per_simd_clock_ghz\": %.3f, "
         "\"nominal_dense_f16_tflops\": 2500.0, \"cus\": %d}\n",
         last_tflops[1], last_cpm[1], last_ghz[1], last_tflops[2], last_cpm[2], last_ghz[2], cus);
  return 0;
}
