// Can the host write device memory directly (large BAR)?  Ping-pong round trip: request through device memory vs through
// mapped host memory; response always through mapped host memory.
#include <hip/hip_runtime.h>
#include <chrono>
#include <csetjmp>
#include <csignal>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <immintrin.h>
static sigjmp_buf jb;
static void on_segv(int) { siglongjmp(jb, 1); }
__global__ void pong(const volatile uint32_t* req, volatile uint32_t* resp, int lines) {
  const unsigned long long t0 = wall_clock64();
  uint32_t last = 0;
  for (;;) {
    uint32_t v = 0;
    if ((int)threadIdx.x < lines * 16) v = req[threadIdx.x];
    const uint32_t seq = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
    if (seq == 0xffffffffu || wall_clock64() - t0 > 300000000ull) return;
    if (seq != last) {
      last = seq;
      if (blockIdx.x == 0 && threadIdx.x == 0) resp[0] = seq;
    }
    __builtin_amdgcn_s_sleep(8);
  }
}
static double run(const char* name, volatile uint32_t* req_host_view, const uint32_t* req_dev, volatile uint32_t* resp, uint32_t* resp_dev,
                  int wgs, int lines, bool wc) {
  hipStream_t st;
  hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  req_host_view[0] = 0;
  if (wc) _mm_sfence();
  resp[0] = 0;
  hipLaunchKernelGGL(pong, dim3(wgs), dim3(64), 0, st, req_dev, resp_dev, lines);
  const int reps = 20000;
  // warm
  for (uint32_t s = 1; s <= 100; ++s) { req_host_view[0] = s; if (wc) _mm_sfence(); while (resp[0] != s) {} }
  auto t0 = std::chrono::steady_clock::now();
  for (uint32_t s = 101; s <= 100 + reps; ++s) {
    req_host_view[0] = s;
    if (wc) _mm_sfence();
    auto tw = std::chrono::steady_clock::now();
    while (resp[0] != s) {
      if (std::chrono::steady_clock::now() - tw > std::chrono::milliseconds(200)) { std::printf("%s: no answer\n", name); goto out; }
    }
  }
out:
  auto t1 = std::chrono::steady_clock::now();
  req_host_view[0] = 0xffffffffu;
  if (wc) _mm_sfence();
  hipStreamSynchronize(st);
  hipStreamDestroy(st);
  const double us = std::chrono::duration<double, std::micro>(t1 - t0).count() / reps;
  std::printf("%-40s wgs=%3d lines=%d: %.2f us per round trip\n", name, wgs, lines, us);
  return us;
}
int main() {
  void *hreq = nullptr, *hreq_d = nullptr, *hresp = nullptr, *hresp_d = nullptr;
  hipHostMalloc(&hreq, 4096, hipHostMallocMapped); hipHostGetDevicePointer(&hreq_d, hreq, 0);
  hipHostMalloc(&hresp, 4096, hipHostMallocMapped); hipHostGetDevicePointer(&hresp_d, hresp, 0);
  std::memset(hreq, 0, 4096); std::memset(hresp, 0, 4096);
  for (int wgs : {1, 32, 64}) for (int lines : {1, 3})
    run("request in mapped host memory", (volatile uint32_t*)hreq, (const uint32_t*)hreq_d, (volatile uint32_t*)hresp, (uint32_t*)hresp_d, wgs, lines, false);
  void* dplain = nullptr; void* dfine = nullptr;
  hipMalloc(&dplain, 4096); hipMemset(dplain, 0, 4096);
  hipError_t e = hipExtMallocWithFlags(&dfine, 4096, hipDeviceMallocFinegrained);
  std::printf("hipExtMallocWithFlags(finegrained): %s\n", hipGetErrorString(e));
  if (e == hipSuccess) hipMemset(dfine, 0, 4096);
  hipDeviceSynchronize();
  std::signal(SIGSEGV, on_segv); std::signal(SIGBUS, on_segv);
  for (int which = 0; which < 2; ++which) {
    void* d = which ? dfine : dplain;
    if (!d) continue;
    const char* nm = which ? "request in fine-grained device memory" : "request in hipMalloc device memory";
    if (sigsetjmp(jb, 1)) { std::printf("%s: host write faults (no CPU mapping)\n", nm); continue; }
    ((volatile uint32_t*)d)[1] = 7;   // faults here if not mapped
    _mm_sfence();
    std::printf("%s: host write did not fault; read back %u\n", nm, ((volatile uint32_t*)d)[1]);
    for (int wgs : {1, 32, 64}) for (int lines : {1, 3})
      run(nm, (volatile uint32_t*)d, (const uint32_t*)d, (volatile uint32_t*)hresp, (uint32_t*)hresp_d, wgs, lines, true);
  }
  return 0;
}
