// TEST DOUBLE, not product code: a stand-in for librccl.so that lets SEVERAL PROCESSES form a communicator on ONE GPU
// (the real RCCL refuses: "Duplicate GPU detected").  Selected with $ARTP_RCCL_LIB, it sits behind exactly the call
// sites libartp.so's device groups use (csrc/group.h: ncclGetUniqueId, ncclCommInitRank, ncclAllGather, ncclAllReduce,
// ncclGroupStart / End, ncclCommGetAsyncError, ncclCommAbort / Destroy), so that artp_group_create_rank, the id hand-off,
// the rank offsets of the gathered blocks and the double-buffer events run with real inter-process concurrency on a
// one-GPU box.  Transport: every rank owns a device staging buffer, published through a POSIX
// shared-memory segment named after the unique id as a hipIpcMemHandle; an all-gather = copy my block to my staging
// buffer, barrier, copy every rank's staging buffer into my receive buffer, barrier.  The calls BLOCK the calling host
// thread (the real library's are asynchronous): correct, just slower.
//   g++ -O2 -fPIC -shared -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -o libloopback_rccl.so loopback_rccl.cpp -L/opt/rocm/lib -lamdhip64 -lrt
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>

#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4,
               ncclInvalidUsage = 5, ncclRemoteError = 6, ncclInProgress = 7 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6,
               ncclFloat32 = 7, ncclFloat64 = 8 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
}

namespace {
constexpr int MAX_RANKS = 16;
constexpr size_t STAGING_BYTES = 8u << 20;
constexpr int BARRIER_TIMEOUT_S = 60;

struct Shared {   // lives in the shared-memory segment; zero-initialised by ftruncate
  std::atomic<uint32_t> count, generation;
  std::atomic<uint32_t> ready[MAX_RANKS];
  hipIpcMemHandle_t staging[MAX_RANKS];
  uint64_t values[MAX_RANKS];
};
}  // namespace

struct ncclComm {
  int rank = 0, n = 0, device = 0;
  Shared* sh = nullptr;
  char shm_name[64] = {0};
  void* staging = nullptr;
  void* peer[MAX_RANKS] = {nullptr};
  bool dead = false;
};

namespace {
bool barrier(ncclComm* c) {
  Shared* s = c->sh;
  const uint32_t gen = s->generation.load(std::memory_order_acquire);
  if (s->count.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)c->n) {
    s->count.store(0, std::memory_order_relaxed);
    s->generation.fetch_add(1, std::memory_order_acq_rel);
    return true;
  }
  const auto t0 = std::chrono::steady_clock::now();
  while (s->generation.load(std::memory_order_acquire) == gen) {
    if (c->dead) return false;
    if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(BARRIER_TIMEOUT_S)) return false;
    std::this_thread::sleep_for(std::chrono::microseconds(20));
  }
  return true;
}
size_t dtype_bytes(ncclDataType_t t) {
  switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    default: return 8;
  }
}
}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  if (!id) return ncclInvalidArgument;
  std::memset(id, 0, sizeof(*id));
  const int fd = open("/dev/urandom", O_RDONLY);
  if (fd < 0 || read(fd, id->internal, 16) != 16) {
    if (fd >= 0) close(fd);
    return ncclSystemError;
  }
  close(fd);
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank) {
  if (!out || nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  ncclComm* c = new ncclComm();
  c->rank = rank;
  c->n = nranks;
  if (hipGetDevice(&c->device) != hipSuccess) { delete c; return ncclUnhandledCudaError; }
  unsigned long long tag;
  std::memcpy(&tag, id.internal, 8);
  std::snprintf(c->shm_name, sizeof(c->shm_name), "/artp_loopback_rccl_%016llx", tag);
  const int fd = shm_open(c->shm_name, O_CREAT | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, sizeof(Shared)) != 0) { if (fd >= 0) close(fd); delete c; return ncclSystemError; }
  void* m = mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (m == MAP_FAILED) { delete c; return ncclSystemError; }
  c->sh = static_cast<Shared*>(m);
  if (hipMalloc(&c->staging, STAGING_BYTES) != hipSuccess ||
      hipIpcGetMemHandle(&c->sh->staging[rank], c->staging) != hipSuccess) {
    std::fprintf(stderr, "loopback_rccl: staging buffer / IPC handle failed on rank %d\n", rank);
    delete c;
    return ncclUnhandledCudaError;
  }
  c->sh->ready[rank].store(1, std::memory_order_release);
  const auto t0 = std::chrono::steady_clock::now();
  for (int p = 0; p < nranks; ++p)
    while (!c->sh->ready[p].load(std::memory_order_acquire)) {
      if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(BARRIER_TIMEOUT_S)) { delete c; return ncclSystemError; }
      std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
  for (int p = 0; p < nranks; ++p) {
    if (p == rank) { c->peer[p] = c->staging; continue; }
    if (hipIpcOpenMemHandle(&c->peer[p], c->sh->staging[p], hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
      std::fprintf(stderr, "loopback_rccl: hipIpcOpenMemHandle of rank %d failed on rank %d (HSA_ENABLE_IPC_MODE_LEGACY=0?)\n", p, rank);
      delete c;
      return ncclUnhandledCudaError;
    }
  }
  if (!barrier(c)) { delete c; return ncclSystemError; }
  *out = c;
  return ncclSuccess;
}

ncclResult_t ncclCommInitAll(ncclComm_t*, int, const int*) { return ncclInvalidUsage; }   // one process per rank only

static ncclResult_t teardown(ncclComm* c, bool orderly) {
  if (!c) return ncclInvalidArgument;
  if (orderly) (void)barrier(c);   // nobody still reads my staging buffer
  c->dead = true;
  for (int p = 0; p < c->n; ++p)
    if (p != c->rank && c->peer[p]) (void)hipIpcCloseMemHandle(c->peer[p]);
  if (c->staging) (void)hipFree(c->staging);
  if (c->sh) munmap(c->sh, sizeof(Shared));
  if (c->rank == 0) shm_unlink(c->shm_name);
  delete c;
  return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t c) { return teardown(c, true); }
ncclResult_t ncclCommAbort(ncclComm_t c) { return teardown(c, false); }
ncclResult_t ncclCommGetAsyncError(ncclComm_t, ncclResult_t* e) { if (e) *e = ncclSuccess; return ncclSuccess; }
ncclResult_t ncclGroupStart(void) { return ncclSuccess; }
ncclResult_t ncclGroupEnd(void) { return ncclSuccess; }
const char* ncclGetErrorString(ncclResult_t r) {
  switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "loopback_rccl: HIP error";
    case ncclSystemError: return "loopback_rccl: system error or a peer did not arrive";
    case ncclInvalidArgument: return "loopback_rccl: invalid argument";
    case ncclInvalidUsage: return "loopback_rccl: invalid usage (one process per rank only)";
    default: return "loopback_rccl: error";
  }
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t dt, ncclComm_t c, hipStream_t st) {
  const size_t bytes = count * dtype_bytes(dt);
  if (!c || bytes > STAGING_BYTES) return ncclInvalidArgument;
  if (hipMemcpyAsync(c->staging, send, bytes, hipMemcpyDeviceToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
    return ncclUnhandledCudaError;
  if (!barrier(c)) return ncclSystemError;
  for (int p = 0; p < c->n; ++p)
    if (hipMemcpyAsync(static_cast<char*>(recv) + (size_t)p * bytes, c->peer[p], bytes, hipMemcpyDeviceToDevice, st) != hipSuccess)
      return ncclUnhandledCudaError;
  if (hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;
  return barrier(c) ? ncclSuccess : ncclSystemError;   // my staging buffer is free again
}

ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t c, hipStream_t st) {
  if (!c || count != 1 || dt != ncclUint64 || op != ncclSum) return ncclInvalidArgument;   // all group.h asks for
  uint64_t v = 0;
  if (hipMemcpyAsync(&v, send, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
    return ncclUnhandledCudaError;
  c->sh->values[c->rank] = v;
  if (!barrier(c)) return ncclSystemError;
  uint64_t sum = 0;
  for (int p = 0; p < c->n; ++p) sum += c->sh->values[p];
  if (!barrier(c)) return ncclSystemError;
  if (hipMemcpyAsync(recv, &sum, 8, hipMemcpyHostToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
    return ncclUnhandledCudaError;
  return ncclSuccess;
}

}  // extern "C"
