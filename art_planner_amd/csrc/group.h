// Device groups: the multi-GPU form of the sampling + validity path behind the C ABI (include/artp_c.h "multi-GPU").
// Included at the end of artp_capi.hip (same translation unit: it uses artp_ctx's stream members).
//
// Design: replicated maps, sharded sample-index ranges, two exchange steps.  A member = one GPU = one
// artp_ctx + one RCCL communicator + one side stream.  Per step and member, on the context's lane-0 stream: fused
// sample + validate of the member's shard, one validity bit per candidate; on the side stream (context lane 3):
// ncclAllGather of the W bitmaps, then the re-materialisation of every rank's accepted states.  Buffers alternate by
// step parity and are guarded by events, so the host only enqueues.  librccl is bound with dlopen at group creation
// (a Python process that already holds torch's copy gets that one), so libartp.so carries no link dependency on it.
#pragma once

#include <dlfcn.h>
// librccl is bound at run time (dlopen); its HEADER is only needed for the types.  A ROCm installation without the RCCL
// development files still builds libartp.so: the few declarations used here are spelled out below (rccl.h 2.x ABI) and
// artp_group_create* reports ARTP_ERR_COMM when no librccl can be loaded.
#if !defined(ARTP_NO_RCCL_HEADER) && defined(__has_include) && __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
extern "C" {
typedef struct ncclComm* ncclComm_t;
#define NCCL_UNIQUE_ID_BYTES 128
typedef struct { char internal[NCCL_UNIQUE_ID_BYTES]; } ncclUniqueId;
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4,
               ncclInvalidUsage = 5, ncclRemoteError = 6, ncclInProgress = 7, ncclNumResults = 8 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1, ncclInt32 = 2, ncclInt = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5,
               ncclFloat16 = 6, ncclHalf = 6, ncclFloat32 = 7, ncclFloat = 7, ncclFloat64 = 8, ncclDouble = 8 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3, ncclAvg = 4 } ncclRedOp_t;
ncclResult_t ncclGetUniqueId(ncclUniqueId* uniqueId);
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId commId, int rank);
ncclResult_t ncclCommInitAll(ncclComm_t* comm, int ndev, const int* devlist);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
ncclResult_t ncclCommAbort(ncclComm_t comm);
ncclResult_t ncclCommGetAsyncError(ncclComm_t comm, ncclResult_t* asyncError);
ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm,
                           hipStream_t stream);
ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op,
                           ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclGroupStart(void);
ncclResult_t ncclGroupEnd(void);
const char* ncclGetErrorString(ncclResult_t result);
}
#endif
#include <atomic>

#include <condition_variable>
#include <functional>
#include <thread>

namespace artp_group_detail {

struct RcclApi {
  void* handle = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommAbort) CommAbort = nullptr;
  decltype(&ncclCommGetAsyncError) CommGetAsyncError = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  std::string error;
};

// One binding per process.  Candidates: $ARTP_RCCL_LIB (if set it is THE choice: a failure to load it is an error), a copy
// the process already holds (torch ships its own librccl.so), then the ROCm installation's.
inline RcclApi& rccl_storage() {
  static RcclApi api;
  return api;
}
inline RcclApi* rccl_api() {
  RcclApi& api = rccl_storage();
  static std::once_flag once;
  std::call_once(once, [&api] {
    std::vector<std::string> names;
    void* h = nullptr;
    static const char kEnv[] = "ARTP_RCCL_LIB";   // (one copy of the name in the binary: the message is built from it)
    if (const char* e = std::getenv(kEnv)) {   // an explicit choice goes first, also in front of a loaded copy
      h = dlopen(e, RTLD_NOW | RTLD_LOCAL);
      if (!h) {
        const char* de = dlerror();
        api.error = std::string(e) + " (named by the environment variable " + kEnv + "): " + (de ? de : "dlopen failed");
        return;
      }
    }
    for (const char* n : {"librccl.so", "librccl.so.1"}) {
      if (h) break;
      h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    }
    for (const char* n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) names.push_back(n);
    for (size_t k = 0; !h && k < names.size(); ++k) h = dlopen(names[k].c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!h) {
      const char* e = dlerror();
      api.error = std::string("librccl not found (set ARTP_RCCL_LIB): ") + (e ? e : "");
      return;
    }
    api.handle = h;
    bool ok = true;
#define ARTP_RCCL_SYM(field, name)                                         \
  api.field = reinterpret_cast<decltype(api.field)>(dlsym(h, #name));      \
  if (!api.field) { ok = false; api.error += std::string(" missing ") + #name; }
    ARTP_RCCL_SYM(GetUniqueId, ncclGetUniqueId)
    ARTP_RCCL_SYM(CommInitRank, ncclCommInitRank)
    ARTP_RCCL_SYM(CommInitAll, ncclCommInitAll)
    ARTP_RCCL_SYM(CommDestroy, ncclCommDestroy)
    ARTP_RCCL_SYM(CommAbort, ncclCommAbort)
    ARTP_RCCL_SYM(CommGetAsyncError, ncclCommGetAsyncError)
    ARTP_RCCL_SYM(AllGather, ncclAllGather)
    ARTP_RCCL_SYM(AllReduce, ncclAllReduce)
    ARTP_RCCL_SYM(GroupStart, ncclGroupStart)
    ARTP_RCCL_SYM(GroupEnd, ncclGroupEnd)
    ARTP_RCCL_SYM(GetErrorString, ncclGetErrorString)
#undef ARTP_RCCL_SYM
    if (!ok) api.handle = nullptr;
  });
  return api.handle ? &api : nullptr;
}

inline std::string rccl_error() {
  (void)rccl_api();
  return "RCCL unavailable:" + rccl_storage().error;
}

// a host thread per member of a single-process group: the launches of a step are issued on all GPUs at once
struct Worker {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::function<int()> job;
  bool has_job = false, done = false, quit = false;
  int result = 0;

  void start() {
    th = std::thread([this] {
      std::unique_lock<std::mutex> lk(mu);
      for (;;) {
        cv.wait(lk, [this] { return has_job || quit; });
        if (quit) return;
        std::function<int()> j = std::move(job);
        has_job = false;
        lk.unlock();
        const int r = j();
        lk.lock();
        result = r;
        done = true;
        cv.notify_all();
      }
    });
  }
  void post(std::function<int()> j) {
    std::lock_guard<std::mutex> lk(mu);
    job = std::move(j);
    has_job = true;
    done = false;
    cv.notify_all();
  }
  int wait() {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [this] { return done; });
    return result;
  }
  void stop() {
    {
      std::lock_guard<std::mutex> lk(mu);
      quit = true;
      cv.notify_all();
    }
    if (th.joinable()) th.join();
  }
};

struct Member {
  artp_ctx* ctx = nullptr;
  int device = 0, rank = 0;
  ncclComm_t comm = nullptr;
  hipStream_t comm_stream = nullptr;
  hipEvent_t ready[2] = {nullptr, nullptr};   // bits[b] packed (compute stream)
  hipEvent_t done[2] = {nullptr, nullptr};    // exchange of buffer b finished (side stream)
  hipEvent_t pushed[2] = {nullptr, nullptr};  // peer-copy transport: this member's bitmap is in every gathered[b]
  hipEvent_t edge_ready = nullptr, edge_done = nullptr, edge_pushed = nullptr;
  double* se3 = nullptr;
  uint8_t* valid = nullptr;
  uint64_t* bits[2] = {nullptr, nullptr};
  uint64_t* gathered[2] = {nullptr, nullptr};
  double* mat_states[2] = {nullptr, nullptr};
  uint64_t* mat_counts[2] = {nullptr, nullptr};
  uint64_t* ones = nullptr;  // ranks_seen scratch: [0] in, [1] out
  uint32_t* rec_local = nullptr;
  uint64_t* rec_count = nullptr;
  uint32_t* rec_gathered = nullptr;
  uint64_t* rec_counts = nullptr;
  size_t edge_cap = 0;
  Worker* worker = nullptr;
  std::string error;
};

}  // namespace artp_group_detail

struct artp_group {
  int world = 0, transport = ARTP_GROUP_RCCL;
  bool single_process = true;
  std::vector<artp_group_detail::Member> m;
  artp_group_detail::RcclApi* rccl = nullptr;
  uint64_t seed = 0;
  size_t batch = 0, words = 0, mat_cap = 0, prefix = 0;
  bool configured = false;
  std::atomic<bool> aborted{false};   // set by artp_group_abort without g->mu: a synchronize waiting for a peer sees it
  std::string last_error;
  std::mutex mu;  // group calls are serialised
};

namespace artp_group_detail {

#define GRP_HIP(mem, expr)                                                              \
  do {                                                                                  \
    hipError_t _e = (expr);                                                             \
    if (_e != hipSuccess) {                                                             \
      (mem).error = std::string(#expr) + ": " + hipGetErrorString(_e);                  \
      return (int)ARTP_ERR_HIP; \
    }                                                                                   \
  } while (0)
#define GRP_NCCL(g, mem, expr)                                                          \
  do {                                                                                  \
    ncclResult_t _r = (expr);                                                           \
    if (_r != ncclSuccess) {                                                            \
      (mem).error = std::string(#expr) + ": " + (g)->rccl->GetErrorString(_r);          \
      return (int)ARTP_ERR_COMM; \
    }                                                                                   \
  } while (0)
#define GRP_CTX(mem, expr)                                                              \
  do {                                                                                  \
    int _rc = (expr);                                                                   \
    if (_rc != ARTP_OK) {                                                               \
      (mem).error = std::string(#expr) + ": " + artp_status_string(_rc) + " (" +        \
                    artp_last_error((mem).ctx) + ")";                                   \
      return _rc;                                                                       \
    }                                                                                   \
  } while (0)

// runs fn(member index) for every local member -- on the members' worker threads when there are several -- and
// returns the first failure (text in g->last_error)
inline int for_each_member(artp_group* g, const std::function<int(int)>& fn) {
  const int n = (int)g->m.size();
  int rc = ARTP_OK;
  if (n == 1 || !g->m[0].worker) {
    for (int i = 0; i < n; ++i) {
      const int r = fn(i);
      if (r != ARTP_OK && rc == ARTP_OK) {
        rc = r;
        g->last_error = "member " + std::to_string(i) + " (rank " + std::to_string(g->m[i].rank) + "): " + g->m[i].error;
      }
    }
    return rc;
  }
  for (int i = 0; i < n; ++i) g->m[i].worker->post([&fn, i] { return fn(i); });
  for (int i = 0; i < n; ++i) {
    const int r = g->m[i].worker->wait();
    if (r != ARTP_OK && rc == ARTP_OK) {
      rc = r;
      g->last_error = "member " + std::to_string(i) + " (rank " + std::to_string(g->m[i].rank) + "): " + g->m[i].error;
    }
  }
  return rc;
}

// the context's lane-0 stream (what artp_sample_and_validate_dev runs on); lane 3 carries the side stream
inline hipStream_t compute_stream(Member& mem) {
  (void)artp_set_lane(mem.ctx, 0);
  return mem.ctx->stream;
}

inline void free_exchange_buffers(Member& mem) {
  (void)hipSetDevice(mem.device);
  auto fr = [](auto*& p) {
    if (p) (void)hipFree(p);
    p = nullptr;
  };
  fr(mem.se3);
  fr(mem.valid);
  for (int b = 0; b < 2; ++b) {
    fr(mem.bits[b]);
    fr(mem.gathered[b]);
    fr(mem.mat_states[b]);
    fr(mem.mat_counts[b]);
  }
}

inline void free_edge_buffers(Member& mem) {
  (void)hipSetDevice(mem.device);
  auto fr = [](auto*& p) {
    if (p) (void)hipFree(p);
    p = nullptr;
  };
  fr(mem.rec_local);
  fr(mem.rec_count);
  fr(mem.rec_gathered);
  fr(mem.rec_counts);
  mem.edge_cap = 0;
}

inline int init_member_common(artp_group* g, Member& mem, const artp_params* params) {
  GRP_CTX(mem, artp_create(mem.device, params, &mem.ctx));
  GRP_HIP(mem, hipSetDevice(mem.device));
  GRP_HIP(mem, hipStreamCreateWithFlags(&mem.comm_stream, hipStreamNonBlocking));
  // the side stream's launches (re-materialisation) get a lane of their own: scratch, counters, stream
  GRP_CTX(mem, artp_set_lane(mem.ctx, 3));
  GRP_CTX(mem, artp_set_stream(mem.ctx, mem.comm_stream));
  GRP_CTX(mem, artp_set_lane(mem.ctx, 0));
  hipEvent_t* evs[] = {&mem.ready[0], &mem.ready[1], &mem.done[0], &mem.done[1], &mem.pushed[0], &mem.pushed[1],
                       &mem.edge_ready, &mem.edge_done, &mem.edge_pushed};
  for (hipEvent_t* e : evs) {
    GRP_HIP(mem, hipEventCreateWithFlags(e, hipEventDisableTiming));
    GRP_HIP(mem, hipEventRecord(*e, mem.comm_stream));  // "free" from the start
  }
  GRP_HIP(mem, hipMalloc(reinterpret_cast<void**>(&mem.ones), 2 * sizeof(uint64_t)));
  return ARTP_OK;
}

inline void destroy_member(artp_group* g, Member& mem) {
  (void)hipSetDevice(mem.device);
  if (mem.comm_stream && !g->aborted) (void)hipStreamSynchronize(mem.comm_stream);
  if (mem.ctx && !g->aborted) (void)artp_synchronize(mem.ctx);
  if (mem.comm && g->rccl) {
    if (g->aborted) (void)g->rccl->CommAbort(mem.comm);
    else (void)g->rccl->CommDestroy(mem.comm);
    mem.comm = nullptr;
  }
  free_exchange_buffers(mem);
  free_edge_buffers(mem);
  if (mem.ones) (void)hipFree(mem.ones);
  hipEvent_t evs[] = {mem.ready[0], mem.ready[1], mem.done[0], mem.done[1], mem.pushed[0], mem.pushed[1],
                      mem.edge_ready, mem.edge_done, mem.edge_pushed};
  for (hipEvent_t e : evs)
    if (e) (void)hipEventDestroy(e);
  if (mem.ctx) {
    // the context does not own the side stream set on lane 3: hand it its own stream back before it is destroyed
    (void)artp_set_lane(mem.ctx, 3);
    (void)artp_use_own_stream(mem.ctx);
    (void)artp_set_lane(mem.ctx, 0);
    artp_destroy(mem.ctx);
  }
  if (mem.comm_stream) (void)hipStreamDestroy(mem.comm_stream);
}

inline int member_by_rank(const artp_group* g, int rank) {
  for (size_t i = 0; i < g->m.size(); ++i)
    if (g->m[i].rank == rank) return (int)i;
  return -1;
}

// all-gather of `bytes` per rank from src (this member) into dst (W blocks) of every member, on the side stream.
// RCCL: one collective.  Peer copy: this member pushes its block into every member's dst; the caller then makes
// every member's side stream wait for all `pushed` events (second phase, after a host barrier).
inline int gather_block(artp_group* g, Member& mem, const void* src, void* const* dst_of_member, size_t bytes,
                        hipEvent_t pushed) {
  if (g->transport == ARTP_GROUP_RCCL) {
    GRP_NCCL(g, mem, g->rccl->AllGather(src, dst_of_member[member_by_rank(g, mem.rank)], bytes, ncclUint8, mem.comm,
                                        mem.comm_stream));
    return ARTP_OK;
  }
  for (size_t j = 0; j < g->m.size(); ++j) {
    char* dst = static_cast<char*>(dst_of_member[j]) + (size_t)mem.rank * bytes;
    GRP_HIP(mem, hipMemcpyPeerAsync(dst, g->m[j].device, src, mem.device, bytes, mem.comm_stream));
  }
  GRP_HIP(mem, hipEventRecord(pushed, mem.comm_stream));
  return ARTP_OK;
}

}  // namespace artp_group_detail

extern "C" {

uint64_t artp_shard_first_index(uint64_t step, int rank, int world, uint64_t batch) {
  return (step * (uint64_t)world + (uint64_t)rank) * batch;
}

const char* artp_group_last_error(const artp_group* g) { return g ? g->last_error.c_str() : ""; }
int artp_group_world_size(const artp_group* g) { return g ? g->world : 0; }
int artp_group_local_count(const artp_group* g) { return g ? (int)g->m.size() : 0; }
int artp_group_rank(const artp_group* g, int local) {
  return (g && local >= 0 && local < (int)g->m.size()) ? g->m[local].rank : -1;
}
artp_ctx* artp_group_ctx(artp_group* g, int local) {
  return (g && local >= 0 && local < (int)g->m.size()) ? g->m[local].ctx : nullptr;
}

void artp_group_destroy(artp_group* g) {
  using namespace artp_group_detail;
  if (!g) return;
  for (auto& mem : g->m) {
    if (mem.worker) {
      mem.worker->stop();
      delete mem.worker;
      mem.worker = nullptr;
    }
  }
  for (auto& mem : g->m) destroy_member(g, mem);
  delete g;
}

int artp_group_create(const int* devices, int n, const artp_params* params, int transport, artp_group** out) {
  using namespace artp_group_detail;
  if (!devices || n < 1 || n > 16 || !params || !out ||
      (transport != ARTP_GROUP_RCCL && transport != ARTP_GROUP_PEER_COPY))
    return ARTP_ERR_INVALID_ARG;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return ARTP_ERR_NO_DEVICE;
  for (int i = 0; i < n; ++i) {
    if (devices[i] < 0 || devices[i] >= count) return ARTP_ERR_NO_DEVICE;
    if (transport == ARTP_GROUP_RCCL)  // RCCL refuses one GPU twice in a communicator
      for (int j = 0; j < i; ++j)
        if (devices[j] == devices[i]) return ARTP_ERR_INVALID_ARG;
  }
  artp_group* g = new artp_group();
  g->world = n;
  g->transport = transport;
  g->single_process = true;
  g->m.resize(n);
  int rc = ARTP_OK;
  for (int i = 0; i < n && rc == ARTP_OK; ++i) {
    g->m[i].device = devices[i];
    g->m[i].rank = i;
    rc = init_member_common(g, g->m[i], params);
    if (rc != ARTP_OK) g->last_error = "member " + std::to_string(i) + ": " + g->m[i].error;
  }
  if (rc == ARTP_OK && transport == ARTP_GROUP_RCCL) {
    g->rccl = rccl_api();
    if (!g->rccl) {
      rc = ARTP_ERR_COMM;
      g->last_error = rccl_error();
    } else {
      std::vector<ncclComm_t> comms(n);
      ncclResult_t r = g->rccl->CommInitAll(comms.data(), n, devices);
      if (r != ncclSuccess) {
        rc = ARTP_ERR_COMM;
        g->last_error = std::string("ncclCommInitAll: ") + g->rccl->GetErrorString(r);
      } else {
        for (int i = 0; i < n; ++i) g->m[i].comm = comms[i];
      }
    }
  }
  if (rc == ARTP_OK && transport == ARTP_GROUP_PEER_COPY) {
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) {
        if (devices[i] == devices[j]) continue;
        (void)hipSetDevice(devices[i]);
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, devices[i], devices[j]) == hipSuccess && can)
          (void)hipDeviceEnablePeerAccess(devices[j], 0);  // already enabled is fine; without it the copies are staged
        (void)hipGetLastError();
      }
  }
  if (rc == ARTP_OK && n > 1)
    for (int i = 0; i < n; ++i) {
      g->m[i].worker = new Worker();
      g->m[i].worker->start();
    }
  if (rc != ARTP_OK) {
    std::fprintf(stderr, "artp_group_create: %s\n", g->last_error.c_str());
    artp_group_destroy(g);
    return rc;
  }
  *out = g;
  return ARTP_OK;
}

int artp_group_unique_id(uint8_t id[ARTP_GROUP_ID_BYTES]) {
  using namespace artp_group_detail;
  static_assert(sizeof(ncclUniqueId) == ARTP_GROUP_ID_BYTES, "ncclUniqueId size");
  if (!id) return ARTP_ERR_INVALID_ARG;
  RcclApi* api = rccl_api();
  if (!api) return ARTP_ERR_COMM;
  ncclUniqueId u;
  if (api->GetUniqueId(&u) != ncclSuccess) return ARTP_ERR_COMM;
  std::memcpy(id, &u, sizeof(u));
  return ARTP_OK;
}

int artp_group_create_rank(int device, int rank, int world, const uint8_t id[ARTP_GROUP_ID_BYTES],
                           const artp_params* params, artp_group** out) {
  using namespace artp_group_detail;
  if (!id || !params || !out || world < 1 || world > 16 || rank < 0 || rank >= world) return ARTP_ERR_INVALID_ARG;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return ARTP_ERR_NO_DEVICE;
  artp_group* g = new artp_group();
  g->world = world;
  g->transport = ARTP_GROUP_RCCL;
  g->single_process = false;
  g->m.resize(1);
  g->m[0].device = device;
  g->m[0].rank = rank;
  int rc = init_member_common(g, g->m[0], params);
  if (rc != ARTP_OK) g->last_error = g->m[0].error;
  if (rc == ARTP_OK) {
    g->rccl = rccl_api();
    if (!g->rccl) {
      rc = ARTP_ERR_COMM;
      g->last_error = rccl_error();
    } else {
      ncclUniqueId u;
      std::memcpy(&u, id, sizeof(u));
      (void)hipSetDevice(device);
      ncclResult_t r = g->rccl->CommInitRank(&g->m[0].comm, world, u, rank);
      if (r != ncclSuccess) {
        rc = ARTP_ERR_COMM;
        g->last_error = std::string("ncclCommInitRank: ") + g->rccl->GetErrorString(r);
      }
    }
  }
  if (rc != ARTP_OK) {
    std::fprintf(stderr, "artp_group_create_rank: %s\n", g->last_error.c_str());
    artp_group_destroy(g);
    return rc;
  }
  *out = g;
  return ARTP_OK;
}

int artp_group_synchronize(artp_group* g, int timeout_ms) {
  using namespace artp_group_detail;
  if (!g) return ARTP_ERR_INVALID_ARG;
  // The streams to wait for are read under the lock; the POLL runs without it, so that artp_group_abort (and any other
  // call) gets through while this thread waits for a peer that never arrives (synchronize(-1) could not be
  // aborted).  Member streams live as long as the group; destroying the group while a thread waits on it is the caller's
  // bug, as with any handle.
  struct Wait { hipStream_t stream; ncclComm_t comm; int device, rank; size_t member; const char* what; };
  std::vector<Wait> waits;
  {
    std::lock_guard<std::mutex> lock(g->mu);
    for (size_t i = 0; i < g->m.size(); ++i) {
      Member& mem = g->m[i];
      waits.push_back({compute_stream(mem), mem.comm, mem.device, mem.rank, i, "compute stream"});
      waits.push_back({mem.comm_stream, mem.comm, mem.device, mem.rank, i, "exchange stream"});
    }
  }
  auto fail = [&](int code, const std::string& msg) {
    std::lock_guard<std::mutex> lock(g->mu);
    g->last_error = msg;
    return code;
  };
  const auto t0 = std::chrono::steady_clock::now();
  for (const Wait& w : waits) {
    if (hipSetDevice(w.device) != hipSuccess)
      return fail(ARTP_ERR_HIP, "member " + std::to_string(w.member) + ": hipSetDevice(" + std::to_string(w.device) + ") failed");
    for (;;) {
      if (g->aborted.load(std::memory_order_acquire)) return fail(ARTP_ERR_COMM, "group was aborted");
      const hipError_t q = hipStreamQuery(w.stream);
      if (q == hipSuccess) break;
      if (q != hipErrorNotReady)
        return fail(ARTP_ERR_HIP, "member " + std::to_string(w.member) + " " + w.what + ": " + hipGetErrorString(q));
      if (w.comm && g->rccl) {
        // the communicator is looked at under the group's lock, re-read from the member: artp_group_abort destroys it
        // (CommAbort) and clears the member's handle under the same lock, so a handle copied at entry is never used
        // after the abort (check-then-use on a destroyed communicator)
        ncclResult_t ar = ncclSuccess;
        bool have = false;
        {
          std::lock_guard<std::mutex> lock(g->mu);
          ncclComm_t live = g->aborted.load(std::memory_order_acquire) ? nullptr : g->m[w.member].comm;
          if (live) have = g->rccl->CommGetAsyncError(live, &ar) == ncclSuccess;
        }
        if (have && ar != ncclSuccess && ar != ncclInProgress)
          return fail(ARTP_ERR_COMM, "member " + std::to_string(w.member) + ": RCCL async error: " + g->rccl->GetErrorString(ar));
      }
      if (timeout_ms >= 0) {
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (ms > (double)timeout_ms)
          return fail(ARTP_ERR_TIMEOUT, "member " + std::to_string(w.member) + " (rank " + std::to_string(w.rank) + "): " + w.what +
                                            " still busy after " + std::to_string(timeout_ms) + " ms");
      }
      std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
  }
  return ARTP_OK;
}

int artp_group_abort(artp_group* g) {
  if (!g) return ARTP_ERR_INVALID_ARG;
  g->aborted.store(true, std::memory_order_release);   // waiters in artp_group_synchronize leave at their next poll
  std::lock_guard<std::mutex> lock(g->mu);
  for (auto& mem : g->m)
    if (mem.comm && g->rccl) {
      (void)g->rccl->CommAbort(mem.comm);
      mem.comm = nullptr;
    }
  return ARTP_OK;
}

int artp_group_ranks_seen(artp_group* g, int* ranks_seen) {
  using namespace artp_group_detail;
  if (!g || !ranks_seen) return ARTP_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lock(g->mu);
  if (g->aborted) return ARTP_ERR_COMM;
  *ranks_seen = 0;
  if (g->transport == ARTP_GROUP_PEER_COPY) {  // every member adds one to member 0's counter
    Member& m0 = g->m[0];
    if (hipSetDevice(m0.device) != hipSuccess || hipMemset(m0.ones, 0, 2 * sizeof(uint64_t)) != hipSuccess)
      return ARTP_ERR_HIP;
    uint64_t total = 0;
    for (auto& mem : g->m) {
      const uint64_t one = 1;
      uint64_t back = 0;
      if (hipSetDevice(mem.device) != hipSuccess ||
          hipMemcpy(mem.ones + 1, &one, sizeof(one), hipMemcpyHostToDevice) != hipSuccess ||
          hipMemcpyPeer(m0.ones, m0.device, mem.ones + 1, mem.device, sizeof(one)) != hipSuccess ||
          hipMemcpy(&back, m0.ones, sizeof(back), hipMemcpyDeviceToHost) != hipSuccess)
        return ARTP_ERR_HIP;
      total += back;
    }
    *ranks_seen = (int)total;
    return ARTP_OK;
  }
  int rc = for_each_member(g, [g](int i) -> int {
    Member& mem = g->m[i];
    const uint64_t init[2] = {1, 0};
    GRP_HIP(mem, hipSetDevice(mem.device));
    GRP_HIP(mem, hipMemcpyAsync(mem.ones, init, sizeof(init), hipMemcpyHostToDevice, mem.comm_stream));
    GRP_HIP(mem, hipStreamSynchronize(mem.comm_stream));
    GRP_NCCL(g, mem, g->rccl->AllReduce(mem.ones, mem.ones + 1, 1, ncclUint64, ncclSum, mem.comm, mem.comm_stream));
    return (int)ARTP_OK;
  });
  if (rc != ARTP_OK) return rc;
  Member& m0 = g->m[0];
  uint64_t seen = 0;
  if (hipSetDevice(m0.device) != hipSuccess) return ARTP_ERR_HIP;
  // bounded wait: a peer that never joins must not hang the caller
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    const hipError_t q = hipStreamQuery(m0.comm_stream);
    if (q == hipSuccess) break;
    if (q != hipErrorNotReady) return ARTP_ERR_HIP;
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 60.0) {
      g->last_error = "artp_group_ranks_seen: the all-reduce did not finish within 60 s";
      return ARTP_ERR_TIMEOUT;
    }
    std::this_thread::sleep_for(std::chrono::microseconds(100));
  }
  if (hipMemcpy(&seen, m0.ones + 1, sizeof(seen), hipMemcpyDeviceToHost) != hipSuccess) return ARTP_ERR_HIP;
  *ranks_seen = (int)seen;
  return ARTP_OK;
}

int artp_group_configure(artp_group* g, uint64_t seed, size_t batch, size_t materialise_cap, size_t prefix) {
  using namespace artp_group_detail;
  if (!g || batch == 0 || batch >= (1ull << 32)) return ARTP_ERR_INVALID_ARG;
  if (prefix == 0 || prefix > batch) prefix = batch;
  if (materialise_cap > prefix) materialise_cap = prefix;
  {
    // the previous configuration's work must have drained before its buffers go; bounded: a peer that never arrived must
    // not hang a re-configuration for ever ($ARTP_GROUP_CONFIGURE_TIMEOUT_MS, default one minute)
    // (-1 = wait for ever; anything that does not parse as an integer >= -1 keeps the default)
    int wait_ms = 60000;
    if (const char* ev = std::getenv("ARTP_GROUP_CONFIGURE_TIMEOUT_MS")) {
      char* end = nullptr;
      const long v = std::strtol(ev, &end, 10);
      if (end != ev && *end == '\0' && v >= -1 && v <= 86400000L) wait_ms = (int)v;
    }
    const int rc = artp_group_synchronize(g, wait_ms);
    if (rc != ARTP_OK) return rc;
  }
  std::lock_guard<std::mutex> lock(g->mu);
  g->configured = false;
  const size_t words = (batch + 63) / 64;
  const int W = g->world;
  int rc = for_each_member(g, [&](int i) -> int {
    Member& mem = g->m[i];
    free_exchange_buffers(mem);
    GRP_HIP(mem, hipSetDevice(mem.device));
    GRP_HIP(mem, hipMalloc(reinterpret_cast<void**>(&mem.se3), batch * 7 * sizeof(double)));
    GRP_HIP(mem, hipMalloc(reinterpret_cast<void**>(&mem.valid), batch));
    for (int b = 0; b < 2; ++b) {
      GRP_HIP(mem, hipMalloc(reinterpret_cast<void**>(&mem.bits[b]), words * sizeof(uint64_t)));
      GRP_HIP(mem, hipMalloc(reinterpret_cast<void**>(&mem.gathered[b]), (size_t)W * words * sizeof(uint64_t)));
      GRP_HIP(mem, hipMemset(mem.gathered[b], 0, (size_t)W * words * sizeof(uint64_t)));
      GRP_HIP(mem, hipMalloc(reinterpret_cast<void**>(&mem.mat_counts[b]), (size_t)W * sizeof(uint64_t)));
      GRP_HIP(mem, hipMemset(mem.mat_counts[b], 0, (size_t)W * sizeof(uint64_t)));
      if (materialise_cap)
        GRP_HIP(mem, hipMalloc(reinterpret_cast<void**>(&mem.mat_states[b]),
                               (size_t)W * materialise_cap * 7 * sizeof(double)));
    }
    return (int)ARTP_OK;
  });
  if (rc != ARTP_OK) return rc;
  g->seed = seed;
  g->batch = batch;
  g->words = words;
  g->mat_cap = materialise_cap;
  g->prefix = prefix;
  g->configured = true;
  return ARTP_OK;
}

int artp_group_sample_and_validate_step(artp_group* g, uint64_t step) {
  using namespace artp_group_detail;
  if (!g) return ARTP_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lock(g->mu);
  if (!g->configured || g->aborted) {
    g->last_error = g->aborted ? "group was aborted" : "artp_group_configure first";
    return ARTP_ERR_INVALID_ARG;
  }
  const int b = (int)(step & 1);
  const int W = g->world;
  const size_t S = g->batch, words = g->words;
  uint64_t bases[16];
  for (int r = 0; r < W; ++r) bases[r] = artp_shard_first_index(step, r, W, S);
  std::vector<void*> dst(g->m.size());
  for (size_t j = 0; j < g->m.size(); ++j) dst[j] = g->m[j].gathered[b];

  auto materialise = [&](Member& mem) -> int {
    GRP_CTX(mem, artp_set_lane(mem.ctx, 3));
    int rcm = ARTP_OK;
    if (g->mat_cap)
      rcm = artp_materialise_from_bits_dev(mem.ctx, g->seed, mem.gathered[b], W, words, g->prefix, bases, g->mat_cap,
                                           mem.mat_states[b], mem.mat_counts[b]);
    (void)artp_set_lane(mem.ctx, 0);
    if (rcm != ARTP_OK) {
      mem.error = std::string("artp_materialise_from_bits_dev: ") + artp_status_string(rcm) + " (" +
                  artp_last_error(mem.ctx) + ")";
      return rcm;
    }
    GRP_HIP(mem, hipEventRecord(mem.done[b], mem.comm_stream));
    return (int)ARTP_OK;
  };

  int rc = for_each_member(g, [&](int i) -> int {
    Member& mem = g->m[i];
    GRP_HIP(mem, hipSetDevice(mem.device));
    hipStream_t cs = compute_stream(mem);
    GRP_CTX(mem, artp_sample_and_validate_dev(mem.ctx, g->seed, bases[mem.rank], S, mem.se3, mem.valid, nullptr));
    GRP_HIP(mem, hipStreamWaitEvent(cs, mem.done[b], 0));  // buffer b's previous exchange is over
    GRP_CTX(mem, artp_pack_valid_bits_dev(mem.ctx, mem.valid, S, mem.bits[b]));
    GRP_HIP(mem, hipEventRecord(mem.ready[b], cs));
    GRP_HIP(mem, hipStreamWaitEvent(mem.comm_stream, mem.ready[b], 0));
    if (g->transport == ARTP_GROUP_PEER_COPY)  // the receivers' previous use of gathered[b] must be over too
      for (auto& other : g->m) GRP_HIP(mem, hipStreamWaitEvent(mem.comm_stream, other.done[b], 0));
    int rg = gather_block(g, mem, mem.bits[b], dst.data(), words * sizeof(uint64_t), mem.pushed[b]);
    if (rg != ARTP_OK) return rg;
    if (g->transport == ARTP_GROUP_RCCL) return materialise(mem);
    return (int)ARTP_OK;
  });
  if (rc != ARTP_OK || g->transport == ARTP_GROUP_RCCL) return rc;
  // peer copy, second phase: every `pushed` event of this step has been recorded by now
  return for_each_member(g, [&](int i) -> int {
    Member& mem = g->m[i];
    GRP_HIP(mem, hipSetDevice(mem.device));
    for (auto& other : g->m) GRP_HIP(mem, hipStreamWaitEvent(mem.comm_stream, other.pushed[b], 0));
    return materialise(mem);
  });
}

int artp_group_step_buffers(artp_group* g, int local, uint64_t step, const double** se3, const uint8_t** valid,
                            const uint64_t** bits, const double** states, const uint64_t** counts) {
  if (!g || local < 0) return ARTP_ERR_INVALID_ARG;
  // The pointers stay valid until the next artp_group_configure.  Buffers alternate by step parity: step k + 2 overwrites
  // what step k produced as soon as the group's OWN work of step k is done -- a consumer that reads them asynchronously on
  // its own stream must have finished (or be ordered in front of the enqueue of step k + 2) by then.
  std::lock_guard<std::mutex> lock(g->mu);
  if (local >= (int)g->m.size() || !g->configured) return ARTP_ERR_INVALID_ARG;
  const artp_group_detail::Member& mem = g->m[local];
  const int b = (int)(step & 1);
  if (se3) *se3 = mem.se3;
  if (valid) *valid = mem.valid;
  if (bits) *bits = mem.gathered[b];
  if (states) *states = mem.mat_states[b];
  if (counts) *counts = mem.mat_counts[b];
  return ARTP_OK;
}

int artp_group_exchange_edges(artp_group* g, const artp_group_edges* per_local, size_t cap) {
  using namespace artp_group_detail;
  if (!g || !per_local || cap == 0 || cap >= (1ull << 31)) return ARTP_ERR_INVALID_ARG;
  for (size_t i = 0; i < g->m.size(); ++i)
    if (per_local[i].n > cap || (per_local[i].n && (!per_local[i].valid || !per_local[i].edge_i ||
                                                    !per_local[i].edge_j || !per_local[i].cost)))
      return ARTP_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lock(g->mu);
  if (g->aborted) return ARTP_ERR_COMM;
  const int W = g->world;
  const size_t block = cap * 5 * sizeof(uint32_t);
  std::vector<void*> dst_rec(g->m.size()), dst_cnt(g->m.size());
  int rc = for_each_member(g, [&](int i) -> int {
    Member& mem = g->m[i];
    GRP_HIP(mem, hipSetDevice(mem.device));
    if (mem.edge_cap != cap) {
      GRP_HIP(mem, hipStreamSynchronize(mem.comm_stream));
      free_edge_buffers(mem);
      GRP_HIP(mem, hipMalloc(reinterpret_cast<void**>(&mem.rec_local), block));
      GRP_HIP(mem, hipMalloc(reinterpret_cast<void**>(&mem.rec_count), sizeof(uint64_t)));
      GRP_HIP(mem, hipMalloc(reinterpret_cast<void**>(&mem.rec_gathered), (size_t)W * block));
      GRP_HIP(mem, hipMalloc(reinterpret_cast<void**>(&mem.rec_counts), (size_t)W * sizeof(uint64_t)));
      GRP_HIP(mem, hipMemset(mem.rec_local, 0, block));
      mem.edge_cap = cap;
    }
    return (int)ARTP_OK;
  });
  if (rc != ARTP_OK) return rc;
  for (size_t j = 0; j < g->m.size(); ++j) {
    dst_rec[j] = g->m[j].rec_gathered;
    dst_cnt[j] = g->m[j].rec_counts;
  }
  rc = for_each_member(g, [&](int i) -> int {
    Member& mem = g->m[i];
    const artp_group_edges& e = per_local[i];
    GRP_HIP(mem, hipSetDevice(mem.device));
    hipStream_t cs = compute_stream(mem);
    GRP_HIP(mem, hipStreamWaitEvent(cs, mem.edge_done, 0));  // the previous exchange has read rec_local
    if (e.n)
      GRP_CTX(mem, artp_pack_edge_results_dev(mem.ctx, e.valid, e.edge_i, e.edge_j, e.cost, e.n, mem.rec_local,
                                              mem.rec_count));
    else
      GRP_HIP(mem, hipMemsetAsync(mem.rec_count, 0, sizeof(uint64_t), cs));
    GRP_HIP(mem, hipEventRecord(mem.edge_ready, cs));
    GRP_HIP(mem, hipStreamWaitEvent(mem.comm_stream, mem.edge_ready, 0));
    if (g->transport == ARTP_GROUP_RCCL) {
      GRP_NCCL(g, mem, g->rccl->GroupStart());
      int r1 = gather_block(g, mem, mem.rec_count, dst_cnt.data(), sizeof(uint64_t), nullptr);
      int r2 = r1 == ARTP_OK ? gather_block(g, mem, mem.rec_local, dst_rec.data(), block, nullptr) : r1;
      GRP_NCCL(g, mem, g->rccl->GroupEnd());
      if (r2 != ARTP_OK) return r2;
      GRP_HIP(mem, hipEventRecord(mem.edge_done, mem.comm_stream));
      return (int)ARTP_OK;
    }
    for (auto& other : g->m) GRP_HIP(mem, hipStreamWaitEvent(mem.comm_stream, other.edge_done, 0));
    int r1 = gather_block(g, mem, mem.rec_count, dst_cnt.data(), sizeof(uint64_t), mem.edge_pushed);
    if (r1 != ARTP_OK) return r1;
    return gather_block(g, mem, mem.rec_local, dst_rec.data(), block, mem.edge_pushed);
  });
  if (rc != ARTP_OK || g->transport == ARTP_GROUP_RCCL) return rc;
  return for_each_member(g, [&](int i) -> int {
    Member& mem = g->m[i];
    GRP_HIP(mem, hipSetDevice(mem.device));
    for (auto& other : g->m) GRP_HIP(mem, hipStreamWaitEvent(mem.comm_stream, other.edge_pushed, 0));
    GRP_HIP(mem, hipEventRecord(mem.edge_done, mem.comm_stream));
    return (int)ARTP_OK;
  });
}

int artp_group_edge_buffers(artp_group* g, int local, const uint32_t** records, const uint64_t** counts) {
  if (!g || local < 0) return ARTP_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lock(g->mu);   // artp_group_exchange_edges (re)allocates these under the same lock
  if (local >= (int)g->m.size()) return ARTP_ERR_INVALID_ARG;
  if (records) *records = g->m[local].rec_gathered;
  if (counts) *counts = g->m[local].rec_counts;
  return ARTP_OK;
}

}  // extern "C"
