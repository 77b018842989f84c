// art_planner::DeviceGroup -- every GPU of the node behind ONE object of the C++ host.
//
// The reference's node keeps one planner object and one planning thread (art_planner_ros/src/planner_ros.cpp:250-319);
// its sampling loop (prm_motion_cost.cpp:174-186, lazy_prm_star_min_update.cpp:552-554) is what the device group of the C
// ABI shards over the GPUs (include/artp_c.h "multi-GPU"): rank r samples and validates candidates
// [artp_shard_first_index(step, r, W, S), + S) of the (seed, index) stream, the W validity bitmaps are all-gathered over
// xGMI, and every member re-materialises every rank's accepted states.  This class is the RAII form of those calls for
// a C++ host: no interpreter, no torch; errors become exceptions that carry artp_group_last_error.
//
//   art_planner::DeviceGroup grp({0, 1, 2, 3, 4, 5, 6, 7}, params);           // one process, eight GPUs
//   for (int m = 0; m < grp.localCount(); ++m) install_map(grp.context(m));    // maps are replicated
//   grp.configure(seed, 1 << 22, 1 << 16);
//   for (uint64_t step = 0; planning; ++step) {
//     grp.step(step);                                    // asynchronous: the host only enqueues
//     if (step) consume(grp.stepBuffers(0, step - 1));   // W x accepted states of the previous step, on device 0
//   }
//   grp.synchronize(2000);                               // bounded: a peer that never arrives becomes an exception
#pragma once

#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "artp_c.h"

namespace art_planner {

class DeviceGroup {
 public:
  using UniqueId = std::array<uint8_t, ARTP_GROUP_ID_BYTES>;

  // What a step left on one member (device pointers on that member's GPU; see artp_group_step_buffers).
  struct StepBuffers {
    const double* se3 = nullptr;       // the member's own S candidates ...
    const uint8_t* valid = nullptr;    // ... and their labels
    const uint64_t* bits = nullptr;    // W gathered bitmaps, ceil(S / 64) words each
    const double* states = nullptr;    // W x materialise_cap x 7: every rank's accepted states
    const uint64_t* counts = nullptr;  // W accepted counts (among the prefix)
  };
  struct EdgeBuffers {
    const uint32_t* records = nullptr;  // W x cap x {u32 i, u32 j, f32 cost[3]}
    const uint64_t* counts = nullptr;   // W record counts
  };

  // all of `devices` in this process (the library runs one host worker thread per member)
  DeviceGroup(const std::vector<int>& devices, const artp_params& params, int transport = ARTP_GROUP_RCCL) {
    const int rc = artp_group_create(devices.data(), static_cast<int>(devices.size()), &params, transport, &g_);
    if (rc != ARTP_OK) throw std::runtime_error(std::string("artp_group_create: ") + artp_status_string(rc));
  }
  // one rank of a process-per-GPU job: rank 0 calls uniqueId(), the launcher hands the 128 bytes to every rank
  DeviceGroup(int device, int rank, int world, const UniqueId& id, const artp_params& params) {
    const int rc = artp_group_create_rank(device, rank, world, id.data(), &params, &g_);
    if (rc != ARTP_OK) throw std::runtime_error(std::string("artp_group_create_rank: ") + artp_status_string(rc));
  }
  static UniqueId uniqueId() {
    UniqueId id{};
    const int rc = artp_group_unique_id(id.data());
    if (rc != ARTP_OK) throw std::runtime_error(std::string("artp_group_unique_id: ") + artp_status_string(rc));
    return id;
  }
  ~DeviceGroup() {
    if (g_) artp_group_destroy(g_);
  }
  DeviceGroup(const DeviceGroup&) = delete;
  DeviceGroup& operator=(const DeviceGroup&) = delete;
  DeviceGroup(DeviceGroup&& o) noexcept : g_(o.g_) { o.g_ = nullptr; }
  DeviceGroup& operator=(DeviceGroup&& o) noexcept {
    if (this != &o) {
      if (g_) artp_group_destroy(g_);
      g_ = o.g_;
      o.g_ = nullptr;
    }
    return *this;
  }

  int worldSize() const { return artp_group_world_size(g_); }
  int localCount() const { return artp_group_local_count(g_); }
  int rank(int local) const { return artp_group_rank(g_, local); }
  artp_ctx* context(int local) const { return artp_group_ctx(g_, local); }  // owned by the group: install the map here
  artp_group* handle() const { return g_; }

  // the number of ranks the transport itself sees (an all-reduce of ones)
  int ranksSeen() {
    int seen = 0;
    check(artp_group_ranks_seen(g_, &seen), "artp_group_ranks_seen");
    return seen;
  }
  // batch = S candidates per rank and step; materialise_cap accepted states per rank looked for among the first `prefix`
  // candidates (0 = all) are re-materialised on every member (0 = bitmaps only)
  void configure(uint64_t seed, size_t batch, size_t materialise_cap, size_t prefix = 0) {
    check(artp_group_configure(g_, seed, batch, materialise_cap, prefix), "artp_group_configure");
  }
  void step(uint64_t step) { check(artp_group_sample_and_validate_step(g_, step), "artp_group_sample_and_validate_step"); }
  StepBuffers stepBuffers(int local, uint64_t step) const {
    StepBuffers b;
    check(artp_group_step_buffers(g_, local, step, &b.se3, &b.valid, &b.bits, &b.states, &b.counts), "artp_group_step_buffers");
    return b;
  }
  // the edge exchange: per local member the edges it owns (device pointers on that member's GPU), blocks of `cap` records
  void exchangeEdges(const std::vector<artp_group_edges>& per_local, size_t cap) {
    if (static_cast<int>(per_local.size()) != localCount()) throw std::invalid_argument("exchangeEdges: one entry per local member");
    check(artp_group_exchange_edges(g_, per_local.data(), cap), "artp_group_exchange_edges");
  }
  EdgeBuffers edgeBuffers(int local) const {
    EdgeBuffers b;
    check(artp_group_edge_buffers(g_, local, &b.records, &b.counts), "artp_group_edge_buffers");
    return b;
  }
  // waits for every stream of every local member, at most timeout_ms (< 0: no limit); a timeout is an exception that
  // names the member and the stream, after which abort() tears the communicators down
  void synchronize(int timeout_ms = -1) { check(artp_group_synchronize(g_, timeout_ms), "artp_group_synchronize"); }
  void abort() noexcept { (void)artp_group_abort(g_); }

  static uint64_t shardFirstIndex(uint64_t step, int rank, int world, uint64_t batch) {
    return artp_shard_first_index(step, rank, world, batch);
  }

 private:
  void check(int rc, const char* what) const {
    if (rc != ARTP_OK)
      throw std::runtime_error(std::string(what) + ": " + artp_status_string(rc) + " (" + artp_group_last_error(g_) + ")");
  }
  artp_group* g_ = nullptr;
};

}  // namespace art_planner
