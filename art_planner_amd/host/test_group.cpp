// Device groups through the C ABI (include/artp_c.h "multi-GPU"): what a C++ host like the reference's PlannerRos
// (art_planner_ros/src/planner_ros.cpp:250-319: one planner object, one planning thread) needs to use every GPU of the
// node -- no interpreter, no torch.  Runs with however many GPUs are visible:
//   * an RCCL group over ALL visible devices (W = 1 on a one-GPU box: communicator, all-gather and all-reduce are real),
//   * a peer-copy group of THREE ranks on device 0 (the W > 1 sharding / double-buffering / re-materialisation logic on
//     one GPU), and one over all devices when there are several.
// Every rank's gathered bitmap, accepted count, re-materialised states and edge records are compared with what ONE
// plain artp_ctx computes for that rank's shard [artp_shard_first_index(step, r, W, S), +S): bit-identical.
//   test_group <fixture.bin> [out.json]      (fixture: the map part of test_host.cpp's format)
// Exit code 0 = all equal; 3 = no GPU (the library has no CPU path).
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <thread>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "artp_c.h"
#include "art_planner/device_group.h"

namespace {

struct MapData {
  int32_t rows = 0, cols = 0;
  double geo[4] = {0, 0, 0, 0}, zb[2] = {0, 0};
  std::vector<float> layer[8];  // elevation, elevation_masked, cum_prob, cum_prob_rowwise_hack, normal_x/y/z, std
};

bool read_map(const char* path, MapData* m) {
  std::ifstream f(path, std::ios::binary);
  f.read(reinterpret_cast<char*>(&m->rows), 4);
  f.read(reinterpret_cast<char*>(&m->cols), 4);
  f.read(reinterpret_cast<char*>(m->geo), 32);
  const size_t cells = static_cast<size_t>(m->rows) * m->cols;
  for (auto& l : m->layer) {
    l.resize(cells);
    f.read(reinterpret_cast<char*>(l.data()), cells * 4);
  }
  f.read(reinterpret_cast<char*>(m->zb), 16);
  return static_cast<bool>(f);
}

int upload_map(artp_ctx* c, const MapData& m) {
  int rc = artp_upload_layer(c, ARTP_SLOT_BODY, m.layer[0].data(), m.rows, m.cols, m.geo[0], m.geo[1], m.geo[2], m.geo[3]);
  if (rc) return rc;
  rc = artp_upload_layer(c, ARTP_SLOT_FEET, m.layer[1].data(), m.rows, m.cols, m.geo[0], m.geo[1], m.geo[2], m.geo[3]);
  if (rc) return rc;
  // cum_prob_rowwise = column 0 of the "hack" layer (column-major: its first `rows` floats)
  rc = artp_upload_sampler_layers(c, m.layer[2].data(), m.layer[3].data(), m.layer[0].data(), m.layer[4].data(),
                                  m.layer[5].data(), m.layer[6].data(), m.layer[7].data(), m.rows, m.cols, m.geo[0],
                                  m.geo[1], m.geo[2], m.geo[3]);
  if (rc) return rc;
  return artp_set_z_bounds(c, m.zb[0], m.zb[1]);
}

#define CHECK(cond, ...)                        \
  do {                                          \
    if (!(cond)) {                              \
      std::printf("FAIL %s:%d: ", __FILE__, __LINE__); \
      std::printf(__VA_ARGS__);                 \
      std::printf("\n");                        \
      return 1;                                 \
    }                                           \
  } while (0)

template <class T>
std::vector<T> from_device(int device, const T* p, size_t n) {
  std::vector<T> h(n);
  (void)hipSetDevice(device);
  if (n && hipMemcpy(h.data(), p, n * sizeof(T), hipMemcpyDeviceToHost) != hipSuccess) h.clear();
  return h;
}

// what ONE plain context computes for (step, rank): the reference every member's buffers are compared with
struct Shard {
  std::vector<double> se3;
  std::vector<uint8_t> valid;
  std::vector<uint64_t> bits;
  std::vector<size_t> accepted;
};

int make_shard(artp_ctx* ref, uint64_t seed, uint64_t first, size_t S, Shard* sh) {
  sh->se3.resize(S * 7);
  sh->valid.resize(S);
  int rc = artp_sample_and_validate(ref, seed, first, S, sh->se3.data(), sh->valid.data(), nullptr);
  if (rc) return rc;
  sh->bits.assign((S + 63) / 64, 0);
  sh->accepted.clear();
  for (size_t i = 0; i < S; ++i)
    if (sh->valid[i]) {
      sh->bits[i >> 6] |= 1ull << (i & 63);
      sh->accepted.push_back(i);
    }
  return 0;
}

double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int check_step(artp_group* g, artp_ctx* ref, const std::vector<int>& devices, uint64_t seed, uint64_t step, size_t S,
               size_t cap, size_t prefix, bool own_buffers) {
  const int W = artp_group_world_size(g);
  const size_t words = (S + 63) / 64;
  std::vector<Shard> shards(W);
  for (int r = 0; r < W; ++r)
    CHECK(make_shard(ref, seed, artp_shard_first_index(step, r, W, S), S, &shards[r]) == 0, "reference shard: %s",
          artp_last_error(ref));
  for (int l = 0; l < artp_group_local_count(g); ++l) {
    const double* se3 = nullptr;
    const uint8_t* valid = nullptr;
    const uint64_t *bits = nullptr, *counts = nullptr;
    const double* states = nullptr;
    CHECK(artp_group_step_buffers(g, l, step, &se3, &valid, &bits, &states, &counts) == 0, "step_buffers");
    const int dev = devices[l], rank = artp_group_rank(g, l);
    if (own_buffers) {  // single-buffered: only meaningful for the LAST step issued
      auto h_se3 = from_device(dev, se3, S * 7);
      auto h_valid = from_device(dev, valid, S);
      CHECK(h_valid == shards[rank].valid, "step %llu member %d: own labels differ", (unsigned long long)step, l);
      CHECK(std::memcmp(h_se3.data(), shards[rank].se3.data(), S * 56) == 0, "step %llu member %d: own states differ",
            (unsigned long long)step, l);
    }
    auto h_bits = from_device(dev, bits, (size_t)W * words);
    auto h_counts = from_device(dev, counts, (size_t)W);
    auto h_states = cap ? from_device(dev, states, (size_t)W * cap * 7) : std::vector<double>();
    CHECK(h_bits.size() == (size_t)W * words, "copy back");
    for (int r = 0; r < W; ++r) {
      CHECK(std::memcmp(&h_bits[r * words], shards[r].bits.data(), words * 8) == 0,
            "step %llu member %d: bitmap of rank %d differs", (unsigned long long)step, l, r);
      if (!cap) continue;
      size_t in_prefix = 0;
      for (size_t i : shards[r].accepted) in_prefix += i < prefix;
      CHECK(h_counts[r] == in_prefix, "step %llu member %d rank %d: count %llu != %zu", (unsigned long long)step, l, r,
            (unsigned long long)h_counts[r], in_prefix);
      const size_t k = in_prefix < cap ? in_prefix : cap;
      for (size_t j = 0; j < k; ++j)
        CHECK(std::memcmp(&h_states[((size_t)r * cap + j) * 7], &shards[r].se3[shards[r].accepted[j] * 7], 56) == 0,
              "step %llu member %d rank %d: materialised state %zu differs", (unsigned long long)step, l, r, j);
    }
  }
  return 0;
}

int exercise_group(artp_group* g, const char* name, const std::vector<int>& devices, const MapData& map, artp_ctx* ref,
                   std::string* json);

int run_group(const char* name, const std::vector<int>& devices, int transport, const MapData& map, artp_ctx* ref,
              const artp_params& params, std::string* json) {
  artp_group* g = nullptr;
  int rc = artp_group_create(devices.data(), (int)devices.size(), &params, transport, &g);
  CHECK(rc == 0, "%s: artp_group_create -> %s", name, artp_status_string(rc));
  const int W = artp_group_world_size(g);
  CHECK(W == (int)devices.size() && artp_group_local_count(g) == W, "%s: sizes", name);
  for (int l = 0; l < W; ++l) CHECK(artp_group_rank(g, l) == l, "rank numbering");
  return exercise_group(g, name, devices, map, ref, json);
}

// everything a group is asked to do, on a group however it was made: `devices` = the devices of its LOCAL members (all W of
// a single-process group; one of a one-process-per-GPU rank).  Every process checks ALL W ranks' blocks on its members.
int exercise_group(artp_group* g, const char* name, const std::vector<int>& devices, const MapData& map, artp_ctx* ref,
                   std::string* json) {
  const int W = artp_group_world_size(g), NL = artp_group_local_count(g);
  CHECK(NL == (int)devices.size(), "%s: local members", name);
  for (int l = 0; l < NL; ++l)
    CHECK(upload_map(artp_group_ctx(g, l), map) == 0, "%s: map upload on member %d: %s", name, l,
          artp_last_error(artp_group_ctx(g, l)));
  int seen = 0;
  int rc = artp_group_ranks_seen(g, &seen);
  CHECK(rc == 0 && seen == W, "%s: ranks seen %d of %d (%s)", name, seen, W, artp_group_last_error(g));

  const uint64_t seed = 20260927;
  const size_t S = 20000, cap = 700, prefix = 9000;  // S not a multiple of 64, prefix inside a word, cap < count
  CHECK(artp_group_sample_and_validate_step(g, 0) != 0, "a step before configure must fail");
  CHECK(artp_group_configure(g, seed, S, cap, prefix) == 0, "%s: configure: %s", name, artp_group_last_error(g));
  // (1) step by step
  for (uint64_t step = 0; step < 3; ++step) {
    rc = artp_group_sample_and_validate_step(g, step);
    CHECK(rc == 0, "%s: step %llu: %s", name, (unsigned long long)step, artp_group_last_error(g));
    rc = artp_group_synchronize(g, 60000);
    CHECK(rc == 0, "%s: synchronize: %s", name, artp_group_last_error(g));
    if (check_step(g, ref, devices, seed, step, S, cap, prefix, true)) return 1;
  }
  // (2) five steps in flight, no host wait in between: the parity buffers are guarded by events on the device
  for (uint64_t step = 3; step < 8; ++step) {
    rc = artp_group_sample_and_validate_step(g, step);
    CHECK(rc == 0, "%s: step %llu: %s", name, (unsigned long long)step, artp_group_last_error(g));
  }
  CHECK(artp_group_synchronize(g, 60000) == 0, "%s: synchronize: %s", name, artp_group_last_error(g));
  if (check_step(g, ref, devices, seed, 6, S, cap, prefix, false)) return 1;
  if (check_step(g, ref, devices, seed, 7, S, cap, prefix, true)) return 1;
  // (3) bitmaps only
  CHECK(artp_group_configure(g, seed + 1, 4096, 0, 0) == 0, "reconfigure");
  CHECK(artp_group_sample_and_validate_step(g, 11) == 0, "step");
  CHECK(artp_group_synchronize(g, 60000) == 0, "synchronize");
  if (check_step(g, ref, devices, seed + 1, 11, 4096, 0, 4096, true)) return 1;

  // (4) the edge exchange: rank r owns n_r edges, every third (+r) valid
  const size_t ecap = 1500;
  std::vector<artp_group_edges> per(NL);
  std::vector<std::vector<uint32_t>> expect(W);
  std::vector<void*> to_free;
  for (int l = 0; l < W; ++l) {   // l = RANK: every process knows what every rank sends
    int local = -1;
    for (int q = 0; q < NL; ++q)
      if (artp_group_rank(g, q) == l) local = q;
    const size_t n = 1000 + 100 * (size_t)l;
    std::vector<uint8_t> v(n);
    std::vector<uint32_t> ei(n), ej(n);
    std::vector<float> cost(3 * n);
    for (size_t e = 0; e < n; ++e) {
      v[e] = ((e + l) % 3) == 0;
      ei[e] = (uint32_t)(7 * e + l);
      ej[e] = 0xfffffff0u - (uint32_t)e;  // ids use the full 32 bits
      for (int k = 0; k < 3; ++k) cost[3 * e + k] = 0.25f * (float)e + (float)k + 100.0f * (float)l;
      if (v[e]) {
        expect[l].push_back(ei[e]);
        expect[l].push_back(ej[e]);
        for (int k = 0; k < 3; ++k) {
          uint32_t b;
          std::memcpy(&b, &cost[3 * e + k], 4);
          expect[l].push_back(b);
        }
      }
    }
    if (local < 0) continue;   // another process's rank
    (void)hipSetDevice(devices[local]);
    void *dv, *di, *dj, *dc;
    CHECK(hipMalloc(&dv, n) == hipSuccess && hipMalloc(&di, 4 * n) == hipSuccess && hipMalloc(&dj, 4 * n) == hipSuccess &&
              hipMalloc(&dc, 12 * n) == hipSuccess,
          "hipMalloc");
    (void)hipMemcpy(dv, v.data(), n, hipMemcpyHostToDevice);
    (void)hipMemcpy(di, ei.data(), 4 * n, hipMemcpyHostToDevice);
    (void)hipMemcpy(dj, ej.data(), 4 * n, hipMemcpyHostToDevice);
    (void)hipMemcpy(dc, cost.data(), 12 * n, hipMemcpyHostToDevice);
    per[local] = {static_cast<uint8_t*>(dv), static_cast<uint32_t*>(di), static_cast<uint32_t*>(dj), static_cast<float*>(dc), n};
    for (void* p : {dv, di, dj, dc}) to_free.push_back(p);
  }
  for (int round = 0; round < 2; ++round) {  // twice: the second exchange waits for the first on the device
    rc = artp_group_exchange_edges(g, per.data(), ecap);
    CHECK(rc == 0, "%s: exchange_edges: %s", name, artp_group_last_error(g));
  }
  CHECK(artp_group_synchronize(g, 60000) == 0, "synchronize");
  for (int l = 0; l < NL; ++l) {
    const uint32_t* rec = nullptr;
    const uint64_t* cnt = nullptr;
    CHECK(artp_group_edge_buffers(g, l, &rec, &cnt) == 0, "edge_buffers");
    auto h_cnt = from_device(devices[l], cnt, (size_t)W);
    auto h_rec = from_device(devices[l], rec, (size_t)W * ecap * 5);
    for (int r = 0; r < W; ++r) {
      CHECK(h_cnt[r] * 5 == expect[r].size(), "%s: member %d: record count of rank %d", name, l, r);
      CHECK(std::memcmp(&h_rec[(size_t)r * ecap * 5], expect[r].data(), expect[r].size() * 4) == 0,
            "%s: member %d: records of rank %d differ", name, l, r);
    }
  }
  for (void* p : to_free) (void)hipFree(p);

  // (5) throughput of the whole step as a C++ host sees it (one call per step, nothing else on the host)
  const size_t SB = 1u << 20;
  CHECK(artp_group_configure(g, seed, SB, 1u << 14, 8u << 14) == 0, "configure (timing)");
  for (uint64_t s = 0; s < 3; ++s) CHECK(artp_group_sample_and_validate_step(g, s) == 0, "warmup");
  CHECK(artp_group_synchronize(g, 120000) == 0, "synchronize");
  const int K = 20;
  const double t0 = now_ms();
  for (uint64_t s = 0; s < (uint64_t)K; ++s) CHECK(artp_group_sample_and_validate_step(g, 100 + s) == 0, "timed step");
  const double t_enqueue = now_ms() - t0;
  CHECK(artp_group_synchronize(g, 120000) == 0, "synchronize");
  const double dt = now_ms() - t0;
  char buf[512];
  std::snprintf(buf, sizeof(buf),
                "{\"group\": \"%s\", \"world\": %d, \"ranks_seen\": %d, \"states_per_s\": %.4g, \"ms_per_step\": %.4f, "
                "\"host_enqueue_ms_per_step\": %.4f, \"batch_per_rank\": %zu}",
                name, W, seen, (double)W * SB * K / (dt * 1e-3), dt / K, t_enqueue / K, SB);
  std::printf("%s\n", buf);
  if (!json->empty()) *json += ", ";
  *json += buf;
  artp_group_destroy(g);
  return 0;
}

}  // namespace

// The same through the host mirror's class (art_planner/device_group.h): a two-rank peer-copy group on device 0, two steps,
// every rank's accepted count on member 0 against a plain context's count for that rank's shard; errors are exceptions.
int run_class(const MapData& map, artp_ctx* ref, const artp_params& params) {
  try {
    art_planner::DeviceGroup grp(std::vector<int>{0, 0}, params, ARTP_GROUP_PEER_COPY);
    art_planner::DeviceGroup moved(std::move(grp));  // movable, not copyable
    CHECK(moved.worldSize() == 2 && moved.localCount() == 2 && moved.rank(1) == 1, "class: sizes");
    for (int m = 0; m < moved.localCount(); ++m) CHECK(upload_map(moved.context(m), map) == 0, "class: map upload");
    CHECK(moved.ranksSeen() == 2, "class: ranks seen");
    const size_t S = 1 << 14;
    moved.configure(11, S, 256);
    double* d_se3 = nullptr;
    uint8_t* d_valid = nullptr;
    CHECK(hipMalloc(reinterpret_cast<void**>(&d_se3), S * 7 * sizeof(double)) == hipSuccess, "hipMalloc");
    CHECK(hipMalloc(reinterpret_cast<void**>(&d_valid), S) == hipSuccess, "hipMalloc");
    for (uint64_t step = 0; step < 2; ++step) {
      moved.step(step);
      moved.synchronize(20000);
      const art_planner::DeviceGroup::StepBuffers b = moved.stepBuffers(0, step);
      uint64_t counts[2] = {0, 0};
      CHECK(hipMemcpy(counts, b.counts, sizeof(counts), hipMemcpyDeviceToHost) == hipSuccess, "class: counts");
      for (int r = 0; r < 2; ++r) {
        uint64_t want = 0;
        CHECK(artp_sample_and_validate_dev(ref, 11, art_planner::DeviceGroup::shardFirstIndex(step, r, 2, S), S, d_se3, d_valid, &want) == 0,
              "class: reference shard");
        CHECK(artp_synchronize(ref) == 0, "class: reference sync");
        CHECK(counts[r] == want, "class: accepted count of rank %d at step %llu: %llu vs %llu", r, (unsigned long long)step,
              (unsigned long long)counts[r], (unsigned long long)want);
      }
    }
    (void)hipFree(d_se3);
    (void)hipFree(d_valid);
    bool threw = false;
    try {
      moved.exchangeEdges({}, 16);  // one entry per local member is required
    } catch (const std::invalid_argument&) {
      threw = true;
    }
    CHECK(threw, "class: argument check");
  } catch (const std::exception& e) {
    std::printf("FAIL: art_planner::DeviceGroup: %s\n", e.what());
    return 1;
  }
  std::printf("art_planner::DeviceGroup (class): ok\n");
  return 0;
}

int main(int argc, char** argv) {
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) {
    std::printf("no GPU: artp_group_create must refuse\n");
    artp_params p;
    artp_params_yaml(&p);
    artp_group* g = nullptr;
    const int dev0 = 0;
    return artp_group_create(&dev0, 1, &p, ARTP_GROUP_RCCL, &g) == ARTP_ERR_NO_DEVICE && !g ? 3 : 1;
  }
  if (argc < 2) return 2;
  MapData map;
  if (!read_map(argv[1], &map)) return 2;
  artp_params params;
  artp_params_yaml(&params);
  // One process per rank (torchrun / mpirun style), all on device 0:  test_group <fixture> --rank R W <id-file>
  // Rank 0 makes the id (artp_group_unique_id) and leaves it in <id-file>; the others wait for the file -- the launcher's
  // hand-off of the 128 bytes.  With the real RCCL this needs W GPUs; tests/test_host_mirror.py runs it on ONE GPU with
  // $ARTP_RCCL_LIB = tests/cpp/loopback_rccl.cpp (a test double behind the same ncclAllGather / ncclAllReduce call sites).
  if (argc >= 6 && std::string(argv[2]) == "--rank") {
    const int rank = std::atoi(argv[3]), world = std::atoi(argv[4]);
    const std::string id_file = argv[5];
    uint8_t id[ARTP_GROUP_ID_BYTES];
    if (rank == 0) {
      if (artp_group_unique_id(id) != 0) { std::printf("FAIL: artp_group_unique_id\n"); return 1; }
      std::ofstream o(id_file + ".tmp", std::ios::binary);
      o.write(reinterpret_cast<const char*>(id), sizeof(id));
      o.close();
      std::rename((id_file + ".tmp").c_str(), id_file.c_str());
    } else {
      bool got = false;
      for (int tries = 0; tries < 3000 && !got; ++tries) {
        std::ifstream in(id_file, std::ios::binary);
        got = in && in.read(reinterpret_cast<char*>(id), sizeof(id));
        if (!got) std::this_thread::sleep_for(std::chrono::milliseconds(10));
      }
      if (!got) { std::printf("FAIL: rank %d never saw the id file\n", rank); return 1; }
    }
    artp_ctx* ref_r = nullptr;
    if (artp_create(0, &params, &ref_r) != 0 || upload_map(ref_r, map) != 0) return 1;
    artp_group* g = nullptr;
    const int rc = artp_group_create_rank(0, rank, world, id, &params, &g);
    if (rc != 0) { std::printf("FAIL: artp_group_create_rank(rank %d of %d) -> %s\n", rank, world, artp_status_string(rc)); return 1; }
    if (artp_group_world_size(g) != world || artp_group_local_count(g) != 1 || artp_group_rank(g, 0) != rank) return 1;
    std::string json_r;
    const std::string nm = "rank_" + std::to_string(rank) + "_of_" + std::to_string(world) + "_processes_on_device_0";
    if (exercise_group(g, nm.c_str(), std::vector<int>{0}, map, ref_r, &json_r)) return 1;
    artp_destroy(ref_r);
    if (argc > 6) {
      std::ofstream o(argv[6]);
      o << json_r << "\n";
    }
    std::printf("test_group rank %d of %d ok\n", rank, world);
    return 0;
  }
  if (artp_shard_first_index(3, 2, 8, 1000) != (3ull * 8 + 2) * 1000) return 1;

  artp_ctx* ref = nullptr;
  if (artp_create(0, &params, &ref) != 0 || upload_map(ref, map) != 0) {
    std::printf("reference context: %s\n", ref ? artp_last_error(ref) : "artp_create failed");
    return 1;
  }
  std::string json;
  std::vector<int> all(n_dev);
  for (int d = 0; d < n_dev; ++d) all[d] = d;
  // argument errors
  {
    artp_group* g = nullptr;
    const int twice[2] = {0, 0};
    if (artp_group_create(twice, 2, &params, ARTP_GROUP_RCCL, &g) != ARTP_ERR_INVALID_ARG || g) {
      std::printf("FAIL: RCCL group with one device twice must be refused\n");
      return 1;
    }
    const int beyond = n_dev;
    if (artp_group_create(&beyond, 1, &params, ARTP_GROUP_RCCL, &g) != ARTP_ERR_NO_DEVICE || g) return 1;
  }
  if (run_group("rccl_all_devices", all, ARTP_GROUP_RCCL, map, ref, params, &json)) return 1;
  if (run_group("peer_copy_3_ranks_on_device_0", {0, 0, 0}, ARTP_GROUP_PEER_COPY, map, ref, params, &json)) return 1;
  if (n_dev > 1 && run_group("peer_copy_all_devices", all, ARTP_GROUP_PEER_COPY, map, ref, params, &json)) return 1;
  if (run_class(map, ref, params)) return 1;
  artp_destroy(ref);
  if (argc > 2) {
    std::ofstream o(argv[2]);
    o << "{\"visible_gpus\": " << n_dev << ", \"groups\": [" << json << "]}\n";
  }
  std::printf("test_group ok: %d visible GPU(s)\n", n_dev);
  return 0;
}
