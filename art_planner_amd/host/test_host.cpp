// Exercises the C++ host mirror (reference class names / signatures) over libartp.so, compiled against the
// STRICT OMPL-shaped stand-ins of ompl_min.h (every pure virtual of the real base classes is pure there).
//   test_host <fixture.bin> [latency_out.json]
// fixture: int32 rows, cols; float64 len_x len_y pos_x pos_y; float32 elevation[rows*cols] (col-major),
//          float32 elevation_masked, cum_prob, cum_prob_rowwise_hack, normal_x, normal_y, normal_z,
//          plane_fit_std_dev (rows*cols each); float64 z_low, z_high; int32 n; float64 se3[n*7]; uint8 expected[n]
//          (oracle labels); int32 m; float64 s1[m*7], s2[m*7]; uint8 motion_ok[m]; float64 last_t[m],
//          last_state[m*7] (oracle, DiscreteMotionValidator::checkMotion(s1, s2, lastValid));
//          stale-label case: int32 nb; uint64 seed_b; uint8 old_labels[nb], new_labels[nb] (oracle labels of the
//          states sample(seed_b, 0..nb) on the map as uploaded / after the rectangle update); int32 row0, col0,
//          nrows, ncols; float32 patch[nrows*ncols] (col-major, new body-layer samples)
// Exit code 0 = every label (single-state isValid AND batch), every checkMotion verdict and every lastValid pair
// equals the oracle's.  Without a GPU the context constructor must throw (no CPU fallback): exit code 3.
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <type_traits>
#include <vector>

#include "art_planner/objectives/motion_cost_objective.h"
#include "art_planner/planners/batch_prm.h"
#include "art_planner/sampler.h"
#include "art_planner/validity_checker/height_map_box_checker.h"
#include "art_planner/validity_checker/validity_checker.h"

using namespace art_planner;

// the mirror classes must be concrete against OMPL-1.4.2-shaped bases (std::make_shared in INTEGRATION.md)
static_assert(!std::is_abstract<SE3FromSE2Sampler>::value, "SE3FromSE2Sampler misses a StateSampler pure virtual");
static_assert(!std::is_abstract<BatchMotionValidator>::value, "BatchMotionValidator misses a MotionValidator pure virtual");
static_assert(!std::is_abstract<StateValidityChecker>::value, "StateValidityChecker misses isValid");
static_assert(std::is_abstract<ob::StateSampler>::value && std::is_abstract<ob::MotionValidator>::value, "stand-ins must be strict");

static void toState(const double* s, ob::SE3StateSpace::StateType* o) {
  o->setXYZ(s[0], s[1], s[2]);
  o->rotation().x = s[3]; o->rotation().y = s[4]; o->rotation().z = s[5]; o->rotation().w = s[6];
}

static double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char** argv) {
  auto params = std::make_shared<Params>();
  // shipped YAML robot (art_planner_ros/config/params.yaml:55-71)
  params->robot.torso.length = 1.31; params->robot.torso.width = 0.65; params->robot.torso.height = 0.3;
  params->robot.torso.offset.z = 0.04;
  params->robot.feet.offset.x = 0.51; params->robot.feet.offset.y = 0.2; params->robot.feet.offset.z = -0.475;
  params->robot.feet.reach.x = 0.2; params->robot.feet.reach.y = 0.2; params->robot.feet.reach.z = 0.2;
  GpuContextPtr gpu;
  try {
    gpu = std::make_shared<GpuContext>(params, 0);
  } catch (const std::exception& e) {
    std::printf("no GPU context: %s\n", e.what());
    return 3;
  }
  if (argc < 2) return 2;
  std::ifstream f(argv[1], std::ios::binary);
  int32_t rows, cols, n, m;
  double geo[4], zb[2];
  f.read(reinterpret_cast<char*>(&rows), 4);
  f.read(reinterpret_cast<char*>(&cols), 4);
  f.read(reinterpret_cast<char*>(geo), 32);
  const size_t cells = static_cast<size_t>(rows) * cols;
  const char* names[9] = {"elevation", "elevation_masked", "cum_prob", "cum_prob_rowwise_hack", "normal_x",
                          "normal_y", "normal_z", "plane_fit_std_dev", nullptr};
  auto map = std::make_shared<Map>();
  map->setGeometry({rows, cols, geo[0] / rows, geo[0], geo[1], geo[2], geo[3]});
  std::vector<float> layer(cells);
  for (int k = 0; names[k]; ++k) {
    f.read(reinterpret_cast<char*>(layer.data()), cells * 4);
    map->addLayer(names[k], layer.data());
  }
  f.read(reinterpret_cast<char*>(zb), 16);
  f.read(reinterpret_cast<char*>(&n), 4);
  std::vector<double> se3(static_cast<size_t>(n) * 7);
  std::vector<uint8_t> expected(n);
  f.read(reinterpret_cast<char*>(se3.data()), se3.size() * 8);
  f.read(reinterpret_cast<char*>(expected.data()), n);
  f.read(reinterpret_cast<char*>(&m), 4);
  std::vector<double> s1(static_cast<size_t>(m) * 7), s2(s1.size()), last_t(m), last_state(s1.size());
  std::vector<uint8_t> motion_ok(m);
  f.read(reinterpret_cast<char*>(s1.data()), s1.size() * 8);
  f.read(reinterpret_cast<char*>(s2.data()), s2.size() * 8);
  f.read(reinterpret_cast<char*>(motion_ok.data()), m);
  f.read(reinterpret_cast<char*>(last_t.data()), m * 8);
  f.read(reinterpret_cast<char*>(last_state.data()), last_state.size() * 8);
  int32_t nb = 0, rect[4] = {0, 0, 0, 0};
  uint64_t seed_b = 0;
  f.read(reinterpret_cast<char*>(&nb), 4);
  f.read(reinterpret_cast<char*>(&seed_b), 8);
  std::vector<uint8_t> old_labels(nb), new_labels(nb);
  f.read(reinterpret_cast<char*>(old_labels.data()), nb);
  f.read(reinterpret_cast<char*>(new_labels.data()), nb);
  f.read(reinterpret_cast<char*>(rect), 16);
  std::vector<float> patch(static_cast<size_t>(rect[2]) * rect[3]);
  f.read(reinterpret_cast<char*>(patch.data()), patch.size() * 4);
  if (!f) return 2;

  auto si = std::make_shared<ob::SpaceInformation>();
  StateValidityChecker checker(si, params, gpu);
  checker.setMap(map);
  checker.updateHeightField();
  if (!checker.hasMap()) return 4;

  int bad = 0;
  // ---- isValid: batch + single states (arbitrary states: the latency path) ----
  const auto batch = checker.isValidBatch(se3);
  for (int i = 0; i < n; ++i) bad += batch[i] != expected[i];
  const int n_single = n < 2000 ? n : 2000;
  ob::SE3StateSpace::StateType st;
  double t0 = now_us();
  for (int i = 0; i < n_single; ++i) {
    toState(&se3[7 * i], &st);
    bad += checker.isValid(&st) != (expected[i] != 0);
  }
  const double us_single = (now_us() - t0) / n_single;
  // the same through the persistent latency service (artp_set_persistent_latency: one resident workgroup polling a mailbox
  // in mapped host memory, no launch per call) -- same labels
  double us_single_svc = 0;
  {
    bad += artp_set_persistent_latency(gpu->get(), 1) != ARTP_OK;
    for (int i = 0; i < 50; ++i) {   // starts the service
      toState(&se3[7 * i], &st);
      bad += checker.isValid(&st) != (expected[i] != 0);
    }
    t0 = now_us();
    for (int i = 0; i < n_single; ++i) {
      toState(&se3[7 * i], &st);
      bad += checker.isValid(&st) != (expected[i] != 0);
    }
    us_single_svc = (now_us() - t0) / n_single;
    uint64_t svc_stats[2] = {0, 0};
    bad += artp_persistent_latency_stats(gpu->get(), svc_stats) != ARTP_OK;
    bad += (svc_stats[1] < (uint64_t)n_single || svc_stats[0] < 1) ? 1 : 0;
    bad += artp_set_persistent_latency(gpu->get(), 0) != ARTP_OK;
  }
  // a checker without a map must answer false, not throw (ob::StateValidityChecker::isValid never throws)
  {
    auto gpu2 = std::make_shared<GpuContext>(params, 0);
    StateValidityChecker unmapped(si, params, gpu2);
    toState(&se3[0], &st);
    bool threw = false, v = true;
    try { v = unmapped.isValid(&st); } catch (...) { threw = true; }
    bad += (threw || v || gpu2->errorCount() != 1) ? 1 : 0;
  }

  // ---- MotionValidator: both overloads against the oracle ----
  BatchMotionValidator mv(si, gpu);
  mv.setZBounds(zb[0], zb[1]);
  const auto mb = mv.checkMotionBatch(s1, s2);
  for (int i = 0; i < m; ++i) bad += mb[i] != motion_ok[i];
  const int m_single = m < 300 ? m : 300;
  ob::SE3StateSpace::StateType a, b, lv;
  int bad_last = 0;
  for (int i = 0; i < m_single; ++i) {
    toState(&s1[7 * i], &a);
    toState(&s2[7 * i], &b);
    bad += mv.checkMotion(&a, &b) != (motion_ok[i] != 0);
    std::pair<ob::State*, double> last(&lv, -7.0);
    lv.setXYZ(1e9, 1e9, 1e9);
    const bool ok = mv.checkMotion(&a, &b, last);
    bad += ok != (motion_ok[i] != 0);
    if (ok) {
      bad_last += (last.second != -7.0 || lv.getX() != 1e9) ? 1 : 0;  // untouched on success, like OMPL
    } else {
      double got[7];
      flattenSE3(&lv, got);
      bool same = last.second == last_t[i];
      for (int k = 0; k < 7; ++k) same = same && std::fabs(got[k] - last_state[7 * i + k]) <= 1e-12;
      bad_last += same ? 0 : 1;
    }
  }
  bad += bad_last;
  bad += (mv.getCheckedMotionCount() != 2u * m_single) ? 1 : 0;
  // ---- per-call latency of the MotionValidator seam (prm_motion_cost.cpp:652, lazy_prm_star_min_update.cpp:725 and
  // OMPL's PathSimplifier call checkMotion one edge at a time): n = 1 through both overloads, a 32-edge solution path
  // in one artp_check_motions call, and the same two with the latency kernel switched off (the batch pipeline)
  double us_cm1 = 0, us_cm1_last = 0, us_cm32 = 0, us_cm1_batch = 0, us_cm32_batch = 0, us_cm1_pool = 0, us_cm1_last_pool = 0;
  double us_cm1_short = 0, us_cm1_short_pool = 0;
  int n_short = 0;
  {
    const int reps = 400;
    for (int pass = 0; pass < 2; ++pass) {   // pass 0 warms up
      double t1 = now_us();
      for (int i = 0; i < reps; ++i) {
        toState(&s1[7 * (i % m_single)], &a);
        toState(&s2[7 * (i % m_single)], &b);
        bad += mv.checkMotion(&a, &b) != (motion_ok[i % m_single] != 0);
      }
      us_cm1 = (now_us() - t1) / reps;
      t1 = now_us();
      for (int i = 0; i < reps; ++i) {
        toState(&s1[7 * (i % m_single)], &a);
        toState(&s2[7 * (i % m_single)], &b);
        std::pair<ob::State*, double> last(&lv, -7.0);
        bad += mv.checkMotion(&a, &b, last) != (motion_ok[i % m_single] != 0);
      }
      us_cm1_last = (now_us() - t1) / reps;
    }
    // the same two through the resident edge pool (artp_set_persistent_latency: no launch per call)
    bad += artp_set_persistent_latency(gpu->get(), 1) != ARTP_OK;
    for (int pass = 0; pass < 2; ++pass) {
      double t1 = now_us();
      for (int i = 0; i < reps; ++i) {
        toState(&s1[7 * (i % m_single)], &a);
        toState(&s2[7 * (i % m_single)], &b);
        bad += mv.checkMotion(&a, &b) != (motion_ok[i % m_single] != 0);
      }
      us_cm1_pool = (now_us() - t1) / reps;
      t1 = now_us();
      for (int i = 0; i < reps; ++i) {
        toState(&s1[7 * (i % m_single)], &a);
        toState(&s2[7 * (i % m_single)], &b);
        std::pair<ob::State*, double> last(&lv, -7.0);
        const bool ok = mv.checkMotion(&a, &b, last);
        bad += ok != (motion_ok[i % m_single] != 0);
        if (!ok) bad += last.second != last_t[i % m_single];   // (untouched on success, like OMPL)
      }
      us_cm1_last_pool = (now_us() - t1) / reps;
    }
    bad += artp_set_persistent_latency(gpu->get(), 0) != ARTP_OK;
    // the PRM-sized motions of the fixture alone (end states less than 2 m apart: what the reference's planners connect),
    // one launch per call and through the resident pool
    {
      std::vector<int> short_idx;
      for (int i = 0; i < m_single; ++i)
        if (std::hypot(s1[7 * i] - s2[7 * i], s1[7 * i + 1] - s2[7 * i + 1]) < 2.0) short_idx.push_back(i);
      n_short = (int)short_idx.size();
      if (n_short >= 20) {
        for (int svc = 0; svc < 2; ++svc) {
          bad += artp_set_persistent_latency(gpu->get(), svc) != ARTP_OK;
          double t1 = 0;
          for (int pass = 0; pass < 2; ++pass) {
            t1 = now_us();
            for (int i = 0; i < reps; ++i) {
              const int j = short_idx[i % n_short];
              toState(&s1[7 * j], &a);
              toState(&s2[7 * j], &b);
              bad += mv.checkMotion(&a, &b) != (motion_ok[j] != 0);
            }
            t1 = (now_us() - t1) / reps;
          }
          (svc ? us_cm1_short_pool : us_cm1_short) = t1;
        }
        bad += artp_set_persistent_latency(gpu->get(), 0) != ARTP_OK;
      }
    }
    const int np = m < 32 ? m : 32;
    std::vector<uint8_t> okp(np);
    auto time_path = [&](int calls) {
      double t1 = now_us();
      for (int i = 0; i < calls; ++i) {
        const int at = (i * np) % (m - np + 1);
        if (artp_check_motions(gpu->get(), &s1[7 * at], &s2[7 * at], np, okp.data()) != ARTP_OK) ++bad;
        for (int k = 0; k < np; ++k) bad += okp[k] != motion_ok[at + k];
      }
      return (now_us() - t1) / calls;
    };
    time_path(20);
    us_cm32 = time_path(200);
    artp_set_few_edges(gpu->get(), 0);
    time_path(5);
    us_cm32_batch = time_path(50);
    double t1 = now_us();
    for (int i = 0; i < 100; ++i) {
      toState(&s1[7 * (i % m_single)], &a);
      toState(&s2[7 * (i % m_single)], &b);
      bad += mv.checkMotion(&a, &b) != (motion_ok[i % m_single] != 0);
    }
    us_cm1_batch = (now_us() - t1) / 100;
    artp_set_few_edges(gpu->get(), 1);
  }

  // ---- StateSampler: the planners' rejection loop; sampler-issued states are pre-validated ----
  ob::SE3StateSpace space;
  ob::RealVectorBounds bounds(3);
  bounds.setLow(0, geo[2] - geo[0]); bounds.setHigh(0, geo[2] + geo[0]);   // planner.cpp:146-156
  bounds.setLow(1, geo[3] - geo[1]); bounds.setHigh(1, geo[3] + geo[1]);
  bounds.setLow(2, zb[0]); bounds.setHigh(2, zb[1]);
  space.setBounds(bounds);
  SE3FromSE2SamplerAllocator alloc(params, gpu);
  alloc.setMap(map);
  checker.updateHeightField();
  auto sampler = alloc.getSampler(&space);
  const int n_loop = 200000;
  int accepted = 0, attempts = 0;
  std::vector<double> acc_states;
  t0 = now_us();
  std::vector<double> rej_states;
  while (attempts < n_loop) {
    ob::SE3StateSpace::StateType s;
    bool ok = false;
    do {  // prm_motion_cost.cpp:174-186 / lazy_prm_star_min_update.cpp:552-554
      sampler->sampleUniform(&s);
      ++attempts;
      ok = checker.isValid(&s);
      if (!ok && rej_states.size() < 64 * 7) {
        double fl[7];
        flattenSE3(&s, fl);
        rej_states.insert(rej_states.end(), fl, fl + 7);
      }
    } while (!ok && attempts < n_loop);
    if (!ok) break;
    if (accepted < 64) {
      double fl[7];
      flattenSE3(&s, fl);
      acc_states.insert(acc_states.end(), fl, fl + 7);
    }
    ++accepted;
  }
  const double us_loop = (now_us() - t0) / attempts;
  // the looked-up labels are the labels of the batch kernel
  for (uint8_t v : checker.isValidBatch(acc_states)) bad += v ? 0 : 1;
  for (uint8_t v : checker.isValidBatch(rej_states)) bad += v ? 1 : 0;
  bad += (acc_states.size() < 64 * 7 || rej_states.empty()) ? 1 : 0;
  // a map update voids the published labels: after updateHeightField() lookups must miss (-> latency path)
  checker.updateHeightField();
  {
    toState(&acc_states[0], &st);
    uint8_t lab = 0;
    double fl[7];
    flattenSE3(&st, fl);
    bad += gpu->lookupLabel(fl, &lab) ? 1 : 0;
    bad += checker.isValid(&st) ? 0 : 1;  // same map data: still valid, now through a launch
  }
  // The label cache follows the map version INSIDE the artp_ctx: a rectangle update issued straight through
  // GpuContext::get() (what INTEGRATION.md tells a 10 Hz integrator to call) -- no mirror class involved -- must
  // take effect for the very next isValid() on a sampler-issued state (validity_checker.cpp:26-29: updateHeightField
  // is effective immediately in the reference).  Labels before / after against the oracle on the old / new map.
  int stale_flips = 0;
  {
    SE3FromSE2Sampler fresh(&space, map, params, gpu, seed_b, static_cast<size_t>(nb));
    std::vector<double> issued(static_cast<size_t>(nb) * 7);
    const uint64_t v0 = artp_map_version(gpu->get());
    for (int i = 0; i < nb; ++i) {
      fresh.sampleUniform(&st);
      flattenSE3(&st, &issued[7 * i]);
      uint8_t lab = 2;
      bad += gpu->lookupLabel(&issued[7 * i], &lab) ? 0 : 1;  // pre-validated: a lookup, not a launch
      bad += lab != old_labels[i];
      bad += checker.isValid(&st) != (old_labels[i] != 0);
    }
    bad += artp_update_layer_rect(gpu->get(), ARTP_SLOT_BODY, patch.data(), rect[0], rect[1], rect[2], rect[3]) != ARTP_OK;
    bad += artp_map_version(gpu->get()) == v0 ? 1 : 0;
    for (int i = 0; i < nb; ++i) {
      uint8_t lab = 2;
      bad += gpu->lookupLabel(&issued[7 * i], &lab) ? 1 : 0;  // the block died with its map version
      toState(&issued[7 * i], &st);
      bad += checker.isValid(&st) != (new_labels[i] != 0);
      stale_flips += old_labels[i] != new_labels[i];
    }
    bad += stale_flips < 20 ? 1 : 0;  // the fixture must have teeth
    // a block sampled on the new map is served again, with the new map's labels
    SE3FromSE2Sampler again(&space, map, params, gpu, seed_b, static_cast<size_t>(nb));
    for (int i = 0; i < nb; ++i) {
      again.sampleUniform(&st);
      double fl[7];
      flattenSE3(&st, fl);
      uint8_t lab = 2;
      bad += gpu->lookupLabel(fl, &lab) ? 0 : 1;
      bad += lab != new_labels[i];
    }
    checker.updateHeightField();  // back to the fixture's map for what follows
  }
  // sampleUniformNear / sampleGaussian (sampler.cpp:135-187): inside the bounds, yaw-only rotation
  {
    ob::SE3StateSpace::StateType near, out;
    toState(&acc_states[0], &near);
    for (int i = 0; i < 200; ++i) {
      out.rotation().setIdentity();
      if (i & 1) sampler->sampleUniformNear(&out, &near, 0.5); else sampler->sampleGaussian(&out, &near, 0.3);
      const bool in_bounds = out.getX() >= bounds.low[0] && out.getX() <= bounds.high[0] && out.getY() >= bounds.low[1] &&
                             out.getY() <= bounds.high[1] && out.getZ() >= bounds.low[2] && out.getZ() <= bounds.high[2];
      const bool near_enough = !(i & 1) || (std::fabs(out.getX() - near.getX()) <= 0.5 && std::fabs(out.getY() - near.getY()) <= 0.5);
      const bool yaw_only = out.rotation().x == 0 && out.rotation().y == 0 &&
                            std::fabs(out.rotation().w * out.rotation().w + out.rotation().z * out.rotation().z - 1.0) < 1e-12;
      bad += (in_bounds && near_enough && yaw_only) ? 0 : 1;
    }
  }

  // HeightMapBoxChecker at the dPose boundary: an identity-rotation torso far above the map never hits
  HeightMapBoxChecker box(gpu, ARTP_SLOT_BODY, 1.31f, 0.65f, 0.3f);
  box.setHeightField(map, "elevation");
  HeightMapBoxChecker::dPose high;
  high.origin = {0.f, 0.f, 50.f, 0.f};
  bad += box.checkCollision({high}) != 0;
  // BatchPRM compiles against the same Params and builds a small roadmap between two accepted states
  {
    auto p2 = std::make_shared<Params>(*params);
    p2->planner.prm_motion_cost.max_n_vertices = 300;
    BatchPRM prm(p2, gpu);
    toState(&acc_states[0], &a);
    toState(&acc_states[7 * 40], &b);
    bool threw = false;
    try {
      prm.sampleGraph(a, b);
    } catch (const std::exception& e) {
      std::printf("BatchPRM: %s\n", e.what());
      threw = true;
    }
    bad += (threw || prm.numVertices() != 302) ? 1 : 0;
  }
  std::printf("isValid on an arbitrary state: %.1f us per call (one launch), %.1f us through the persistent service\n", us_single,
              us_single_svc);
  std::printf("checkMotion per call: 1 edge %.1f us (lastValid overload %.1f us), 32-edge path %.1f us; through the batch "
              "pipeline: %.1f us / %.1f us; 1 edge through the resident pool: %.1f us (lastValid overload %.1f us); the %d motions "
              "shorter than 2 m alone: %.1f us per call, %.1f us through the pool\n", us_cm1,
              us_cm1_last, us_cm32, us_cm1_batch, us_cm32_batch, us_cm1_pool, us_cm1_last_pool, n_short, us_cm1_short, us_cm1_short_pool);
  std::printf("host mirror: %d states batch + %d single (%.1f us per isValid on arbitrary states), %d motions (%d lastValid "
              "mismatches), rejection loop %d attempts / %d accepted at %.3f us per sampleUniform+isValid, %d labels flipped by a "
              "direct artp_update_layer_rect and served fresh, %d mismatches\n",
              n, n_single, us_single, m, bad_last, attempts, accepted, us_loop, stale_flips, bad);
  if (argc > 2) {
    std::ofstream o(argv[2]);
    o << "{\"isvalid_arbitrary_state_us\": " << us_single << ", \"isvalid_arbitrary_state_us_persistent_service\": " << us_single_svc
      << ", \"check_motion_1_edge_us\": " << us_cm1
      << ", \"check_motion_last_valid_1_edge_us\": " << us_cm1_last << ", \"check_motions_32_edge_path_us\": " << us_cm32
      << ", \"check_motion_1_edge_us_resident_pool\": " << us_cm1_pool
      << ", \"check_motion_last_valid_1_edge_us_resident_pool\": " << us_cm1_last_pool
      << ", \"check_motion_1_edge_us_motions_under_2m\": " << us_cm1_short
      << ", \"check_motion_1_edge_us_motions_under_2m_resident_pool\": " << us_cm1_short_pool
      << ", \"motions_under_2m\": " << n_short
      << ", \"check_motion_1_edge_us_batch_pipeline\": " << us_cm1_batch
      << ", \"check_motions_32_edge_path_us_batch_pipeline\": " << us_cm32_batch
      << ", \"sampler_loop_us_per_state\": " << us_loop
      << ", \"loop_attempts\": " << attempts << ", \"loop_accepted\": " << accepted << "}\n";
  }
  return bad == 0 ? 0 : 1;
}
