// art_planner::SE3FromSE2Sampler with the reference's interface (art_planner/include/art_planner/
// sampler.h:21-62): all three ob::StateSampler virtuals.
//
// sampleUniform() hands out states from a block that the GPU sampled AND validated in one go
// (artp_sample_and_validate); the labels are published to the GpuContext, so the isValid() that follows in the
// planners' rejection loops (prm_motion_cost.cpp:174-186, lazy_prm_star_min_update.cpp:552-554) is a lookup.
// State k of the stream is a pure function of (seed, k) -- counter-based uniforms in the reference's draw order
// (sampler.cpp:58-59,105,116) instead of OMPL's mt19937 stream (SURVEY.md 8c "same seed").
// sampleUniformNear() / sampleGaussian() are the reference's (sampler.cpp:135-187): position from
// RealVectorStateSampler, yaw-only rotation from SO2StateSampler, evaluated on the host (one state per call, a
// dozen flops) with the same counter-based generator.
//
// Differences from the reference's class, on purpose: the constructor takes the GpuContext (+ seed, block size),
// and updateMap() uploads the layers the sampler reads (the reference reads them through Map on every call).
#pragma once

#include <cmath>
#include <cstdint>
#include <memory>
#include <vector>

#include "art_planner/gpu_context.h"
#include "art_planner/map/map.h"
#include "art_planner/ompl_min.h"

namespace ob = ompl::base;

namespace art_planner {

// uniform01 of the device sampler (art_planner_amd/csrc/kernels.h: uniform01): splitmix64 over (seed, index, k)
inline double counterUniform01(uint64_t seed, uint64_t index, unsigned k) {
  uint64_t x = seed + 0x9E3779B97F4A7C15ULL * (index * 8u + static_cast<uint64_t>(k) + 1u);
  x ^= x >> 30;
  x *= 0xBF58476D1CE4E5B9ULL;
  x ^= x >> 27;
  x *= 0x94D049BB133111EBULL;
  x ^= x >> 31;
  x += seed;
  x ^= x >> 30;
  x *= 0xBF58476D1CE4E5B9ULL;
  x ^= x >> 27;
  x *= 0x94D049BB133111EBULL;
  x ^= x >> 31;
  return static_cast<double>(x >> 11) * (1.0 / 9007199254740992.0);
}

// utils.h:80-88 (Scalar = float return type in the reference; the arithmetic is double)
inline float getYawFromSO3f(const ob::SO3StateSpace::StateType& s) {
  return static_cast<float>(std::atan2(2 * (s.w * s.z + s.x * s.y), 1 - 2 * (s.y * s.y + s.z * s.z)));
}
// utils.h:92-97, literally: cos / sin of the FULL yaw, as the reference has it
inline void setSO3FromYaw(ob::SO3StateSpace::StateType& s, double yaw) {
  s.w = std::cos(yaw);
  s.x = 0;
  s.y = 0;
  s.z = std::sin(yaw);
}

class SE3FromSE2Sampler : public ob::StateSampler {
 public:
  SE3FromSE2Sampler(const ob::StateSpace* space, const std::shared_ptr<Map>& map, const ParamsConstPtr& params,
                    const GpuContextPtr& gpu, uint64_t seed = 42, size_t block = 4096)
      : ob::StateSampler(space), params_(params), gpu_(gpu), map_(map), seed_(seed), block_(block) {}

  // uploads the layers samplePositionInMapFromDist / sampleUniform read (sampler.cpp:61-63,99-103)
  void updateMap() {
    const auto g = map_->getGeometry();
    const auto& hack = map_->getLayer("cum_prob_rowwise_hack");  // column 0 holds the row CDF
    throwOnError(gpu_->get(),
                 artp_upload_sampler_layers(gpu_->get(), map_->getLayer("cum_prob").data(), hack.data(),
                                            map_->getLayer(params_->planner.elevation_layer).data(),
                                            map_->getLayer("normal_x").data(), map_->getLayer("normal_y").data(),
                                            map_->getLayer("normal_z").data(),
                                            map_->getLayer("plane_fit_std_dev").data(), g.rows, g.cols,
                                            g.length_x, g.length_y, g.position_x, g.position_y),
                 "artp_upload_sampler_layers");
    gpu_->mapChanged();
    current_.reset();
    cursor_ = 0;
  }

  // sampler.cpp:82-131
  void sampleUniform(ob::State* state) override {
    if (!current_ || cursor_ >= current_->size() || current_->epoch() != gpu_->mapEpoch()) refill();  // new map: new block
    current_->setHint(cursor_);
    const double* s = current_->state(cursor_++);
    auto* se3 = state->as<ob::SE3StateSpace::StateType>();
    se3->setXYZ(s[0], s[1], s[2]);
    se3->rotation().x = s[3];
    se3->rotation().y = s[4];
    se3->rotation().z = s[5];
    se3->rotation().w = s[6];
  }

  // sampler.cpp:135-159.  RealVectorStateSampler::sampleUniformNear: per dimension uniform in
  // [max(low, near - d), min(high, near + d)]; SO2StateSampler::sampleUniformNear: uniform in near -+ d, wrapped
  // to [-pi, pi).  Like the reference, the SO2 centre is the yaw of the OUTPUT state's current rotation
  // (`getYawFromSO3(state_se3->rotation())`, :145), not of `near`.
  void sampleUniformNear(ob::State* state, const ob::State* near, double distance) override {
    auto* state_se3 = state->as<ob::SE3StateSpace::StateType>();
    const auto* near_se3 = near->as<ob::SE3StateSpace::StateType>();
    const double near_pos[3] = {near_se3->getX(), near_se3->getY(), near_se3->getZ()};
    const double near_rot = getYawFromSO3f(state_se3->rotation());
    const auto& bounds = space_->as<ob::SE3StateSpace>()->getBounds();
    const uint64_t idx = aux_index_++;
    double pos[3];
    for (unsigned i = 0; i < 3; ++i) {
      const double lo = std::max(bounds.low[i], near_pos[i] - distance);
      const double hi = std::min(bounds.high[i], near_pos[i] + distance);
      pos[i] = (hi - lo) * counterUniform01(seed_ ^ kAuxStream, idx, i) + lo;  // RNG::uniformReal
    }
    state_se3->setXYZ(pos[0], pos[1], pos[2]);
    const double lo = near_rot - distance, hi = near_rot + distance;
    setSO3FromYaw(state_se3->rotation(), enforceSO2((hi - lo) * counterUniform01(seed_ ^ kAuxStream, idx, 3) + lo));
  }

  // sampler.cpp:163-187.  RealVectorStateSampler::sampleGaussian: per dimension N(mean, std_dev) clamped to the
  // bounds; SO2StateSampler::sampleGaussian: N(mean, std_dev) wrapped.  Same yaw-centre quirk as above (:173).
  void sampleGaussian(ob::State* state, const ob::State* mean, double std_dev) override {
    auto* state_se3 = state->as<ob::SE3StateSpace::StateType>();
    const auto* mean_se3 = mean->as<ob::SE3StateSpace::StateType>();
    const double mean_pos[3] = {mean_se3->getX(), mean_se3->getY(), mean_se3->getZ()};
    const double mean_rot = getYawFromSO3f(state_se3->rotation());
    const auto& bounds = space_->as<ob::SE3StateSpace>()->getBounds();
    const uint64_t idx = aux_index_++;
    double pos[3];
    for (unsigned i = 0; i < 3; ++i) {
      double v = counterGaussian(idx, 2 * i) * std_dev + mean_pos[i];  // RNG::gaussian
      if (v < bounds.low[i])
        v = bounds.low[i];
      else if (v > bounds.high[i])
        v = bounds.high[i];
      pos[i] = v;
    }
    state_se3->setXYZ(pos[0], pos[1], pos[2]);
    setSO3FromYaw(state_se3->rotation(), enforceSO2(counterGaussian(idx, 6) * std_dev + mean_rot));
  }

  // batch form for the batched planner loops: states [first, first + n) of the stream
  std::vector<double> sampleBatch(uint64_t first_index, size_t n) const {
    std::vector<double> out(n * 7);
    throwOnError(gpu_->get(), artp_sample_states(gpu_->get(), seed_, first_index, n, out.data()),
                 "artp_sample_states");
    return out;
  }

 private:
  static constexpr uint64_t kAuxStream = 0x6e656172676175ULL;  // keeps Near / Gaussian draws off the uniform stream

  // standard normal from two counter-based uniforms (Box-Muller); draws k and k + 1 of sample idx
  double counterGaussian(uint64_t idx, unsigned k) const {
    double u1 = counterUniform01(seed_ ^ kAuxStream, idx, k);
    const double u2 = counterUniform01(seed_ ^ kAuxStream, idx, k + 1);
    if (u1 < 1e-300) u1 = 1e-300;
    return std::sqrt(-2.0 * std::log(u1)) * std::cos(2.0 * M_PI * u2);
  }
  // SO2StateSpace::enforceBounds
  static double enforceSO2(double value) {
    double v = std::fmod(value, 2.0 * M_PI);
    if (v < -M_PI)
      v += 2.0 * M_PI;
    else if (v >= M_PI)
      v -= 2.0 * M_PI;
    return v;
  }

  void refill() {
    std::vector<double> se3(block_ * 7);
    std::vector<uint8_t> labels(block_);
    uint64_t epoch = gpu_->mapEpoch();
    // the version the labels were computed on comes back from the call itself (taken under the context's lock)
    int rc = artp_sample_and_validate(gpu_->get(), seed_, next_index_, block_, se3.data(), labels.data(), &epoch);
    const bool validated = rc == ARTP_OK;
    if (rc == ARTP_ERR_NO_MAP)  // height fields not uploaded (yet): plain sampling, isValid() launches
      rc = artp_sample_states(gpu_->get(), seed_, next_index_, block_, se3.data());
    throwOnError(gpu_->get(), rc, "artp_sample_and_validate");
    next_index_ += block_;
    cursor_ = 0;
    current_ = std::make_shared<ValidatedStateBlock>(std::move(se3), std::move(labels), epoch, validated);
    if (validated) slot_ = gpu_->publishBlock(slot_, current_);
  }

  ParamsConstPtr params_;
  GpuContextPtr gpu_;
  std::shared_ptr<Map> map_;
  uint64_t seed_;
  size_t block_;
  uint64_t next_index_{0};
  uint64_t aux_index_{0};
  size_t cursor_{0};
  int slot_{-1};
  std::shared_ptr<ValidatedStateBlock> current_;
};

class SE3FromSE2SamplerAllocator {
 public:
  SE3FromSE2SamplerAllocator(const ParamsConstPtr& params, const GpuContextPtr& gpu) : params_(params), gpu_(gpu) {}
  void setMap(const std::shared_ptr<Map>& map) { map_ = map; }
  std::shared_ptr<SE3FromSE2Sampler> getSampler(const ob::StateSpace* space) {
    auto s = std::make_shared<SE3FromSE2Sampler>(space, map_, params_, gpu_);
    s->updateMap();
    return s;
  }

 private:
  ParamsConstPtr params_;
  GpuContextPtr gpu_;
  std::shared_ptr<Map> map_;
};

}  // namespace art_planner
