// art_planner::SE3FromSE2Sampler with the reference's interface (art_planner/include/art_planner/
// sampler.h:21-62).  sampleUniform() hands out states from a block sampled on the GPU; state k of the
// stream is a pure function of (seed, k) (counter-based uniforms in the reference's draw order,
// sampler.cpp:58-59,105,116) instead of OMPL's mt19937 stream.
#pragma once

#include <memory>
#include <vector>

#include "art_planner/gpu_context.h"
#include "art_planner/map/map.h"
#include "art_planner/ompl_min.h"

namespace ob = ompl::base;

namespace art_planner {

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
    buffer_.clear();
    cursor_ = 0;
  }

  void sampleUniform(ob::State* state) override {
    if (cursor_ * 7 >= buffer_.size()) {
      buffer_.resize(block_ * 7);
      throwOnError(gpu_->get(), artp_sample_states(gpu_->get(), seed_, next_index_, block_, buffer_.data()),
                   "artp_sample_states");
      next_index_ += block_;
      cursor_ = 0;
    }
    const double* s = buffer_.data() + 7 * cursor_++;
    auto* se3 = state->as<ob::SE3StateSpace::StateType>();
    se3->setXYZ(s[0], s[1], s[2]);
    se3->rotation().x = s[3];
    se3->rotation().y = s[4];
    se3->rotation().z = s[5];
    se3->rotation().w = s[6];
  }

  // batch form for the batched planner loops: states [first, first + n) of the stream
  std::vector<double> sampleBatch(uint64_t first_index, size_t n) const {
    std::vector<double> out(n * 7);
    throwOnError(gpu_->get(), artp_sample_states(gpu_->get(), seed_, first_index, n, out.data()),
                 "artp_sample_states");
    return out;
  }

 private:
  ParamsConstPtr params_;
  GpuContextPtr gpu_;
  std::shared_ptr<Map> map_;
  uint64_t seed_;
  size_t block_;
  uint64_t next_index_{0};
  size_t cursor_{0};
  std::vector<double> buffer_;
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
