// art_planner::StateValidityChecker with the reference's interface
// (art_planner/include/art_planner/validity_checker/validity_checker.h:22-47), plus the batch entry the
// new batched planner loops use.  isValid() on a single state is a batch of one on the GPU -- there is
// no CPU path in this library.
#pragma once

#include <memory>
#include <vector>

#include "art_planner/gpu_context.h"
#include "art_planner/map/map.h"
#include "art_planner/ompl_min.h"

namespace ob = ompl::base;

namespace art_planner {

inline void flattenSE3(const ob::State* state, double out[7]) {
  const auto* s = state->as<ob::SE3StateSpace::StateType>();
  out[0] = s->getX();
  out[1] = s->getY();
  out[2] = s->getZ();
  out[3] = s->rotation().x;
  out[4] = s->rotation().y;
  out[5] = s->rotation().z;
  out[6] = s->rotation().w;
}

class StateValidityChecker : public ob::StateValidityChecker {
 public:
  StateValidityChecker(const ob::SpaceInformationPtr& si, const ParamsConstPtr& params, const GpuContextPtr& gpu)
      : ob::StateValidityChecker(si), params_(params), gpu_(gpu) {}

  void setMap(const std::shared_ptr<Map>& map) { map_ = map; }

  // validity_checker.cpp:26-29 -> both HeightMapBoxChecker::setHeightField calls
  // (validity_checker_body.cpp:52-55 uses params.planner.elevation_layer, validity_checker_feet.cpp:80-83
  // uses "elevation_masked")
  void updateHeightField() {
    const auto g = map_->getGeometry();
    const auto& body = map_->getLayer(params_->planner.elevation_layer);
    const auto& feet = map_->getLayer("elevation_masked");
    throwOnError(gpu_->get(), artp_upload_layer(gpu_->get(), ARTP_SLOT_BODY, body.data(), g.rows, g.cols, g.length_x,
                                                g.length_y, g.position_x, g.position_y), "artp_upload_layer");
    throwOnError(gpu_->get(), artp_upload_layer(gpu_->get(), ARTP_SLOT_FEET, feet.data(), g.rows, g.cols, g.length_x,
                                                g.length_y, g.position_x, g.position_y), "artp_upload_layer");
    has_field_ = true;
  }

  bool hasMap() const { return static_cast<bool>(map_) && has_field_; }

  // validity_checker.cpp:39-45
  bool isValid(const ob::State* state) const override {
    double s[7];
    flattenSE3(state, s);
    uint8_t v = 0;
    throwOnError(gpu_->get(), artp_validate_states(gpu_->get(), s, 1, &v, nullptr), "artp_validate_states");
    return v != 0;
  }

  // batch form: states flattened as n x 7 doubles (x y z qx qy qz qw)
  std::vector<uint8_t> isValidBatch(const std::vector<double>& se3) const {
    std::vector<uint8_t> v(se3.size() / 7);
    throwOnError(gpu_->get(), artp_validate_states(gpu_->get(), se3.data(), v.size(), v.data(), nullptr),
                 "artp_validate_states");
    return v;
  }

 private:
  ParamsConstPtr params_;
  GpuContextPtr gpu_;
  std::shared_ptr<Map> map_;
  bool has_field_{false};
};

// ob::MotionValidator over the batched DiscreteMotionValidator kernel.  The reference uses OMPL's
// default DiscreteMotionValidator (si_->checkMotion at prm_motion_cost.cpp:652,
// lazy_prm_star_min_update.cpp:725); z bounds follow Planner::setMap (planner.cpp:146-156).
class BatchMotionValidator : public ob::MotionValidator {
 public:
  BatchMotionValidator(const ob::SpaceInformationPtr& si, const GpuContextPtr& gpu)
      : ob::MotionValidator(si), gpu_(gpu) {}

  void setZBounds(double low, double high) {
    throwOnError(gpu_->get(), artp_set_z_bounds(gpu_->get(), low, high), "artp_set_z_bounds");
  }

  bool checkMotion(const ob::State* s1, const ob::State* s2) const override {
    double a[7], b[7];
    flattenSE3(s1, a);
    flattenSE3(s2, b);
    uint8_t v = 0;
    throwOnError(gpu_->get(), artp_check_motions(gpu_->get(), a, b, 1, &v), "artp_check_motions");
    return v != 0;
  }

  std::vector<uint8_t> checkMotionBatch(const std::vector<double>& s1, const std::vector<double>& s2) const {
    std::vector<uint8_t> v(s1.size() / 7);
    throwOnError(gpu_->get(), artp_check_motions(gpu_->get(), s1.data(), s2.data(), v.size(), v.data()),
                 "artp_check_motions");
    return v;
  }

 private:
  GpuContextPtr gpu_;
};

}  // namespace art_planner
