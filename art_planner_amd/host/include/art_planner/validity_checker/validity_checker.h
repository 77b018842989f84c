// art_planner::StateValidityChecker with the reference's interface
// (art_planner/include/art_planner/validity_checker/validity_checker.h:22-47), plus the batch entry the
// new batched planner loops use.  There is no CPU path in this library: isValid() is either a lookup of the label
// the GPU computed when the sampler mirror issued the state (GpuContext::lookupLabel) or a batch of one through
// the latency path of artp_validate_states (one launch, mapped host memory).  Like the reference's, isValid() and
// checkMotion() never throw: a failing call returns false and is recorded in GpuContext::lastError().
//
// Differences from the reference's class, on purpose: the constructors take the GpuContext.
#pragma once

#include <memory>
#include <utility>
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

inline void unflattenSE3(const double s[7], ob::State* state) {
  auto* o = state->as<ob::SE3StateSpace::StateType>();
  o->setXYZ(s[0], s[1], s[2]);
  o->rotation().x = s[3];
  o->rotation().y = s[4];
  o->rotation().z = s[5];
  o->rotation().w = s[6];
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
    gpu_->mapChanged();  // labels of states issued on the previous map are void
    has_field_ = true;
  }

  // the height fields reached the device another way (art_planner::Planner installs the result of the device
  // preprocessing, artp_preprocessed_install): hasMap() is true from here on
  void heightFieldInstalled() { has_field_ = true; }

  bool hasMap() const { return static_cast<bool>(map_) && has_field_; }

  // validity_checker.cpp:39-45
  bool isValid(const ob::State* state) const override {
    double s[7];
    flattenSE3(state, s);
    uint8_t v = 0;
    if (gpu_->lookupLabel(s, &v)) return v != 0;  // a state the sampler mirror issued on this map
    const int rc = artp_validate_states(gpu_->get(), s, 1, &v, nullptr);
    if (rc != ARTP_OK) {
      gpu_->noteError("artp_validate_states", rc);
      return false;
    }
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
    const int rc = artp_check_motions(gpu_->get(), a, b, 1, &v);
    if (rc != ARTP_OK) {
      gpu_->noteError("artp_check_motions", rc);
      v = 0;
    }
    if (v) ++valid_; else ++invalid_;  // DiscreteMotionValidator keeps these counters
    return v != 0;
  }

  // the second pure virtual of ob::MotionValidator: on failure lastValid.second = the parameter of the last
  // valid state on the discretised segment and *lastValid.first (when not null) = that state
  bool checkMotion(const ob::State* s1, const ob::State* s2, std::pair<ob::State*, double>& lastValid) const override {
    double a[7], b[7], t = 0.0, st[7];
    flattenSE3(s1, a);
    flattenSE3(s2, b);
    uint8_t v = 0;
    const int rc = artp_check_motions_last_valid(gpu_->get(), a, b, 1, &v, &t, st);
    if (rc != ARTP_OK) {
      gpu_->noteError("artp_check_motions_last_valid", rc);
      ++invalid_;
      lastValid.second = 0.0;  // nothing beyond s1 is known to be valid
      if (lastValid.first) unflattenSE3(a, lastValid.first);
      return false;
    }
    if (!v) {
      lastValid.second = t;
      if (lastValid.first) unflattenSE3(st, lastValid.first);
    }
    if (v) ++valid_; else ++invalid_;
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
