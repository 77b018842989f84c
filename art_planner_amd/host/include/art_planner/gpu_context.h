// One artp_ctx (one MI355X) shared by the checker, sampler, motion validator and cost objective of a
// Planner -- the role art_planner::Planner's members play in the reference (planner.h:40-52).
#pragma once

#include <memory>
#include <stdexcept>
#include <string>

#include "art_planner/params.h"
#include "artp_c.h"

namespace art_planner {

class GpuContext {
 public:
  explicit GpuContext(const ParamsConstPtr& params, int device = 0) {
    artp_params p{};
    p.torso_length = params->robot.torso.length;
    p.torso_width = params->robot.torso.width;
    p.torso_height = params->robot.torso.height;
    p.torso_off_x = params->robot.torso.offset.x;
    p.torso_off_y = params->robot.torso.offset.y;
    p.torso_off_z = params->robot.torso.offset.z;
    p.feet_off_x = params->robot.feet.offset.x;
    p.feet_off_y = params->robot.feet.offset.y;
    p.feet_off_z = params->robot.feet.offset.z;
    p.reach_x = params->robot.feet.reach.x;
    p.reach_y = params->robot.feet.reach.y;
    p.reach_z = params->robot.feet.reach.z;
    p.unknown_space_untraversable = params->planner.unknown_space_untraversable ? 1 : 0;
    p.max_pitch_pert = params->sampler.max_pitch_pert;
    p.max_roll_pert = params->sampler.max_roll_pert;
    p.sample_from_distribution = params->sampler.sample_from_distribution ? 1 : 0;
    const int rc = artp_create(device, &p, &ctx_);
    if (rc != ARTP_OK) throw std::runtime_error(std::string("artp_create: ") + artp_status_string(rc));
  }
  ~GpuContext() { artp_destroy(ctx_); }
  GpuContext(const GpuContext&) = delete;
  GpuContext& operator=(const GpuContext&) = delete;
  artp_ctx* get() const { return ctx_; }

 private:
  artp_ctx* ctx_{nullptr};
};

using GpuContextPtr = std::shared_ptr<GpuContext>;

inline void throwOnError(artp_ctx* ctx, int rc, const char* what) {
  if (rc != ARTP_OK)
    throw std::runtime_error(std::string(what) + ": " + artp_status_string(rc) + " " + artp_last_error(ctx));
}

}  // namespace art_planner
