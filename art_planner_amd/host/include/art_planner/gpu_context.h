// One artp_ctx (one MI355X) shared by the checker, sampler, motion validator and cost objective of a
// Planner -- the role art_planner::Planner's members play in the reference (planner.h:40-52).
#pragma once

#include <atomic>
#include <cstdint>
#include <cstring>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "art_planner/params.h"
#include "artp_c.h"

namespace art_planner {

// Labels of states the sampler mirror issued, computed by the SAME validity kernels on the SAME map in the same
// launch that sampled them (artp_sample_and_validate).  The planners' rejection loops call
// `do sampler_->sampleUniform(s); while (!si_->isValid(s))` (prm_motion_cost.cpp:174-186,
// lazy_prm_star_min_update.cpp:552-554): with the block pre-validated, that isValid() is a lookup instead of a
// kernel launch.  A block carries the artp_map_version() it was validated on and is only ever served while the
// context still reports that version: whichever entry point changed the map (a mirror class, artp_update_layer_rect or
// artp_preprocessed_install straight through GpuContext::get()), the very next isValid() sees the new map -- the
// guarantee of StateValidityChecker::updateHeightField (validity_checker.cpp:26-29).
class ValidatedStateBlock {
 public:
  ValidatedStateBlock(std::vector<double>&& se3, std::vector<uint8_t>&& labels, uint64_t epoch, bool validated)
      : se3_(std::move(se3)), labels_(std::move(labels)), epoch_(epoch), validated_(validated) {
    if (!validated_) return;  // plain states: no lookups
    size_t cap = 16;
    while (cap < 4 * labels_.size()) cap <<= 1;
    table_.assign(cap, 0xffffffffu);
    for (uint32_t i = 0; i < labels_.size(); ++i) {
      size_t slot = hash(state(i)) & (cap - 1);
      while (table_[slot] != 0xffffffffu) slot = (slot + 1) & (cap - 1);
      table_[slot] = i;
    }
  }
  size_t size() const { return labels_.size(); }
  uint64_t epoch() const { return epoch_; }
  bool validated() const { return validated_; }
  const double* state(size_t i) const { return se3_.data() + 7 * i; }
  uint8_t label(size_t i) const { return labels_[i]; }
  // the state last handed out is the one the next isValid() most likely asks about
  void setHint(size_t i) { hint_.store(i, std::memory_order_relaxed); }
  bool lookup(const double s[7], uint8_t* label) const {
    if (!validated_) return false;
    const size_t hint = hint_.load(std::memory_order_relaxed);
    if (hint < labels_.size() && same(state(hint), s)) {
      *label = labels_[hint];
      return true;
    }
    const size_t mask = table_.size() - 1;
    for (size_t slot = hash(s) & mask; table_[slot] != 0xffffffffu; slot = (slot + 1) & mask)
      if (same(state(table_[slot]), s)) {
        *label = labels_[table_[slot]];
        return true;
      }
    return false;
  }

 private:
  static bool same(const double* a, const double* b) { return std::memcmp(a, b, 7 * sizeof(double)) == 0; }
  static size_t hash(const double* s) {
    uint64_t h = 0x9e3779b97f4a7c15ull;
    for (int k = 0; k < 7; ++k) {
      uint64_t u;
      std::memcpy(&u, s + k, 8);
      h = (h ^ u) * 0xbf58476d1ce4e5b9ull;
      h ^= h >> 29;
    }
    return static_cast<size_t>(h);
  }
  std::vector<double> se3_;
  std::vector<uint8_t> labels_;
  std::vector<uint32_t> table_;
  uint64_t epoch_;
  bool validated_;
  std::atomic<size_t> hint_{0};
};

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

  // isValid() on an arbitrary state and checkMotion() on one edge WITHOUT a kernel launch per call: resident workgroups poll
  // for requests (artp_set_persistent_latency).  Off by default: while they are resident (until 200 us after the last
  // call) hipDeviceSynchronize / hipFree of the process wait for them, and up to half the CUs are theirs.
  void setPersistentLatency(bool on) {
    const int rc = artp_set_persistent_latency(ctx_, on ? 1 : 0);
    if (rc != ARTP_OK) throw std::runtime_error(std::string("artp_set_persistent_latency: ") + artp_last_error(ctx_));
  }

  // ---- validated-state blocks (see ValidatedStateBlock) ----
  // The epoch of a block is the map version INSIDE the artp_ctx (artp_map_version): every upload, rectangle update,
  // install or sampler re-weighting bumps it, whoever issued it.  mapChanged() only drops the dead blocks early.
  void mapChanged() {
    std::lock_guard<std::mutex> lock(blocks_mutex_);
    for (auto& b : blocks_) b.reset();
  }
  uint64_t mapEpoch() const { return artp_map_version(ctx_); }
  // a sampler publishes its current block in its own slot (slot = -1: take a free one); returns the slot
  int publishBlock(int slot, const std::shared_ptr<ValidatedStateBlock>& block) {
    std::lock_guard<std::mutex> lock(blocks_mutex_);
    if (slot < 0) slot = next_slot_++ % kBlockSlots;
    blocks_[slot] = (block->validated() && block->epoch() == artp_map_version(ctx_)) ? block : nullptr;
    return slot;
  }
  bool lookupLabel(const double s[7], uint8_t* label) const {
    std::lock_guard<std::mutex> lock(blocks_mutex_);
    const uint64_t now = artp_map_version(ctx_);
    for (const auto& b : blocks_)
      if (b && b->epoch() == now && b->lookup(s, label)) return true;
    return false;
  }
  // isValid() / checkMotion() never throw (ob::StateValidityChecker contract): a failing call is remembered here
  void noteError(const char* what, int rc) const {
    std::lock_guard<std::mutex> lock(blocks_mutex_);
    last_error_ = std::string(what) + ": " + artp_status_string(rc) + " " + artp_last_error(ctx_);
    ++n_errors_;
  }
  std::string lastError() const {
    std::lock_guard<std::mutex> lock(blocks_mutex_);
    return last_error_;
  }
  size_t errorCount() const {
    std::lock_guard<std::mutex> lock(blocks_mutex_);
    return n_errors_;
  }

 private:
  static constexpr int kBlockSlots = 4;
  artp_ctx* ctx_{nullptr};
  mutable std::mutex blocks_mutex_;
  std::shared_ptr<ValidatedStateBlock> blocks_[kBlockSlots];
  int next_slot_{0};
  mutable std::string last_error_;
  mutable size_t n_errors_{0};
};

using GpuContextPtr = std::shared_ptr<GpuContext>;

// The caller's MotionCostFunc reported failure: the reference's own exception and text (motion_cost_objective.cpp:78-83).
// Its own type so that the "rebuild / NOT_SOLVED" catch blocks of the mirror let it through, as the reference's
// `catch (ompl::Exception&)` (planner.cpp:247) does.
struct MotionCostCallFailed : std::runtime_error {
  MotionCostCallFailed() : std::runtime_error("Motion cost call failed") {}
};

inline void throwOnError(artp_ctx* ctx, int rc, const char* what) {
  if (rc == ARTP_ERR_COST_FUNC) throw MotionCostCallFailed();
  if (rc != ARTP_OK)
    throw std::runtime_error(std::string(what) + ": " + artp_status_string(rc) + " " + artp_last_error(ctx));
}

}  // namespace art_planner
