// art_planner::PRMMotionCost / PRMMotionCostMaintainer with the reference's public members
// (art_planner/include/art_planner/planners/prm_motion_cost.h:39-160), over the batched roadmap on the MI355X.
// What PlannerRos does with them (planner_ros.cpp): ss_->getPlanner()->as<PRMMotionCost>()->getPlannerData(dat, get_invalid)
// (:243), ->setMaintainer(std::unique_ptr<PRMMotionCostMaintainer>(new PRMMotionCostMaintainer(map_, params_, cost_func)))
// (:309-312), ->sampleGraph() (:376-377), and ss_->clear() / ss_->setup() / ss_->solve() through og::SimpleSetup.
#pragma once

#include <functional>
#include <memory>

#include "art_planner/map/map.h"
#include "art_planner/objectives/motion_cost_objective.h"
#include "art_planner/params.h"
#include "art_planner/planners/gpu_roadmap_planner.h"

namespace art_planner {

class PRMMotionCost;

// The reference's maintainer owns the motion-cost functor (a ROS service client in PlannerRos) and prices the graph's edges
// THROUGH it: one batched call, edge matrix [B x 6] in, costs [B x 3] out (updateEdges, prm_motion_cost.cpp:27-73).  So does
// this one: while a maintainer is installed, every learned-cost batch of the roadmap (build, grow, re-query, path
// simplification) leaves the device, goes through the functor and comes back (artp_cost_set_external_query) -- a
// PlannerRos that hands in its service-client functor gets ITS cost source.  A functor that returns false fails the
// planning call with std::runtime_error("Motion cost call failed") (motion_cost_objective.cpp:78-83; the reference's
// maintainer ignores the failure -- "TODO: Catch error here", :208 -- and plans on stale weights).
// setDevicePricing(true) is the explicit opt-in to the other source: the network on the device (artp_cost_query_dev
// inside the build, no host round trip; needs artp_cost_load_weights + artp_cost_update_map), the functor kept for the
// caller (motionCostFunction()) but out of the planning loop.
// update() / sampleGraph() keep the kept roadmap current with the installed map (artp_roadmap_grow: invalidated
// milestones dropped and replenished).
class PRMMotionCostMaintainer {
 public:
  PRMMotionCostMaintainer(const std::shared_ptr<Map>& map, const ParamsConstPtr& params,
                          std::unique_ptr<MotionCostFunc>&& motion_cost_func)
      : params_(params), map_(map), motion_cost_func_(std::move(motion_cost_func)) {}
  ~PRMMotionCostMaintainer() { uninstall(); }
  PRMMotionCostMaintainer(const PRMMotionCostMaintainer&) = delete;
  PRMMotionCostMaintainer& operator=(const PRMMotionCostMaintainer&) = delete;
  inline void setPlanner(PRMMotionCost* planner);
  void update() { sampleGraph(); }
  inline void sampleGraph();
  const MotionCostFunc* motionCostFunction() const { return motion_cost_func_.get(); }

  // false (default): the roadmap's edges are priced through the functor; true: on the device
  void setDevicePricing(bool on) {
    device_pricing_ = on;
    install();
  }
  bool devicePricing() const { return device_pricing_; }
  size_t functorCalls() const { return functor_calls_; }
  size_t functorEdges() const { return functor_edges_; }

 private:
  // artp_cost_query_fn: the roadmap's edge matrix through the functor (host buffers in artp_cost_query's layout)
  static int costQueryThunk(void* user, const float* edges, size_t b, float* cost) {
    auto* self = static_cast<PRMMotionCostMaintainer*>(user);
    if (!self || !self->motion_cost_func_ || !*self->motion_cost_func_) return 1;
    EdgeMatrix em(static_cast<long>(b), 6), ec(static_cast<long>(b), 3);
    for (size_t i = 0; i < b; ++i)
      for (int k = 0; k < 6; ++k) em(static_cast<long>(i), k) = edges[6 * i + k];
    ++self->functor_calls_;
    self->functor_edges_ += b;
    bool ok = false;
    try {
      ok = (*self->motion_cost_func_)(em, &ec);
    } catch (...) {   // an exception must not cross the C boundary: reported as the failed call it is
      ok = false;
    }
    if (!ok) return 1;
    for (size_t i = 0; i < b; ++i)
      for (int k = 0; k < 3; ++k) cost[3 * i + k] = ec(static_cast<long>(i), k);
    return 0;
  }
  inline void install();
  inline void uninstall();

  ParamsConstPtr params_;
  PRMMotionCost* p_{nullptr};
  std::shared_ptr<Map> map_;
  std::unique_ptr<MotionCostFunc> motion_cost_func_;
  GpuContextPtr installed_on_;     // the context whose roadmap batches come here
  bool device_pricing_{false};
  size_t functor_calls_{0}, functor_edges_{0};
};

class PRMMotionCost : public GpuRoadmapPlanner {
  friend PRMMotionCostMaintainer;

 public:
  explicit PRMMotionCost(const ob::SpaceInformationPtr& si, bool /*starStrategy*/ = false)
      : GpuRoadmapPlanner(si, "PRMMotionCost", false) {}

  void setMaintainer(std::unique_ptr<PRMMotionCostMaintainer>&& maintainer) {
    maintainer_ = std::move(maintainer);   // (the old one takes its functor off the context as it goes)
    if (device_pricing_) maintainer_->setDevicePricing(true);
    maintainer_->setPlanner(this);
  }
  // Explicit opt-in (Planner::setDevicePricing): price the roadmap's edges with the network on the device instead of
  // through the maintainer's functor.  Applies to the current maintainer and to any set later.
  void setDevicePricing(bool on) {
    device_pricing_ = on;
    if (maintainer_) maintainer_->setDevicePricing(on);
  }
  bool devicePricing() const { return device_pricing_; }
  bool hasMaintainer() const { return static_cast<bool>(maintainer_); }
  PRMMotionCostMaintainer* maintainer() const { return maintainer_.get(); }

  void getPlannerData(ob::PlannerData& data, bool get_invalid) const { exportPlannerData(data, get_invalid); }
  void getPlannerData(ob::PlannerData& data) const override { exportPlannerData(data, false); }

  // prm_motion_cost.cpp:295-303: the maintainer tops the graph up, then the search
  ob::PlannerStatus solve(const ob::PlannerTerminationCondition& ptc) override {
    if (maintainer_) maintainer_->sampleGraph();
    return GpuRoadmapPlanner::solve(ptc);
  }

  // prm_motion_cost.cpp:677-679.  The batched roadmap is built around a query (start and goal are vertices 0 and 1):
  // before the first query there is nothing to sample into, the build happens in solve() / Planner::plan().
  void sampleGraph() {
    if (maintainer_) maintainer_->sampleGraph();
  }

 private:
  std::unique_ptr<PRMMotionCostMaintainer> maintainer_;
  bool device_pricing_{false};
};

inline void PRMMotionCostMaintainer::setPlanner(PRMMotionCost* planner) {
  p_ = planner;
  install();
}

inline void PRMMotionCostMaintainer::install() {
  uninstall();
  if (device_pricing_ || !p_ || !motion_cost_func_) return;
  const std::shared_ptr<RoadmapHandle>& h = p_->roadmapHandle();
  if (!h || !h->prm) return;
  installed_on_ = h->prm->gpu();
  throwOnError(installed_on_->get(), artp_cost_set_external_query(installed_on_->get(), &PRMMotionCostMaintainer::costQueryThunk, this),
               "artp_cost_set_external_query");
}

inline void PRMMotionCostMaintainer::uninstall() {
  if (!installed_on_) return;
  (void)artp_cost_set_external_query(installed_on_->get(), nullptr, nullptr);
  installed_on_.reset();
}

inline void PRMMotionCostMaintainer::sampleGraph() {
  if (!p_) return;
  const std::shared_ptr<RoadmapHandle>& h = p_->roadmapHandle();
  if (!h || !h->built) return;
  try {
    const size_t dropped = h->prm->grow(0);      // milestones the installed map no longer accepts
    if (dropped) h->prm->grow(dropped);          // ... replaced by as many new samples
  } catch (const MotionCostCallFailed&) {
    throw;
  } catch (const std::exception&) {              // start / goal themselves became invalid: rebuild at the next query
    h->prm->clear();
    h->built = false;
  }
}

}  // namespace art_planner
