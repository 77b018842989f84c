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

// The reference's maintainer owns the motion-cost functor (a ROS service client in PlannerRos) and re-prices / re-samples
// the graph between queries (prm_motion_cost.cpp:27-219).  Here the roadmap prices its edges ON THE DEVICE
// (artp_roadmap_params::objective = 2: artp_cost_query inside the build, no host round trip per edge batch), so the functor
// handed over is kept for the caller (motionCostFunction()) but is not in the planning loop; update() / sampleGraph() keep
// the kept roadmap current with the installed map (artp_roadmap_grow: invalidated milestones dropped and replenished).
class PRMMotionCostMaintainer {
 public:
  PRMMotionCostMaintainer(const std::shared_ptr<Map>& map, const ParamsConstPtr& params,
                          std::unique_ptr<MotionCostFunc>&& motion_cost_func)
      : params_(params), map_(map), motion_cost_func_(std::move(motion_cost_func)) {}
  inline void setPlanner(PRMMotionCost* planner) { p_ = planner; }
  void update() { sampleGraph(); }
  inline void sampleGraph();
  const MotionCostFunc* motionCostFunction() const { return motion_cost_func_.get(); }

 private:
  ParamsConstPtr params_;
  PRMMotionCost* p_{nullptr};
  std::shared_ptr<Map> map_;
  std::unique_ptr<MotionCostFunc> motion_cost_func_;
};

class PRMMotionCost : public GpuRoadmapPlanner {
  friend PRMMotionCostMaintainer;

 public:
  explicit PRMMotionCost(const ob::SpaceInformationPtr& si, bool /*starStrategy*/ = false)
      : GpuRoadmapPlanner(si, "PRMMotionCost", false) {}

  void setMaintainer(std::unique_ptr<PRMMotionCostMaintainer>&& maintainer) {
    maintainer_ = std::move(maintainer);
    maintainer_->setPlanner(this);
  }
  bool hasMaintainer() const { return static_cast<bool>(maintainer_); }

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
};

inline void PRMMotionCostMaintainer::sampleGraph() {
  if (!p_) return;
  const std::shared_ptr<RoadmapHandle>& h = p_->roadmapHandle();
  if (!h || !h->built) return;
  try {
    const size_t dropped = h->prm->grow(0);      // milestones the installed map no longer accepts
    if (dropped) h->prm->grow(dropped);          // ... replaced by as many new samples
  } catch (const std::exception&) {              // start / goal themselves became invalid: rebuild at the next query
    h->prm->clear();
    h->built = false;
  }
}

}  // namespace art_planner
