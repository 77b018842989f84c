// art_planner::LazyPRMStarMinUpdate / LazyPRMStarMinUpdateMaintainer with the reference's public members
// (art_planner/include/art_planner/planners/lazy_prm_star_min_update.h:27-120), over the batched roadmap on the MI355X:
// the planner Planner::Planner installs for params.planner.name == "lazy_prm_star_min_update" (planner.cpp:98-99,112-117).
#pragma once

#include <memory>

#include "art_planner/map/map.h"
#include "art_planner/params.h"
#include "art_planner/planners/gpu_roadmap_planner.h"

namespace art_planner {

class LazyPRMStarMinUpdate;

// The reference's maintainer invalidates the graph components inside the map's "updated" layer and re-validates edges on
// a background thread (lazy_prm_star_min_update.cpp:18-217); the batched roadmap re-validates EVERYTHING in one device
// pass (artp_roadmap_revalidate, 0.7 ms at 10^4 vertices), which update() runs.
class LazyPRMStarMinUpdateMaintainer {
 public:
  LazyPRMStarMinUpdateMaintainer(const std::shared_ptr<Map>& map, const ParamsConstPtr& params) : params_(params), map_(map) {}
  inline void setPlanner(LazyPRMStarMinUpdate* planner) { p_ = planner; }
  inline void update();

 private:
  ParamsConstPtr params_;
  LazyPRMStarMinUpdate* p_{nullptr};
  std::shared_ptr<Map> map_;
};

class LazyPRMStarMinUpdate : public GpuRoadmapPlanner {
  friend LazyPRMStarMinUpdateMaintainer;

 public:
  explicit LazyPRMStarMinUpdate(const ob::SpaceInformationPtr& si, bool /*starStrategy*/ = true)
      : GpuRoadmapPlanner(si, "LazyPRMStarMinUpdate", true) {}

  void setMaintainer(std::unique_ptr<LazyPRMStarMinUpdateMaintainer>&& maintainer) {
    maintainer_ = std::move(maintainer);
    maintainer_->setPlanner(this);
  }
  // "Contrary to the default version, this only returns validated edges." (lazy_prm_star_min_update.h:62-63)
  void getPlannerData(ob::PlannerData& data) const override { exportPlannerData(data, false); }

  // lazy_prm_star_min_update.cpp solve(): maintainer update, then baseSolve
  ob::PlannerStatus solve(const ob::PlannerTerminationCondition& ptc) override {
    if (maintainer_) maintainer_->update();
    return GpuRoadmapPlanner::solve(ptc);
  }

 private:
  std::unique_ptr<LazyPRMStarMinUpdateMaintainer> maintainer_;
};

inline void LazyPRMStarMinUpdateMaintainer::update() {
  if (!p_) return;
  const std::shared_ptr<RoadmapHandle>& h = p_->roadmapHandle();
  if (!h || !h->built) return;
  try {
    if (!h->prm->revalidate()) {   // start or goal no longer valid: the next query sets new ones
    }
  } catch (const std::exception&) {
    h->prm->clear();
    h->built = false;
  }
}

}  // namespace art_planner
