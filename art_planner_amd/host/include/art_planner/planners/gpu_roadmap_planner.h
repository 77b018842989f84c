// The ob::Planner-shaped front of the batched roadmap: what art_planner's own planner classes -- PRMMotionCost
// (include/art_planner/planners/prm_motion_cost.h:39-95), LazyPRMStarMinUpdate (planners/lazy_prm_star_min_update.h:27-75)
// -- present to og::SimpleSetup and to art_planner_ros's PlannerRos (planner_ros.cpp:242-243,309-318,359,373-377), over
// BatchPRM (artp_roadmap_* of the C ABI) instead of og::LazyPRMstar's boost graph.  Needs OMPL's planning layer
// (<ompl/base/Planner.h> ...): compiled with -DARTP_HAVE_OMPL only (tests/fake_include in this image).
#pragma once

#ifndef ARTP_HAVE_OMPL
#error "the ob::Planner shells need OMPL (or the scaffold under tests/fake_include): build with -DARTP_HAVE_OMPL"
#endif

#include <memory>
#include <mutex>
#include <vector>

#include <ompl/base/Planner.h>
#include <ompl/base/PlannerData.h>
#include <ompl/base/PlannerStatus.h>
#include <ompl/base/PlannerTerminationCondition.h>
#include <ompl/base/ProblemDefinition.h>
#include <ompl/base/goals/GoalSampleableRegion.h>
#include <ompl/geometric/PathGeometric.h>

#include "art_planner/planners/batch_prm.h"
#include "art_planner/validity_checker/validity_checker.h"

namespace og = ompl::geometric;

namespace art_planner {

class GpuRoadmapPlanner : public ob::Planner {
 public:
  GpuRoadmapPlanner(const ob::SpaceInformationPtr& si, const std::string& name, bool lazy_star)
      : ob::Planner(si, name), lazy_star_(lazy_star) {}
  ~GpuRoadmapPlanner() override { freeStates(); }

  // The roadmap art_planner::Planner drives (Planner::Planner binds it right after `ss_->setPlanner(planner)`,
  // planner.cpp:107): the shell and Planner::plan() work on the SAME graph.
  void bindRoadmap(const std::shared_ptr<RoadmapHandle>& h) { h_ = h; }
  const std::shared_ptr<RoadmapHandle>& roadmapHandle() const { return h_; }

  // PRMMotionCost::clear / LazyPRMStarMinUpdate::clear (prm_motion_cost.cpp:236-247): the graph goes, the next solve
  // samples a new one
  void clear() override {
    ob::Planner::clear();
    std::lock_guard<std::mutex> lock(mutex_);
    freeStates();
    if (h_) {
      h_->prm->clear();
      h_->built = false;
    }
  }
  void setup() override { ob::Planner::setup(); }
  void clearQuery() override {}   // a query replaces the roadmap's start / goal vertices (artp_roadmap_set_query)

  // solve (prm_motion_cost.cpp:295-303 -> baseSolve :440-532 / lazy_prm_star_min_update.cpp:496-615): start and goal of
  // the problem definition, roadmap built or re-queried, cheapest valid path added to the problem definition.
  ob::PlannerStatus solve(const ob::PlannerTerminationCondition& ptc) override {
    std::lock_guard<std::mutex> lock(mutex_);
    if (!h_ || !pdef_) return ob::PlannerStatus::ABORT;
    if (pdef_->getStartStateCount() == 0) return ob::PlannerStatus::INVALID_START;
    const ob::GoalPtr goal = pdef_->getGoal();
    if (!goal) return ob::PlannerStatus::INVALID_GOAL;
    BatchPRM::StateArray s, g;
    flattenSE3(pdef_->getStartState(0), s.data());
    {
      ob::State* gs = si_->allocState();
      goal->as<ob::GoalSampleableRegion>()->sampleGoal(gs);
      flattenSE3(gs, g.data());
      si_->freeState(gs);
    }
    std::vector<BatchPRM::StateArray> flat;
    double cost = 0.0;
    bool solved = false;
    try {
      if (h_->built) {
        h_->prm->setQuery(s, g);
      } else {
        h_->prm->sampleGraph(s, g);
        h_->built = true;
      }
      if (ptc()) return ob::PlannerStatus::TIMEOUT;
      solved = lazy_star_ ? h_->prm->solveUntil(h_->params->planner.plan_time, 1000, &flat, &cost)
                          : h_->prm->solve(&flat, &cost);
    } catch (const std::exception&) {   // start or goal rejected by the device, or every path edge invalid
      return ob::PlannerStatus::ABORT;
    }
    if (!solved) return ob::PlannerStatus::TIMEOUT;
    auto path = std::make_shared<og::PathGeometric>(si_);
    ob::State* st = si_->allocState();
    for (const BatchPRM::StateArray& f : flat) {
      unflattenSE3(f.data(), st);
      path->append(st);
    }
    si_->freeState(st);
    pdef_->addSolutionPath(path, false, 0.0, getName());
    last_cost_ = cost;
    return ob::PlannerStatus::EXACT_SOLUTION;
  }
  double lastSolutionCost() const { return last_cost_; }

  // Planner data as PlannerRos' visualiser reads it (planner_ros.cpp:242-243; prm_motion_cost.cpp:259-292,
  // lazy_prm_star_min_update.cpp getPlannerData: validated edges only): start / goal vertices tagged 1, every edge in both
  // directions (an undirected roadmap), weight = the edge's cost.  The states live in this planner until the next
  // export / clear, like the vertices of OMPL's planners do.
  void exportPlannerData(ob::PlannerData& data, bool get_invalid) const {
    std::lock_guard<std::mutex> lock(mutex_);
    freeStates();
    if (!h_ || !h_->built) return;
    const size_t nv = h_->prm->numVertices();
    std::vector<double> verts = h_->prm->vertices();
    std::vector<uint32_t> uv;
    std::vector<uint8_t> valid, removed;
    std::vector<double> cost;
    h_->prm->edges(&uv, &valid, &removed, &cost);
    states_.reserve(nv);
    for (size_t i = 0; i < nv; ++i) {
      ob::State* st = si_->allocState();
      unflattenSE3(&verts[7 * i], st);
      states_.push_back(st);
    }
    if (nv >= 2) {
      data.addStartVertex(ob::PlannerDataVertex(states_[0], 1));
      data.addGoalVertex(ob::PlannerDataVertex(states_[1], 1));
    }
    for (size_t e = 0; e < valid.size(); ++e) {
      const bool ok = valid[e] && !removed[e];
      if (!ok && !get_invalid) continue;
      const uint32_t u = uv[2 * e], v = uv[2 * e + 1];
      const unsigned int iu = data.addVertex(ob::PlannerDataVertex(states_[u], 1));
      const unsigned int iv = data.addVertex(ob::PlannerDataVertex(states_[v], 1));
      data.addEdge(iu, iv, ob::PlannerDataEdge(), ob::Cost(cost[e]));
      data.addEdge(iv, iu, ob::PlannerDataEdge(), ob::Cost(cost[e]));
    }
  }

 protected:
  void freeStates() const {
    for (ob::State* s : states_) si_->freeState(s);
    states_.clear();
  }
  std::shared_ptr<RoadmapHandle> h_;
  bool lazy_star_;
  double last_cost_{0.0};
  mutable std::mutex mutex_;
  mutable std::vector<ob::State*> states_;
};

}  // namespace art_planner
