// OMPL-1.4.2-shaped stand-ins of the PLANNING layer -- ompl::base::Planner, PlannerData, PlannerStatus,
// PlannerTerminationCondition, ProblemDefinition, OptimizationObjective, Goal / GoalSampleableRegion / GoalState, Path --
// as far as art_planner's planner classes (planners/prm_motion_cost.h, planners/lazy_prm_star_min_update.h,
// objectives/motion_cost_objective.h) and art_planner_ros's PlannerRos (planner_ros.cpp:242-243,309-318,359,373-377)
// use them.  tests/fake_include presents them under the real include names; nothing of OMPL is executed here.
// Written from the published OMPL 1.4.2 headers (ompl/base/Planner.h, PlannerData.h, PlannerStatus.h,
// PlannerTerminationCondition.h, ProblemDefinition.h, OptimizationObjective.h, Cost.h, Goal.h, goals/GoalState.h, Path.h).
#pragma once

#include <chrono>
#include <limits>
#include <functional>
#include <memory>
#include <string>
#include <vector>

#include "art_planner/ompl_standins.h"

namespace ompl {
namespace base {

// ompl/base/Cost.h
class Cost {
 public:
  Cost() = default;
  explicit Cost(double v) : v_(v) {}
  double value() const { return v_; }

 private:
  double v_{0.0};
};

// ompl/base/PlannerStatus.h
struct PlannerStatus {
  enum StatusType { UNKNOWN = 0, INVALID_START, INVALID_GOAL, UNRECOGNIZED_GOAL_TYPE, TIMEOUT, APPROXIMATE_SOLUTION,
                    EXACT_SOLUTION, CRASH, ABORT, TYPE_COUNT };
  PlannerStatus(StatusType s = UNKNOWN) : status_(s) {}
  operator StatusType() const { return status_; }
  explicit operator bool() const { return status_ == APPROXIMATE_SOLUTION || status_ == EXACT_SOLUTION; }

 private:
  StatusType status_;
};

// ompl/base/PlannerTerminationCondition.h
class PlannerTerminationCondition {
 public:
  explicit PlannerTerminationCondition(std::function<bool()> fn) : fn_(std::move(fn)) {}
  bool operator()() const { return fn_(); }
  bool eval() const { return fn_(); }

 private:
  std::function<bool()> fn_;
};
inline PlannerTerminationCondition timedPlannerTerminationCondition(double seconds) {
  const auto end = std::chrono::steady_clock::now() + std::chrono::duration<double>(seconds);
  return PlannerTerminationCondition([end]() { return std::chrono::steady_clock::now() >= end; });
}
inline PlannerTerminationCondition plannerNonTerminatingCondition() {
  return PlannerTerminationCondition([]() { return false; });
}

// ompl/base/Path.h
class Path {
 public:
  explicit Path(SpaceInformationPtr si) : si_(std::move(si)) {}
  virtual ~Path() = default;
  template <class T>
  T* as() { return static_cast<T*>(this); }
  template <class T>
  const T* as() const { return static_cast<const T*>(this); }
  const SpaceInformationPtr& getSpaceInformation() const { return si_; }

 protected:
  SpaceInformationPtr si_;
};
using PathPtr = std::shared_ptr<Path>;

// ompl/base/Goal.h, goals/GoalSampleableRegion.h, goals/GoalState.h
class Goal {
 public:
  explicit Goal(SpaceInformationPtr si) : si_(std::move(si)) {}
  virtual ~Goal() = default;
  template <class T>
  T* as() { return static_cast<T*>(this); }
  template <class T>
  const T* as() const { return static_cast<const T*>(this); }

 protected:
  SpaceInformationPtr si_;
};
using GoalPtr = std::shared_ptr<Goal>;
class GoalSampleableRegion : public Goal {
 public:
  using Goal::Goal;
  virtual void sampleGoal(State* st) const = 0;
  virtual unsigned int maxSampleCount() const = 0;
};
class GoalState : public GoalSampleableRegion {
 public:
  explicit GoalState(const SpaceInformationPtr& si) : GoalSampleableRegion(si), state_(si->allocState()) {}
  ~GoalState() override { si_->freeState(state_); }
  void setState(const State* st) { si_->copyState(state_, st); }
  const State* getState() const { return state_; }
  void sampleGoal(State* st) const override { si_->copyState(st, state_); }
  unsigned int maxSampleCount() const override { return 1; }

 private:
  State* state_;
};

// ompl/base/OptimizationObjective.h
class OptimizationObjective {
 public:
  explicit OptimizationObjective(SpaceInformationPtr si) : si_(std::move(si)) {}
  virtual ~OptimizationObjective() = default;
  virtual Cost stateCost(const State* s) const = 0;
  virtual Cost motionCost(const State* s1, const State* s2) const = 0;
  virtual Cost motionCostHeuristic(const State*, const State*) const { return Cost(0.0); }
  virtual bool isCostBetterThan(Cost c1, Cost c2) const { return c1.value() < c2.value(); }
  virtual Cost combineCosts(Cost c1, Cost c2) const { return Cost(c1.value() + c2.value()); }
  virtual Cost identityCost() const { return Cost(0.0); }
  virtual Cost infiniteCost() const { return Cost(std::numeric_limits<double>::infinity()); }
  const SpaceInformationPtr& getSpaceInformation() const { return si_; }

 protected:
  SpaceInformationPtr si_;
};
using OptimizationObjectivePtr = std::shared_ptr<OptimizationObjective>;

// ompl/base/ProblemDefinition.h
class ProblemDefinition {
 public:
  explicit ProblemDefinition(SpaceInformationPtr si) : si_(std::move(si)) {}
  ~ProblemDefinition() { clearStartStates(); }
  void addStartState(const State* st) {
    State* s = si_->allocState();
    si_->copyState(s, st);
    starts_.push_back(s);
  }
  void clearStartStates() {
    for (State* s : starts_) si_->freeState(s);
    starts_.clear();
  }
  unsigned int getStartStateCount() const { return static_cast<unsigned int>(starts_.size()); }
  const State* getStartState(unsigned int i) const { return starts_[i]; }
  void setGoal(const GoalPtr& g) { goal_ = g; }
  void setGoalState(const State* st) {
    auto g = std::make_shared<GoalState>(si_);
    g->setState(st);
    goal_ = g;
  }
  const GoalPtr& getGoal() const { return goal_; }
  void setOptimizationObjective(const OptimizationObjectivePtr& o) { obj_ = o; }
  const OptimizationObjectivePtr& getOptimizationObjective() const { return obj_; }
  bool hasOptimizationObjective() const { return static_cast<bool>(obj_); }
  void addSolutionPath(const PathPtr& path, bool /*approximate*/ = false, double /*difference*/ = -1.0,
                       const std::string& /*plannerName*/ = "Unknown") {
    solutions_.push_back(path);
  }
  bool hasSolution() const { return !solutions_.empty(); }
  PathPtr getSolutionPath() const { return solutions_.empty() ? PathPtr() : solutions_.back(); }
  void clearSolutionPaths() { solutions_.clear(); }

 private:
  SpaceInformationPtr si_;
  std::vector<State*> starts_;
  GoalPtr goal_;
  OptimizationObjectivePtr obj_;
  std::vector<PathPtr> solutions_;
};
using ProblemDefinitionPtr = std::shared_ptr<ProblemDefinition>;

// ompl/base/PlannerData.h (states are referenced, not copied: the planner that filled it keeps them alive, as in OMPL)
class PlannerDataVertex {
 public:
  PlannerDataVertex(const State* st, int tag = 0) : state_(st), tag_(tag) {}
  virtual ~PlannerDataVertex() = default;
  const State* getState() const { return state_; }
  int getTag() const { return tag_; }
  void setTag(int tag) { tag_ = tag; }

 private:
  const State* state_;
  int tag_;
};
class PlannerDataEdge {
 public:
  virtual ~PlannerDataEdge() = default;
};
class PlannerData {
 public:
  static const unsigned int INVALID_INDEX = 0xffffffffu;
  explicit PlannerData(SpaceInformationPtr si) : si_(std::move(si)) {}
  unsigned int addVertex(const PlannerDataVertex& v) {
    const unsigned int at = vertexIndex(v);
    if (at != INVALID_INDEX) return at;
    vertices_.push_back(v);
    return static_cast<unsigned int>(vertices_.size() - 1);
  }
  unsigned int addStartVertex(const PlannerDataVertex& v) {
    const unsigned int i = addVertex(v);
    starts_.push_back(i);
    return i;
  }
  unsigned int addGoalVertex(const PlannerDataVertex& v) {
    const unsigned int i = addVertex(v);
    goals_.push_back(i);
    return i;
  }
  bool addEdge(unsigned int v1, unsigned int v2, const PlannerDataEdge& = PlannerDataEdge(), Cost weight = Cost(1.0)) {
    if (v1 >= vertices_.size() || v2 >= vertices_.size()) return false;
    edges_.push_back({v1, v2, weight.value()});
    return true;
  }
  bool addEdge(const PlannerDataVertex& v1, const PlannerDataVertex& v2, const PlannerDataEdge& e = PlannerDataEdge(),
               Cost weight = Cost(1.0)) {
    return addEdge(addVertex(v1), addVertex(v2), e, weight);
  }
  bool tagState(const State* st, int tag) {
    for (PlannerDataVertex& v : vertices_)
      if (v.getState() == st) {
        v.setTag(tag);
        return true;
      }
    return false;
  }
  unsigned int vertexIndex(const PlannerDataVertex& v) const {
    for (size_t i = 0; i < vertices_.size(); ++i)
      if (vertices_[i].getState() == v.getState()) return static_cast<unsigned int>(i);
    return INVALID_INDEX;
  }
  unsigned int numVertices() const { return static_cast<unsigned int>(vertices_.size()); }
  unsigned int numEdges() const { return static_cast<unsigned int>(edges_.size()); }
  unsigned int numStartVertices() const { return static_cast<unsigned int>(starts_.size()); }
  unsigned int numGoalVertices() const { return static_cast<unsigned int>(goals_.size()); }
  const PlannerDataVertex& getVertex(unsigned int i) const { return vertices_[i]; }
  struct EdgeRecord { unsigned int v1, v2; double weight; };
  const std::vector<EdgeRecord>& edgeRecords() const { return edges_; }   // (stand-in only: OMPL walks a boost graph)
  const SpaceInformationPtr& getSpaceInformation() const { return si_; }

 private:
  SpaceInformationPtr si_;
  std::vector<PlannerDataVertex> vertices_;
  std::vector<unsigned int> starts_, goals_;
  std::vector<EdgeRecord> edges_;
};

// ompl/base/Planner.h
class Planner {
 public:
  Planner(SpaceInformationPtr si, std::string name) : si_(std::move(si)), name_(std::move(name)) {}
  virtual ~Planner() = default;
  Planner(const Planner&) = delete;
  Planner& operator=(const Planner&) = delete;
  template <class T>
  T* as() { return static_cast<T*>(this); }
  template <class T>
  const T* as() const { return static_cast<const T*>(this); }
  const SpaceInformationPtr& getSpaceInformation() const { return si_; }
  const ProblemDefinitionPtr& getProblemDefinition() const { return pdef_; }
  virtual void setProblemDefinition(const ProblemDefinitionPtr& pdef) { pdef_ = pdef; }
  virtual PlannerStatus solve(const PlannerTerminationCondition& ptc) = 0;
  PlannerStatus solve(double solveTime) { return solve(timedPlannerTerminationCondition(solveTime)); }
  virtual void clear() {}
  virtual void clearQuery() {}
  virtual void getPlannerData(PlannerData&) const {}
  virtual void setup() { setup_ = true; }
  bool isSetup() const { return setup_; }
  const std::string& getName() const { return name_; }

 protected:
  SpaceInformationPtr si_;
  ProblemDefinitionPtr pdef_;
  std::string name_;
  bool setup_{false};
};
using PlannerPtr = std::shared_ptr<Planner>;

}  // namespace base
}  // namespace ompl
