// TEST SCAFFOLD, not OMPL: ompl::geometric::SimpleSetup as far as art_planner::Planner (planner.cpp:75-131,165-262) and
// PlannerRos (planner_ros.cpp:242-243,309-318,359,373-377) reach into it through the protected member ss_.
#pragma once
#include <memory>
#include "art_planner/ompl_standins_planning.h"
#include "ompl/base/ScopedState.h"
#include "ompl/geometric/PathGeometric.h"
namespace ompl {
namespace geometric {
class SimpleSetup {
 public:
  explicit SimpleSetup(const base::StateSpacePtr& space)
      : si_(std::make_shared<base::SpaceInformation>(space)), pdef_(std::make_shared<base::ProblemDefinition>(si_)) {}
  const base::SpaceInformationPtr& getSpaceInformation() const { return si_; }
  const base::StateSpacePtr& getStateSpace() const { return si_->getStateSpace(); }
  const base::ProblemDefinitionPtr& getProblemDefinition() const { return pdef_; }
  void setStateValidityChecker(const base::StateValidityCheckerPtr& svc) { si_->setStateValidityChecker(svc); }
  void setPlanner(const base::PlannerPtr& planner) {
    planner_ = planner;
    if (planner_) planner_->setProblemDefinition(pdef_);
  }
  const base::PlannerPtr& getPlanner() const { return planner_; }
  void setOptimizationObjective(const base::OptimizationObjectivePtr& o) { pdef_->setOptimizationObjective(o); }
  const base::OptimizationObjectivePtr& getOptimizationObjective() const { return pdef_->getOptimizationObjective(); }
  void setStartState(const base::ScopedState<>& st) {
    pdef_->clearStartStates();
    pdef_->addStartState(st.get());
  }
  void setGoalState(const base::ScopedState<>& st) { pdef_->setGoalState(st.get()); }
  void setGoal(const base::GoalPtr& g) { pdef_->setGoal(g); }
  base::PlannerStatus solve(double time = 1.0) {
    if (!planner_) return base::PlannerStatus::ABORT;
    if (!planner_->isSetup()) planner_->setup();
    return planner_->solve(time);
  }
  bool haveSolutionPath() const { return pdef_->hasSolution(); }
  PathGeometric& getSolutionPath() const { return *pdef_->getSolutionPath()->as<PathGeometric>(); }
  void clear() {   // SimpleSetup::clear: the planner's data structures and the problem's solution paths
    if (planner_) planner_->clear();
    pdef_->clearSolutionPaths();
  }
  void setup() {
    if (planner_ && !planner_->isSetup()) planner_->setup();
  }

 private:
  base::SpaceInformationPtr si_;
  base::ProblemDefinitionPtr pdef_;
  base::PlannerPtr planner_;
};
}  // namespace geometric
}  // namespace ompl
