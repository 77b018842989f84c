// TEST SCAFFOLD, not OMPL: ompl::geometric::SimpleSetup as far as art_planner::Planner (planner.cpp:75-131) and
// PlannerRos (planner_ros.cpp:242,313,359,373-374) reach into it through the protected member ss_.
#pragma once
#include <memory>
#include "art_planner/ompl_standins.h"
namespace ompl {
namespace geometric {
class SimpleSetup {
 public:
  explicit SimpleSetup(const base::StateSpacePtr& space) : si_(std::make_shared<base::SpaceInformation>(space)) {}
  const base::SpaceInformationPtr& getSpaceInformation() const { return si_; }
  const base::StateSpacePtr& getStateSpace() const { return si_->getStateSpace(); }
  void setStateValidityChecker(const base::StateValidityCheckerPtr& svc) { si_->setStateValidityChecker(svc); }
  void clear() {}
  void setup() {}

 private:
  base::SpaceInformationPtr si_;
};
}  // namespace geometric
}  // namespace ompl
