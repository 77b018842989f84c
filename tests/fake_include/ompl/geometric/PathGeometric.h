// TEST SCAFFOLD, not OMPL: ompl::geometric::PathGeometric as far as Planner::getSolutionPath's reference signature
// (planner.h:70) and PlannerRos's converter (pathOmplToRos walks getStates()) use it.
#pragma once
#include <vector>
#include "art_planner/ompl_standins_planning.h"
namespace ompl {
namespace geometric {
class PathGeometric : public base::Path {
 public:
  explicit PathGeometric(const base::SpaceInformationPtr& si) : base::Path(si) {}
  PathGeometric(const PathGeometric& other) : base::Path(other.si_) {
    for (const base::State* s : other.states_) append(s);
  }
  PathGeometric& operator=(const PathGeometric&) = delete;
  ~PathGeometric() override {
    for (base::State* s : states_) si_->freeState(s);
  }
  void append(const base::State* state) {  // copies, like OMPL
    base::State* s = si_->allocState();
    si_->copyState(s, state);
    states_.push_back(s);
  }
  std::size_t getStateCount() const { return states_.size(); }
  base::State* getState(unsigned int index) { return states_[index]; }
  const base::State* getState(unsigned int index) const { return states_[index]; }
  std::vector<base::State*>& getStates() { return states_; }

 private:
  std::vector<base::State*> states_;
};
}  // namespace geometric
}  // namespace ompl
