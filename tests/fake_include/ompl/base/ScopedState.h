// TEST SCAFFOLD, not OMPL: ompl::base::ScopedState<> as far as art_planner::Planner's reference signatures use it
// (planner.h:67-68: plan(const ob::ScopedState<>&, ...); callers fill it through get() / operator->), so that the
// -DARTP_HAVE_OMPL -DARTP_HAVE_GRID_MAP branch of the host mirror goes through a compiler in this image.
#pragma once
#include "art_planner/ompl_standins.h"
namespace ompl {
namespace base {
template <class T = StateSpace>
class ScopedState {
 public:
  explicit ScopedState(const StateSpacePtr& space) : space_(space), state_(space->allocState()) {}
  explicit ScopedState(const SpaceInformationPtr& si) : space_(si->getStateSpace()), state_(space_->allocState()) {}
  ScopedState(const ScopedState& other) : space_(other.space_), state_(space_->allocState()) {
    space_->copyState(state_, other.state_);
  }
  ScopedState& operator=(const ScopedState&) = delete;
  ~ScopedState() { space_->freeState(state_); }
  State* get() { return state_; }
  const State* get() const { return state_; }
  State* operator->() { return state_; }
  const State* operator->() const { return state_; }
  const StateSpacePtr& getSpace() const { return space_; }

 private:
  StateSpacePtr space_;
  State* state_;
};
}  // namespace base
}  // namespace ompl
