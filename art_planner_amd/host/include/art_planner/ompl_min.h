// Minimal OMPL-shaped interfaces (just the virtuals the hot path overrides) for builds where OMPL
// 1.4.2 is not installed (it is not in this image; art_planner/README.md:53 pins it).  With
// -DARTP_HAVE_OMPL the real headers are used instead and this file is empty.
#pragma once

#ifdef ARTP_HAVE_OMPL
#include <ompl/base/MotionValidator.h>
#include <ompl/base/SpaceInformation.h>
#include <ompl/base/StateSampler.h>
#include <ompl/base/StateValidityChecker.h>
#include <ompl/base/spaces/SE3StateSpace.h>
#else

#include <memory>

namespace ompl {
namespace base {

class State {
 public:
  virtual ~State() = default;
  template <class T>
  const T* as() const { return static_cast<const T*>(this); }
  template <class T>
  T* as() { return static_cast<T*>(this); }
};

class SO3StateSpace {
 public:
  struct StateType : public State {
    double x{0}, y{0}, z{0}, w{1};
    void setIdentity() { x = y = z = 0; w = 1; }
  };
};

class StateSpace {
 public:
  virtual ~StateSpace() = default;
};

class SE3StateSpace : public StateSpace {
 public:
  class StateType : public State {
    double xyz_[3]{0, 0, 0};
    SO3StateSpace::StateType rot_;
   public:
    double getX() const { return xyz_[0]; }
    double getY() const { return xyz_[1]; }
    double getZ() const { return xyz_[2]; }
    void setX(double v) { xyz_[0] = v; }
    void setY(double v) { xyz_[1] = v; }
    void setZ(double v) { xyz_[2] = v; }
    void setXYZ(double x, double y, double z) { xyz_[0] = x; xyz_[1] = y; xyz_[2] = z; }
    const SO3StateSpace::StateType& rotation() const { return rot_; }
    SO3StateSpace::StateType& rotation() { return rot_; }
  };
};

class SpaceInformation {};
using SpaceInformationPtr = std::shared_ptr<SpaceInformation>;

class StateValidityChecker {
 public:
  explicit StateValidityChecker(const SpaceInformationPtr& si) : si_(si) {}
  virtual ~StateValidityChecker() = default;
  virtual bool isValid(const State* state) const = 0;
 protected:
  SpaceInformationPtr si_;
};

class MotionValidator {
 public:
  explicit MotionValidator(const SpaceInformationPtr& si) : si_(si) {}
  virtual ~MotionValidator() = default;
  virtual bool checkMotion(const State* s1, const State* s2) const = 0;
 protected:
  SpaceInformationPtr si_;
};

class StateSampler {
 public:
  explicit StateSampler(const StateSpace* space) : space_(space) {}
  virtual ~StateSampler() = default;
  virtual void sampleUniform(State* state) = 0;
 protected:
  const StateSpace* space_;
};

}  // namespace base
}  // namespace ompl
#endif
