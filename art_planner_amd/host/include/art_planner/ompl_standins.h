// The OMPL-1.4.2-shaped stand-in definitions of ompl_min.h (see there).  Kept in a header of their own so that
// tests/fake_include/ompl/... can present them under the REAL include names: tests/test_host_mirror.py compiles the
// host mirror with -DARTP_HAVE_OMPL -DARTP_HAVE_EIGEN against those, which is the preprocessor branch INTEGRATION.md
// tells a maintainer to build (the branch that includes <ompl/base/...> and <Eigen/Dense>).
#pragma once

#include <cmath>
#include <functional>
#include <memory>
#include <random>
#include <utility>
#include <vector>

namespace ompl {

// ompl/util/RandomNumbers.h (the members the path's call sites use: sampler.cpp:58-59,105,116)
class RNG {
 public:
  RNG() : generator_(std::random_device{}()) {}
  explicit RNG(std::uint_fast32_t seed) : generator_(seed) {}
  double uniform01() { return uniDist_(generator_); }
  double uniformReal(double lower_bound, double upper_bound) { return (upper_bound - lower_bound) * uniDist_(generator_) + lower_bound; }
  double gaussian01() { return normalDist_(generator_); }
  double gaussian(double mean, double stddev) { return normalDist_(generator_) * stddev + mean; }
  void eulerRPY(double value[3]) {
    value[0] = M_PI * (-2.0 * uniDist_(generator_) + 1.0);
    value[1] = std::acos(1.0 - 2.0 * uniDist_(generator_)) - M_PI / 2.0;
    value[2] = M_PI * (-2.0 * uniDist_(generator_) + 1.0);
  }

 private:
  std::mt19937 generator_;
  std::uniform_real_distribution<> uniDist_{0.0, 1.0};
  std::normal_distribution<> normalDist_{0.0, 1.0};
};

namespace base {

// ompl/base/State.h: not copyable, only constructible through derived state types (as in OMPL, code that wants a
// second state allocates one and copies the VALUES over)
class State {
 private:
  State(const State&) = delete;
  State& operator=(const State&) = delete;

 protected:
  State() = default;
  virtual ~State() = default;

 public:
  template <class T>
  const T* as() const { return static_cast<const T*>(this); }
  template <class T>
  T* as() { return static_cast<T*>(this); }
};

// ompl/base/spaces/RealVectorBounds.h
class RealVectorBounds {
 public:
  explicit RealVectorBounds(unsigned int dim) { resize(dim); }
  void setLow(double value) { low.assign(low.size(), value); }
  void setHigh(double value) { high.assign(high.size(), value); }
  void setLow(unsigned int index, double value) { low[index] = value; }
  void setHigh(unsigned int index, double value) { high[index] = value; }
  void resize(std::size_t size) {
    low.resize(size, 0.0);
    high.resize(size, 0.0);
  }
  std::vector<double> low;
  std::vector<double> high;
};

class StateSampler;
class StateSpace;
using StateSamplerPtr = std::shared_ptr<StateSampler>;
using StateSamplerAllocator = std::function<StateSamplerPtr(const StateSpace*)>;

class StateSpace {
 public:
  virtual ~StateSpace() = default;
  template <class T>
  const T* as() const { return static_cast<const T*>(this); }
  template <class T>
  T* as() { return static_cast<T*>(this); }
  // ompl/base/StateSpace.h: pure virtual there too
  virtual State* allocState() const = 0;
  virtual void freeState(State* state) const = 0;
  virtual void copyState(State* destination, const State* source) const = 0;
  void setStateSamplerAllocator(const StateSamplerAllocator& ssa) { ssa_ = ssa; }
  StateSamplerPtr allocStateSampler() const { return ssa_ ? ssa_(this) : StateSamplerPtr(); }

 private:
  StateSamplerAllocator ssa_;
};
using StateSpacePtr = std::shared_ptr<StateSpace>;

class SO3StateSpace : public StateSpace {
 public:
  class StateType : public State {
   public:
    StateType() = default;
    ~StateType() override = default;
    void setIdentity() { x = y = z = 0; w = 1; }
    double x{0}, y{0}, z{0}, w{1};
  };
  State* allocState() const override { return new StateType(); }
  void freeState(State* state) const override { delete state->as<StateType>(); }
  void copyState(State* destination, const State* source) const override {
    auto* d = destination->as<StateType>();
    const auto* q = source->as<StateType>();
    d->x = q->x; d->y = q->y; d->z = q->z; d->w = q->w;
  }
};

class SE3StateSpace : public StateSpace {
 public:
  class StateType : public State {
   public:
    StateType() = default;
    ~StateType() override = default;
    double getX() const { return xyz_[0]; }
    double getY() const { return xyz_[1]; }
    double getZ() const { return xyz_[2]; }
    void setX(double v) { xyz_[0] = v; }
    void setY(double v) { xyz_[1] = v; }
    void setZ(double v) { xyz_[2] = v; }
    void setXYZ(double x, double y, double z) { xyz_[0] = x; xyz_[1] = y; xyz_[2] = z; }
    const SO3StateSpace::StateType& rotation() const { return rot_; }
    SO3StateSpace::StateType& rotation() { return rot_; }

   private:
    double xyz_[3]{0, 0, 0};
    SO3StateSpace::StateType rot_;
  };
  void setBounds(const RealVectorBounds& bounds) { bounds_ = bounds; }
  const RealVectorBounds& getBounds() const { return bounds_; }
  State* allocState() const override { return new StateType(); }
  void freeState(State* state) const override { delete state->as<StateType>(); }
  void copyState(State* destination, const State* source) const override {
    auto* d = destination->as<StateType>();
    const auto* q = source->as<StateType>();
    d->setXYZ(q->getX(), q->getY(), q->getZ());
    d->rotation().x = q->rotation().x; d->rotation().y = q->rotation().y;
    d->rotation().z = q->rotation().z; d->rotation().w = q->rotation().w;
  }

 private:
  RealVectorBounds bounds_{3};
};

class StateValidityChecker;
class MotionValidator;
using StateValidityCheckerPtr = std::shared_ptr<StateValidityChecker>;
using MotionValidatorPtr = std::shared_ptr<MotionValidator>;

// ompl/base/SpaceInformation.h: the members the mirror's real-library branch touches (the real class has no default
// constructor; the stand-in keeps one for tests that only need a handle to pass around)
class SpaceInformation {
 public:
  SpaceInformation() = default;
  explicit SpaceInformation(StateSpacePtr space) : space_(std::move(space)) {}
  virtual ~SpaceInformation() = default;
  const StateSpacePtr& getStateSpace() const { return space_; }
  State* allocState() const { return space_->allocState(); }
  void freeState(State* state) const { space_->freeState(state); }
  void copyState(State* destination, const State* source) const { space_->copyState(destination, source); }
  void setStateValidityChecker(const StateValidityCheckerPtr& svc) { svc_ = svc; }
  const StateValidityCheckerPtr& getStateValidityChecker() const { return svc_; }
  void setMotionValidator(const MotionValidatorPtr& mv) { mv_ = mv; }
  const MotionValidatorPtr& getMotionValidator() const { return mv_; }

 private:
  StateSpacePtr space_;
  StateValidityCheckerPtr svc_;
  MotionValidatorPtr mv_;
};
using SpaceInformationPtr = std::shared_ptr<SpaceInformation>;

// ompl/base/StateValidityChecker.h
class StateValidityChecker {
 public:
  explicit StateValidityChecker(SpaceInformation* si) : si_(si) {}
  explicit StateValidityChecker(const SpaceInformationPtr& si) : si_(si.get()) {}
  virtual ~StateValidityChecker() = default;
  virtual bool isValid(const State* state) const = 0;
  virtual bool isValid(const State* state, double& dist) const {
    dist = clearance(state);
    return isValid(state);
  }
  virtual bool isValid(const State* state, double& dist, State* /*validState*/, bool& validStateAvailable) const {
    dist = clearance(state);
    validStateAvailable = false;
    return isValid(state);
  }
  virtual double clearance(const State* /*state*/) const { return 0.0; }

 protected:
  SpaceInformation* si_;
};

// ompl/base/MotionValidator.h
class MotionValidator {
 public:
  explicit MotionValidator(SpaceInformation* si) : si_(si) {}
  explicit MotionValidator(const SpaceInformationPtr& si) : si_(si.get()) {}
  virtual ~MotionValidator() = default;
  virtual bool checkMotion(const State* s1, const State* s2) const = 0;
  virtual bool checkMotion(const State* s1, const State* s2, std::pair<State*, double>& lastValid) const = 0;
  unsigned int getValidMotionCount() const { return valid_; }
  unsigned int getInvalidMotionCount() const { return invalid_; }
  unsigned int getCheckedMotionCount() const { return valid_ + invalid_; }
  double getValidMotionFraction() const { return valid_ == 0 ? 0.0 : (double)valid_ / (double)(invalid_ + valid_); }
  void resetMotionCounter() { valid_ = invalid_ = 0; }

 protected:
  SpaceInformation* si_;
  mutable unsigned int valid_{0};
  mutable unsigned int invalid_{0};
};

// ompl/base/StateSampler.h
class StateSampler {
 public:
  explicit StateSampler(const StateSpace* space) : space_(space) {}
  virtual ~StateSampler() = default;
  virtual void sampleUniform(State* state) = 0;
  virtual void sampleUniformNear(State* state, const State* near, double distance) = 0;
  virtual void sampleGaussian(State* state, const State* mean, double stdDev) = 0;

 protected:
  const StateSpace* space_;
  RNG rng_;
};

}  // namespace base
}  // namespace ompl
