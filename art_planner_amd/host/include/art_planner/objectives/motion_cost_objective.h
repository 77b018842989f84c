// The batch cost functor of the reference's MotionCostObjective
// (art_planner/include/art_planner/objectives/motion_cost_objective.h:22-23,54-66), implemented on the
// GPU instead of a ROS service call to the Python node (art_planner_ros/src/planner_ros.cpp:283-318).
#pragma once

#include <functional>
#include <memory>
#include <vector>

#ifdef ARTP_HAVE_EIGEN
#include <Eigen/Dense>
#endif

#include "art_planner/gpu_context.h"

namespace art_planner {

#ifdef ARTP_HAVE_EIGEN
using EdgeMatrix = Eigen::Matrix<float, Eigen::Dynamic, Eigen::Dynamic, Eigen::RowMajor>;
#else
// Row-major float matrix with the three Eigen members the reference's call sites use.
class EdgeMatrix {
 public:
  EdgeMatrix() = default;
  EdgeMatrix(long r, long c) { resize(r, c); }
  void resize(long r, long c) {
    rows_ = r;
    cols_ = c;
    data_.assign(static_cast<size_t>(r * c), 0.0f);
  }
  long rows() const { return rows_; }
  long cols() const { return cols_; }
  float& operator()(long r, long c) { return data_[static_cast<size_t>(r * cols_ + c)]; }
  float operator()(long r, long c) const { return data_[static_cast<size_t>(r * cols_ + c)]; }
  float* data() { return data_.data(); }
  const float* data() const { return data_.data(); }

 private:
  long rows_{0}, cols_{0};
  std::vector<float> data_;
};
#endif

using MotionCostFunc = std::function<bool(const EdgeMatrix&, EdgeMatrix*)>;

// "query only" functor (planner_ros.cpp:309-318): edge_matrix is B x 6 (target x y yaw, start x y yaw),
// edge_cost must already be sized B x 3 by the caller (motion_cost_objective.cpp:30,
// prm_motion_cost.cpp:29); returns false on failure like the service client does.
inline std::unique_ptr<MotionCostFunc> makeGpuMotionCostFunc(const GpuContextPtr& gpu) {
  return std::unique_ptr<MotionCostFunc>(new MotionCostFunc([gpu](const EdgeMatrix& edges, EdgeMatrix* cost) {
    if (!cost || edges.cols() != 6 || cost->rows() != edges.rows() || cost->cols() != 3) return false;
    return artp_cost_query(gpu->get(), edges.data(), static_cast<size_t>(edges.rows()), cost->data()) == ARTP_OK;
  }));
}

// MotionCostObjective::getCost / isFeasible (motion_cost_objective.h:54-66) on one cost row
inline double motionCostWeighted(const Params& p, const float* c) {
  const auto& w = p.planner.prm_motion_cost.cost_weights;
  return c[0] * w.energy + c[1] * w.time + c[2] * w.risk;
}
inline bool motionCostFeasible(const Params& p, const float* c) { return c[2] <= p.planner.prm_motion_cost.risk_threshold; }

}  // namespace art_planner

#ifdef ARTP_HAVE_OMPL
#include <cmath>
#include <limits>
#include <stdexcept>
#include <ompl/base/OptimizationObjective.h>
#include "art_planner/params.h"
#include "art_planner/validity_checker/validity_checker.h"

namespace art_planner {

// MotionCostObjective with the reference's constructor and members (motion_cost_objective.h:28-75,
// motion_cost_objective.cpp:14-95): what PlannerRos hands to ss_->setOptimizationObjective (planner_ros.cpp:313-318).
// motionCost splits a motion into n_interp + 1 = (unsigned)(lateral distance / max_query_edge_length) + 1 cost queries
// through the functor; any infeasible part makes it infinite.  (The roadmap itself prices its edges on the device; this
// class serves callers that ask the objective directly -- path simplification in OMPL, tests.)
class MotionCostObjective : public ob::OptimizationObjective {
 public:
  using EdgeMatrix = art_planner::EdgeMatrix;
  using MotionCostFunc = art_planner::MotionCostFunc;

  MotionCostObjective(const ob::SpaceInformationPtr& si, const ParamsConstPtr& params,
                      std::unique_ptr<MotionCostFunc>&& motion_cost_func)
      : ob::OptimizationObjective(si), params_(params), motion_cost_func_(std::move(motion_cost_func)) {}

  double getCost(const float* c) const { return motionCostWeighted(*params_, c); }
  bool isFeasible(const float* c) const { return motionCostFeasible(*params_, c); }

  bool costQuery(const EdgeMatrix& edge_matrix, EdgeMatrix* edge_cost) const {
    edge_cost->resize(edge_matrix.rows(), 3);
    return (*motion_cost_func_)(edge_matrix, edge_cost);
  }

  ob::Cost stateCost(const ob::State*) const override { return ob::Cost(0.0); }

  ob::Cost motionCost(const ob::State* s1, const ob::State* s2) const override {
    double a[7], b[7];
    flattenSE3(s1, a);
    flattenSE3(s2, b);
    const double dist = std::hypot(b[0] - a[0], b[1] - a[1]);   // lateralDistance (utils.h:52-61)
    const unsigned n_interp = static_cast<unsigned>(dist / params_->planner.prm_motion_cost.max_query_edge_length);
    EdgeMatrix em(n_interp + 1, 6), ec(n_interp + 1, 3);
    const double div = 1.0 / (n_interp + 1);
    em(0, 3) = static_cast<float>(a[0]);
    em(0, 4) = static_cast<float>(a[1]);
    em(0, 5) = static_cast<float>(yawOf(a));
    em(n_interp, 0) = static_cast<float>(b[0]);
    em(n_interp, 1) = static_cast<float>(b[1]);
    em(n_interp, 2) = static_cast<float>(yawOf(b));
    for (unsigned step = 1; step < n_interp + 1; ++step) {
      double c[7];
      interpolateSE3(a, b, step * div, c);
      const float x = static_cast<float>(c[0]), y = static_cast<float>(c[1]), yaw = static_cast<float>(yawOf(c));
      em(step - 1, 0) = x;
      em(step - 1, 1) = y;
      em(step - 1, 2) = yaw;
      em(step, 3) = x;
      em(step, 4) = y;
      em(step, 5) = yaw;
    }
    if (!costQuery(em, &ec)) throw std::runtime_error("Motion cost call failed");
    double cost = 0.0;
    for (unsigned i = 0; i < n_interp + 1; ++i) {
      const float row[3] = {ec(i, 0), ec(i, 1), ec(i, 2)};
      if (row[2] > params_->planner.prm_motion_cost.risk_threshold) return ob::Cost(std::numeric_limits<double>::infinity());
      cost += getCost(row);
    }
    return ob::Cost(cost);
  }
  ob::Cost motionCostHeuristic(const ob::State*, const ob::State*) const override { return ob::Cost(0.0); }

 private:
  static double yawOf(const double* s) {   // getYawFromSO3 (utils.h:80-88)
    return std::atan2(2 * (s[6] * s[5] + s[3] * s[4]), 1 - 2 * (s[4] * s[4] + s[5] * s[5]));
  }
  // SE3StateSpace::interpolate of OMPL 1.4.2 on flattened states: xyz lerp, SO3 slerp (theta = acos(|q1.q2|), sign flip,
  // copy of q1 below the numerical threshold) -- the arithmetic oracle/artp_oracle.c:1113-1165 restates
  static void interpolateSE3(const double* a, const double* b, double t, double* out) {
    for (int i = 0; i < 3; ++i) out[i] = a[i] + (b[i] - a[i]) * t;
    double dq = a[3] * b[3] + a[4] * b[4] + a[5] * b[5] + a[6] * b[6];
    double adq = std::fabs(dq);
    const double theta = adq > 1.0 - 1e-9 ? 0.0 : std::acos(adq);
    if (theta > std::numeric_limits<double>::epsilon()) {
      const double d = 1.0 / std::sin(theta), s0 = std::sin((1.0 - t) * theta);
      double s1 = std::sin(t * theta);
      if (dq < 0) s1 = -s1;
      for (int i = 3; i < 7; ++i) out[i] = (a[i] * s0 + b[i] * s1) * d;
    } else {
      for (int i = 3; i < 7; ++i) out[i] = a[i];
    }
  }
  ParamsConstPtr params_;
  std::unique_ptr<MotionCostFunc> motion_cost_func_;
};

}  // namespace art_planner
#endif  // ARTP_HAVE_OMPL
