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
