// art_planner::Params for callers of the MI355X hot path.
//
// Member names, nesting and default values follow the reference's parameter struct
// (art_planner/include/art_planner/params.h:14-123) so that code written against it -- e.g.
// PlannerRos::loadRosParameters filling `params->robot.torso.length` -- compiles unchanged.  The hot path
// itself reads: robot.*, sampler.{max_pitch_pert, max_roll_pert, sample_from_distribution},
// planner.{name, unknown_space_untraversable, elevation_layer, prm_motion_cost.*} and
// objectives.custom_path_length.* ; the remaining members are carried for source compatibility only.
#pragma once

#include <cmath>
#include <memory>
#include <string>

namespace art_planner {

namespace params_detail {

struct Vec3 {
  double x, y, z;
};

struct Safety {  // planner.safety
  double foothold_margin = 0.0, foothold_margin_max_hole_size = 0.0, foothold_margin_max_drop = 0.0;
  double foothold_margin_max_drop_search_radius = 0.0, foothold_margin_min_step = 0.0, foothold_size = 0.0;
};

struct StartGoalSearch {  // planner.start_goal_search
  double start_radius = 0.0, goal_radius = 0.0;
  unsigned int n_iter = 0;
};

struct LazyPrmStarMinUpdate {  // planner.lazy_prm_star_min_update
  bool invalidate_updated_graph_components = false;
  float height_change_for_update = 0.05f;
  bool cleanup_when_not_planning = false;
};

struct CostWeights {  // planner.prm_motion_cost.cost_weights
  float energy = 0.0f, time = 1.0f, risk = 5.0f;
};

struct PrmMotionCost {  // planner.prm_motion_cost
  double max_sample_time = 2.0;
  unsigned int max_n_vertices = 10000u, max_n_edges = 50000u, recompute_density_after_n_samples = 1000u;
  float max_query_edge_length = 0.5f, risk_threshold = 0.1f;
  CostWeights cost_weights;
};

struct PlannerSection {
  std::string name = "lazy_prm_star";
  std::string elevation_layer = "elevation", traversability_layer = "traversability";
  double plan_time = 1.0;
  unsigned int n_threads = 1;
  double replan_freq = 1.0;
  float traversability_thres = 0.5f;
  bool simplify_solution = true, snap_goal_to_map = true, unknown_space_untraversable = true;
  Safety safety;
  StartGoalSearch start_goal_search;
  LazyPrmStarMinUpdate lazy_prm_star_min_update;
  PrmMotionCost prm_motion_cost;
};

struct CustomPathLength {  // objectives.custom_path_length
  bool use_directional_cost = false;
  double max_lon_vel = 0.5, max_lat_vel = 0.1, max_ang_vel = 0.5;
};

struct ObjectivesSection {
  CustomPathLength custom_path_length;
};

struct SamplerSection {
  double max_pitch_pert = 10.0 / 180 * M_PI, max_roll_pert = 3.33 / 180 * M_PI;
  bool sample_from_distribution = true, use_inverse_vertex_density = false, use_max_prob_unknown_samples = false;
  double max_prob_unknown_samples = 0.1;
};

struct Torso {
  double length = 1.05, width = 0.55, height = 0.2;
  Vec3 offset{0.0, 0.0, 0.0};
};

struct Feet {
  Vec3 offset{0.362, 0.225, -0.525};
  Vec3 reach{0.25, 0.1, 0.15};
};

struct RobotSection {
  std::string base_frame = "base";
  Torso torso;
  Feet feet;
};

}  // namespace params_detail

struct Params {
  params_detail::PlannerSection planner;
  params_detail::ObjectivesSection objectives;
  params_detail::SamplerSection sampler;
  params_detail::RobotSection robot;
  bool verbose = false;
};

using ParamsPtr = std::shared_ptr<Params>;
using ParamsConstPtr = std::shared_ptr<const Params>;

}  // namespace art_planner
