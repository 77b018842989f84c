// art_planner::BatchPRM -- the batched counterpart of the reference's PRM planners
// (PRMMotionCost, art_planner/include/art_planner/planners/prm_motion_cost.h:39-120; driven by
// Planner::plan, art_planner/src/planner.cpp:218-330): sampleGraph + addValidMilestone's connection rule
// + baseSolve / constructSolution, each as one batch on the MI355X (include/artp_c.h: artp_roadmap_*).
// Reads the same Params members the reference planners read (planner.prm_motion_cost.max_n_vertices,
// objectives.custom_path_length.*).  The caller keeps the map current through StateValidityChecker /
// SE3FromSE2Sampler / BatchMotionValidator::setZBounds exactly as for the per-state interfaces.
#pragma once

#include <array>
#include <cstdint>
#include <vector>

#include "art_planner/gpu_context.h"
#include "art_planner/ompl_min.h"
#include "art_planner/params.h"

namespace ob = ompl::base;

namespace art_planner {

class BatchPRM {
 public:
  using StateArray = std::array<double, 7>;  // x y z qx qy qz qw (OMPL SE3 state, flattened)

  BatchPRM(const ParamsConstPtr& params, const GpuContextPtr& gpu) : params_(params), gpu_(gpu) {}
  const GpuContextPtr& gpu() const { return gpu_; }
  ~BatchPRM() { clear(); }
  BatchPRM(const BatchPRM&) = delete;
  BatchPRM& operator=(const BatchPRM&) = delete;

  void setSeed(uint64_t seed) { seed_ = seed; }
  // artp_roadmap_params::construction (include/artp_c.h): 0 = the batched front end (default), 1 = the graph of
  // PRMMotionCost::addValidMilestone (chain vertices included), 2 = the graph of LazyPRMStarMinUpdate.  Takes effect at
  // the next sampleGraph.
  void setConstruction(int construction) { construction_ = construction; }
  int construction() const { return construction_; }
  void clear() {
    if (rm_) artp_roadmap_destroy(rm_);
    rm_ = nullptr;
  }

  // PRMMotionCostMaintainer::sampleGraph + the connection loop: (re)build the roadmap for start / goal.
  // Throws on an invalid start or goal, like the reference's INVALID_START / INVALID_GOAL statuses.
  void sampleGraph(const ob::SE3StateSpace::StateType& start, const ob::SE3StateSpace::StateType& goal) {
    sampleGraph(flatten(start), flatten(goal));
  }
  void sampleGraph(const StateArray& s, const StateArray& g) {
    clear();
    artp_roadmap_params p;
    artp_roadmap_params_defaults(&p);
    p.seed = seed_;
    p.construction = construction_;
    p.n_milestones = params_->planner.prm_motion_cost.max_n_vertices;
    // planner.name selects the objective like Planner::Planner (planner.cpp:108-127): the learned motion
    // cost for "prm_motion_cost", PathLengthObjective otherwise
    p.objective = params_->planner.name == "prm_motion_cost"
                      ? 2
                      : (params_->objectives.custom_path_length.use_directional_cost ? 1 : 0);
    p.w_energy = params_->planner.prm_motion_cost.cost_weights.energy;
    p.w_time = params_->planner.prm_motion_cost.cost_weights.time;
    p.w_risk = params_->planner.prm_motion_cost.cost_weights.risk;
    p.risk_threshold = params_->planner.prm_motion_cost.risk_threshold;
    p.max_query_edge_length = params_->planner.prm_motion_cost.max_query_edge_length;  // motion_cost_objective.cpp:42
    p.max_lon_vel = params_->objectives.custom_path_length.max_lon_vel;
    p.max_lat_vel = params_->objectives.custom_path_length.max_lat_vel;
    p.max_ang_vel = params_->objectives.custom_path_length.max_ang_vel;
    // sampleGraph's budgets and in-build re-weighting (prm_motion_cost.cpp:171-193); the re-weighting needs the
    // preprocessing result of the installed map (setDensityMap)
    const bool prm_motion_cost = params_->planner.name == "prm_motion_cost";  // the budgets are that planner's
    p.max_n_edges = prm_motion_cost ? params_->planner.prm_motion_cost.max_n_edges : 0;
    p.max_sample_time = prm_motion_cost ? params_->planner.prm_motion_cost.max_sample_time : 0.0;
    if (prm_motion_cost && density_map_ && params_->sampler.sample_from_distribution &&
        params_->sampler.use_inverse_vertex_density) {
      p.recompute_density_after_n_samples = params_->planner.prm_motion_cost.recompute_density_after_n_samples;
      p.density_map = density_map_;
      p.density_params = &density_params_;
    }
    throwOnError(gpu_->get(), artp_roadmap_build(gpu_->get(), &p, s.data(), g.data(), &rm_), "artp_roadmap_build");
  }

  // baseSolve / constructSolution: false = start and goal are not connected (PlannerStatus::TIMEOUT).
  bool solve(std::vector<StateArray>* path, double* cost = nullptr) {
    if (!rm_) throw std::runtime_error("BatchPRM::solve before sampleGraph");
    size_t n = 0;
    double c = 0.0;
    int replans = 0;
    int rc = artp_roadmap_solve(rm_, nullptr, 0, &n, &c, &replans);  // length first
    if (rc != ARTP_OK) throwOnError(gpu_->get(), rc, "artp_roadmap_solve");
    if (n == 0) return false;
    path->resize(n);
    rc = artp_roadmap_solve(rm_, (*path)[0].data(), n, &n, &c, &replans);
    throwOnError(gpu_->get(), rc, "artp_roadmap_solve");
    if (cost) *cost = c;
    return true;
  }

  // LazyPRMStarMinUpdate::baseSolve (lazy_prm_star_min_update.cpp:552-615): grow while planning until plan_time is
  // over, return the best solution found.  false = never connected (PlannerStatus::TIMEOUT).
  bool solveUntil(double plan_time, unsigned grow_step, std::vector<StateArray>* path, double* cost = nullptr) {
    if (!rm_) throw std::runtime_error("BatchPRM::solveUntil before sampleGraph");
    std::vector<StateArray> buf(4096);
    size_t n = 0;
    double c = 0.0;
    uint64_t stats[3] = {0, 0, 0};
    int rc = artp_roadmap_solve_until(rm_, plan_time, grow_step, buf[0].data(), buf.size(), &n, &c, stats);
    if (rc == ARTP_ERR_CAPACITY && n > buf.size()) {  // a longer path than the buffer: once more with room
      buf.resize(n);
      rc = artp_roadmap_solve_until(rm_, 0.0, 0, buf[0].data(), buf.size(), &n, &c, stats);
    }
    throwOnError(gpu_->get(), rc, "artp_roadmap_solve_until");
    if (n == 0) return false;
    buf.resize(n);
    path->swap(buf);
    if (cost) *cost = c;
    return true;
  }

  // The preprocessing result of the map the context has installed (Planner::setMap): what the in-build and
  // in-growth re-weighting of the sampling distribution is computed on.  Call it again after every map change,
  // BEFORE the old result is destroyed (nullptr = no re-weighting).
  void setDensityMap(artp_preprocessed* pp, const artp_preprocess_params& prm) {
    density_map_ = pp;
    density_params_ = prm;
    if (rm_)
      throwOnError(gpu_->get(), artp_roadmap_set_density_map(rm_, pp, pp ? &density_params_ : nullptr),
                   "artp_roadmap_set_density_map");
  }

  // LazyPRMStarMinUpdate's roadmap maintenance (lazy_prm_star_min_update.cpp:18-217), batched: after the map
  // changed, re-check every vertex and edge of the kept roadmap.  Returns false when start or goal became
  // invalid (the caller then re-queries with new ones).
  bool revalidate() {
    if (!rm_) throw std::runtime_error("BatchPRM::revalidate before sampleGraph");
    uint64_t out[4] = {};
    throwOnError(gpu_->get(), artp_roadmap_revalidate(rm_, out), "artp_roadmap_revalidate");
    return (out[3] & 3u) == 3u;
  }

  // PRMMotionCostMaintainer::sampleGraph between queries (prm_motion_cost.cpp:145-219): n_more milestones on top
  // of the ones the current map still accepts; returns how many of the old ones the map invalidated
  size_t grow(size_t n_more) {
    if (!rm_) throw std::runtime_error("BatchPRM::grow before sampleGraph");
    uint64_t out[2] = {};
    throwOnError(gpu_->get(), artp_roadmap_grow(rm_, n_more, out), "artp_roadmap_grow");
    return static_cast<size_t>(out[1]);
  }

  // new start / goal on the kept roadmap (every OMPL query adds them as milestones)
  void setQuery(const ob::SE3StateSpace::StateType& start, const ob::SE3StateSpace::StateType& goal) {
    setQuery(flatten(start), flatten(goal));
  }
  void setQuery(const StateArray& s, const StateArray& g) {
    if (!rm_) throw std::runtime_error("BatchPRM::setQuery before sampleGraph");
    throwOnError(gpu_->get(), artp_roadmap_set_query(rm_, s.data(), g.data()), "artp_roadmap_set_query");
  }

  // Planner::getSolutionPath(simplify = true) (planner.cpp:266-280): deterministic batched shortcutting
  void simplify(std::vector<StateArray>* path, double* cost = nullptr) {
    if (!rm_ || path->empty()) return;
    std::vector<StateArray> out(path->size());
    size_t n = 0;
    double c = 0.0;
    throwOnError(gpu_->get(),
                 artp_roadmap_simplify_path(rm_, (*path)[0].data(), path->size(), out[0].data(), &n, &c),
                 "artp_roadmap_simplify_path");
    out.resize(n);
    path->swap(out);
    if (cost) *cost = c;
  }

  // the roadmap's vertices (start, goal, milestones), n x 7 -- what computeInverseSampleDensity counts per cell
  std::vector<double> vertices() const {
    std::vector<double> v(numVertices() * 7);
    if (rm_ && !v.empty())
      throwOnError(gpu_->get(),
                   artp_roadmap_export(rm_, v.data(), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr),
                   "artp_roadmap_export");
    return v;
  }

  // the roadmap's candidate edges: uv (n x 2, u < v), the interpolation-rule verdict, "removed by the lazy path check",
  // the objective's cost (any pointer may be null)
  void edges(std::vector<uint32_t>* uv, std::vector<uint8_t>* valid, std::vector<uint8_t>* removed,
             std::vector<double>* cost) const {
    const size_t n = rm_ ? stat(1) : 0;
    if (uv) uv->assign(2 * n, 0u);
    if (valid) valid->assign(n, 0);
    if (removed) removed->assign(n, 0);
    if (cost) cost->assign(n, 0.0);
    if (n)
      throwOnError(gpu_->get(),
                   artp_roadmap_export(rm_, nullptr, nullptr, nullptr, uv ? uv->data() : nullptr, valid ? valid->data() : nullptr,
                                       nullptr, cost ? cost->data() : nullptr, removed ? removed->data() : nullptr),
                   "artp_roadmap_export");
  }

  size_t numVertices() const { return stat(0); }
  size_t numEdges() const { return stat(2); }
  size_t numCandidateEdges() const { return stat(1); }

  static StateArray flatten(const ob::SE3StateSpace::StateType& s) {
    return {s.getX(), s.getY(), s.getZ(), s.rotation().x, s.rotation().y, s.rotation().z, s.rotation().w};
  }

 private:
  size_t stat(int i) const {
    uint64_t out[8] = {};
    if (rm_) artp_roadmap_stats(rm_, out);
    return static_cast<size_t>(out[i]);
  }
  ParamsConstPtr params_;
  GpuContextPtr gpu_;
  artp_roadmap* rm_{nullptr};
  uint64_t seed_{42};
  int construction_{0};
  artp_preprocessed* density_map_{nullptr};
  artp_preprocess_params density_params_{};
};

// The roadmap as art_planner::Planner and the ob::Planner shell set as ss_'s planner (planners/prm_motion_cost.h,
// planners/lazy_prm_star_min_update.h) share it: one graph, whoever drives it.
struct RoadmapHandle {
  std::shared_ptr<BatchPRM> prm;
  ParamsConstPtr params;
  bool built{false};   // sampleGraph has run for the current graph (Planner's have_roadmap_)
};

}  // namespace art_planner
