// Exercises art_planner::Planner of the host mirror (reference surface: setMap / plan / getSolutionPath).
//   test_planner <fixture.bin>
// fixture: int32 rows, cols; float64 len_x len_y pos_x pos_y; float32 elevation[rows*cols] (col-major),
//          float32 traversability[rows*cols]; float64 start[7], goal[7] (x y z qx qy qz qw, both valid)
// Exit code 0 = every check below holds; 3 = no GPU (the constructor throws: no CPU fallback).
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <fstream>
#include <vector>

#include "art_planner/planner.h"

using namespace art_planner;

// OMPL states are not copyable: fill a caller-owned one
static void toState(const double* s, Planner::StateType* st) {
  st->setXYZ(s[0], s[1], s[2]);
  st->rotation().x = s[3];
  st->rotation().y = s[4];
  st->rotation().z = s[5];
  st->rotation().w = s[6];
}

static int fails = 0;
#define CHECK(cond)                                              \
  do {                                                           \
    if (!(cond)) {                                               \
      std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
      ++fails;                                                   \
    }                                                            \
  } while (0)

int main(int argc, char** argv) {
  auto params = std::make_shared<Params>();
  // shipped YAML robot (art_planner_ros/config/params.yaml:55-71)
  params->robot.torso.length = 1.31; params->robot.torso.width = 0.65; params->robot.torso.height = 0.3;
  params->robot.torso.offset.z = 0.04;
  params->robot.feet.offset.x = 0.51; params->robot.feet.offset.y = 0.2; params->robot.feet.offset.z = -0.475;
  params->robot.feet.reach.x = 0.2; params->robot.feet.reach.y = 0.2; params->robot.feet.reach.z = 0.2;
  params->planner.start_goal_search.start_radius = 0.3;
  params->planner.start_goal_search.goal_radius = 0.3;
  params->planner.start_goal_search.n_iter = 64;
  params->planner.prm_motion_cost.max_n_vertices = 4000;
  // the default planner name is LazyPRM*: it keeps growing the roadmap for plan_time (baseSolve's while (!ptc))
  params->planner.plan_time = 0.05;
  std::unique_ptr<Planner> planner;
  try {
    planner.reset(new Planner(params, 0));
  } catch (const std::exception& e) {
    std::printf("no GPU context: %s\n", e.what());
    return 3;
  }
  if (argc < 2) return 2;
  std::ifstream f(argv[1], std::ios::binary);
  int32_t rows, cols;
  double geo[4], sg[14];
  f.read(reinterpret_cast<char*>(&rows), 4);
  f.read(reinterpret_cast<char*>(&cols), 4);
  f.read(reinterpret_cast<char*>(geo), 32);
  std::vector<float> elev(static_cast<size_t>(rows) * cols), trav(elev.size());
  f.read(reinterpret_cast<char*>(elev.data()), elev.size() * 4);
  f.read(reinterpret_cast<char*>(trav.data()), trav.size() * 4);
  f.read(reinterpret_cast<char*>(sg), sizeof(sg));
  if (!f) return 2;
  Planner::StateType start, goal;
  toState(sg, &start);
  toState(sg + 7, &goal);

  // no map yet (planner.cpp:196-199); a failed plan has no path (:268-270)
  CHECK(planner->plan(start, goal) == PlannerStatus::NO_MAP);
  bool threw = false;
  try {
    planner->getSolutionPathFlat();
  } catch (const std::exception&) {
    threw = true;
  }
  CHECK(threw);

  const Map::Geometry g{rows, cols, geo[0] / rows, geo[0], geo[1], geo[2], geo[3]};
  {  // a map without the elevation layer is ignored (:137-144)
    std::unique_ptr<Map> m(new Map);
    m->setGeometry(g);
    m->addLayer("something_else", elev.data());
    planner->setMap(std::move(m));
    CHECK(!planner->hasMap());
  }
  auto make_map = [&]() {
    std::unique_ptr<Map> m(new Map);
    m->setGeometry(g);
    m->addLayer("elevation", elev.data());
    m->addLayer("traversability", trav.data());
    return m;
  };
  planner->setMap(make_map());
  CHECK(planner->hasMap());

  for (int round = 0; round < 2; ++round) {  // round 1 re-queries the kept roadmap after another setMap
    if (round == 1) planner->setMap(make_map());
    const PlannerStatus st = planner->plan(start, goal);
    CHECK(st == PlannerStatus::SOLVED);
    if (st != PlannerStatus::SOLVED) continue;
    const auto path = planner->getSolutionPathFlat(false);
    const double cost = planner->getSolutionCost();
    const auto simple = planner->getSolutionPathFlat(true);
    CHECK(path.size() >= 2 && simple.size() >= 2 && simple.size() <= path.size());
    for (const auto* p : {&path, &simple}) {
      // endpoints: the start as given (it is valid), the goal within the goal region
      CHECK(std::hypot((*p)[0][0] - sg[0], (*p)[0][1] - sg[1]) < 1e-9);
      CHECK(std::hypot(p->back()[0] - sg[7], p->back()[1] - sg[8]) <= 0.3 + 1e-9);
      // every state valid, every motion valid -- checked through the per-state C ABI
      std::vector<double> flat;
      for (const auto& s : *p) flat.insert(flat.end(), s.begin(), s.end());
      std::vector<uint8_t> v(p->size());
      CHECK(artp_validate_states(planner->gpu()->get(), flat.data(), v.size(), v.data(), nullptr) == ARTP_OK);
      for (const uint8_t x : v) CHECK(x != 0);
      std::vector<uint8_t> mv(p->size() - 1);
      CHECK(artp_check_motions(planner->gpu()->get(), flat.data(), flat.data() + 7, mv.size(), mv.data()) == ARTP_OK);
      for (const uint8_t x : mv) CHECK(x != 0);
    }
    std::printf("round %d: %zu states, cost %.4f; simplified %zu states; roadmap %zu vertices %zu edges\n", round,
                path.size(), cost, simple.size(), planner->roadmap()->numVertices(), planner->roadmap()->numEdges());
    CHECK(planner->roadmap()->numVertices() > 4002);  // it grew while planning
  }

  {  // the reference planner's own graph construction (LazyPRM* here: predecessor-only direct edges) behind the same surface
    planner->setReferenceConstruction(true);
    const PlannerStatus st = planner->plan(start, goal);
    CHECK(st == PlannerStatus::SOLVED);
    if (st == PlannerStatus::SOLVED) {
      const auto path = planner->getSolutionPathFlat(false);
      std::vector<double> flat;
      for (const auto& s : path) flat.insert(flat.end(), s.begin(), s.end());
      std::vector<uint8_t> mv(path.size() - 1);
      CHECK(artp_check_motions(planner->gpu()->get(), flat.data(), flat.data() + 7, mv.size(), mv.data()) == ARTP_OK);
      for (const uint8_t x : mv) CHECK(x != 0);
      std::printf("reference construction: %zu states, cost %.4f; roadmap %zu vertices %zu edges\n", path.size(),
                  planner->getSolutionCost(), planner->roadmap()->numVertices(), planner->roadmap()->numEdges());
    }
    planner->setReferenceConstruction(false);
  }

  // a start far off the map cannot be repaired by the region search (start.cpp:40-46 -> INVALID_START)
  Planner::StateType off;
  toState(sg, &off);
  off.setX(geo[2] + 10.0 * geo[0]);
  CHECK(planner->plan(off, goal) == PlannerStatus::INVALID_START);
  threw = false;
  try {
    planner->getSolutionPathFlat();
  } catch (const std::exception&) {
    threw = true;
  }
  CHECK(threw);
  // a goal outside the bounds is clipped to them (planner.cpp:204-221), then found invalid there
  Planner::StateType far_goal;
  toState(sg + 7, &far_goal);
  far_goal.setX(geo[2] + 10.0 * geo[0]);
  CHECK(planner->plan(start, far_goal) == PlannerStatus::INVALID_GOAL);

  std::printf("planner mirror: %d failed checks\n", fails);
  return fails == 0 ? 0 : 1;
}
