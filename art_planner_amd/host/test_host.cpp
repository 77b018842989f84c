// Exercises the C++ host mirror (reference class names / signatures) over libartp.so.
//   test_host <fixture.bin>
// fixture: int32 rows, cols; float64 len_x len_y pos_x pos_y; float32 elevation[rows*cols] (col-major),
//          float32 elevation_masked[rows*cols]; int32 n; float64 se3[n*7]; uint8 expected[n]
// Exit code 0 = every label (single-state isValid AND batch) equals the expected (oracle) label.
// Without a GPU the context constructor must throw (no CPU fallback): exit code 3.
#include <cstdint>
#include <cstdio>
#include <fstream>
#include <vector>

#include "art_planner/objectives/motion_cost_objective.h"
#include "art_planner/sampler.h"
#include "art_planner/planners/batch_prm.h"
#include "art_planner/validity_checker/height_map_box_checker.h"
#include "art_planner/validity_checker/validity_checker.h"

using namespace art_planner;

int main(int argc, char** argv) {
  auto params = std::make_shared<Params>();
  // shipped YAML robot (art_planner_ros/config/params.yaml:55-71)
  params->robot.torso.length = 1.31; params->robot.torso.width = 0.65; params->robot.torso.height = 0.3;
  params->robot.torso.offset.z = 0.04;
  params->robot.feet.offset.x = 0.51; params->robot.feet.offset.y = 0.2; params->robot.feet.offset.z = -0.475;
  params->robot.feet.reach.x = 0.2; params->robot.feet.reach.y = 0.2; params->robot.feet.reach.z = 0.2;
  GpuContextPtr gpu;
  try {
    gpu = std::make_shared<GpuContext>(params, 0);
  } catch (const std::exception& e) {
    std::printf("no GPU context: %s\n", e.what());
    return 3;
  }
  if (argc < 2) return 2;
  std::ifstream f(argv[1], std::ios::binary);
  int32_t rows, cols, n;
  double geo[4];
  f.read(reinterpret_cast<char*>(&rows), 4);
  f.read(reinterpret_cast<char*>(&cols), 4);
  f.read(reinterpret_cast<char*>(geo), 32);
  std::vector<float> elev(static_cast<size_t>(rows) * cols), masked(elev.size());
  f.read(reinterpret_cast<char*>(elev.data()), elev.size() * 4);
  f.read(reinterpret_cast<char*>(masked.data()), masked.size() * 4);
  f.read(reinterpret_cast<char*>(&n), 4);
  std::vector<double> se3(static_cast<size_t>(n) * 7);
  std::vector<uint8_t> expected(n);
  f.read(reinterpret_cast<char*>(se3.data()), se3.size() * 8);
  f.read(reinterpret_cast<char*>(expected.data()), n);
  if (!f) return 2;

  auto map = std::make_shared<Map>();
  map->setGeometry({rows, cols, geo[0] / rows, geo[0], geo[1], geo[2], geo[3]});
  map->addLayer("elevation", elev.data());
  map->addLayer("elevation_masked", masked.data());

  StateValidityChecker checker(std::make_shared<ob::SpaceInformation>(), params, gpu);
  checker.setMap(map);
  checker.updateHeightField();
  if (!checker.hasMap()) return 4;

  int bad = 0;
  const auto batch = checker.isValidBatch(se3);
  for (int i = 0; i < n; ++i) bad += batch[i] != expected[i];
  const int n_single = n < 200 ? n : 200;
  for (int i = 0; i < n_single; ++i) {
    ob::SE3StateSpace::StateType s;
    s.setXYZ(se3[7 * i], se3[7 * i + 1], se3[7 * i + 2]);
    s.rotation().x = se3[7 * i + 3]; s.rotation().y = se3[7 * i + 4];
    s.rotation().z = se3[7 * i + 5]; s.rotation().w = se3[7 * i + 6];
    bad += checker.isValid(&s) != (expected[i] != 0);
  }
  // HeightMapBoxChecker at the dPose boundary: an identity-rotation torso far above the map never hits
  HeightMapBoxChecker box(gpu, ARTP_SLOT_BODY, 1.31f, 0.65f, 0.3f);
  box.setHeightField(map, "elevation");
  HeightMapBoxChecker::dPose high;
  high.origin = {0.f, 0.f, 50.f, 0.f};
  bad += box.checkCollision({high}) != 0;
  // BatchPRM compiles against the same Params; it needs the sampler layers, which this fixture does not
  // carry: building a roadmap must fail loudly, not fall back to anything
  {
    BatchPRM prm(params, gpu);
    ob::SE3StateSpace::StateType a;
    a.setXYZ(se3[0], se3[1], se3[2]);
    a.rotation().x = se3[3]; a.rotation().y = se3[4]; a.rotation().z = se3[5]; a.rotation().w = se3[6];
    bool threw = false;
    try {
      prm.sampleGraph(a, a);
    } catch (const std::exception&) {
      threw = true;
    }
    bad += threw ? 0 : 1;
  }
  std::printf("host mirror: %d states batch + %d single, %d mismatches\n", n, n_single, bad);
  return bad == 0 ? 0 : 1;
}
