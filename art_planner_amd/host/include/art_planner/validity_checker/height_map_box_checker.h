// art_planner::HeightMapBoxChecker with the reference's interface
// (art_planner/include/art_planner/validity_checker/height_map_box_checker.h:18-61), backed by the HIP
// box-vs-heightfield kernels instead of ODE.
#pragma once

#include <array>
#include <memory>
#include <string>
#include <vector>

#include "art_planner/gpu_context.h"
#include "art_planner/map/map.h"

namespace art_planner {

class HeightMapBoxChecker {
 public:
  struct dPose {  // same layout as the reference (origin[4], rotation[12], dReal = float)
    std::array<float, 4> origin{0, 0, 0, 0};
    std::array<float, 12> rotation{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
  };
  static_assert(sizeof(dPose) == 16 * sizeof(float), "dPose must be 16 packed floats");

  HeightMapBoxChecker(const GpuContextPtr& gpu, int slot, float length_x, float length_y, float length_z)
      : gpu_(gpu), slot_(slot), lengths_{length_x, length_y, length_z} {}

  // height_map_box_checker.cpp:38-54
  void setHeightField(std::shared_ptr<Map> map, const std::string& layer_name) {
    const auto g = map->getGeometry();
    const auto& layer = map->getLayer(layer_name);
    throwOnError(gpu_->get(), artp_upload_layer(gpu_->get(), slot_, layer.data(), g.rows, g.cols, g.length_x,
                                                g.length_y, g.position_x, g.position_y), "artp_upload_layer");
    gpu_->mapChanged();
  }

  // height_map_box_checker.cpp:58-72: number of poses in contact
  int checkCollision(const std::vector<dPose>& box_poses) const {
    if (box_poses.empty()) return 0;
    std::vector<uint8_t> hit(box_poses.size());
    throwOnError(gpu_->get(), artp_check_boxes(gpu_->get(), slot_, lengths_.data(), box_poses[0].origin.data(),
                                               box_poses.size(), hit.data(), nullptr), "artp_check_boxes");
    int n = 0;
    for (uint8_t h : hit) n += h ? 1 : 0;
    return n;
  }

 private:
  GpuContextPtr gpu_;
  int slot_;
  std::array<float, 3> lengths_;
};

}  // namespace art_planner
