// art_planner::Map -- the accessors of the reference's grid_map wrapper the hot path uses
// (art_planner/include/art_planner/map/map.h:59-164), over plain column-major float layers
// (grid_map::Matrix storage) because grid_map_core is not installed here.
#pragma once

#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#ifdef ARTP_HAVE_GRID_MAP
#include <grid_map_core/GridMap.hpp>
#endif

namespace art_planner {

class Map {
 public:
  Map() = default;
#ifdef ARTP_HAVE_GRID_MAP
  // The reference's Map owns the grid_map::GridMap (map.h:59-67) and hands it out with getMap() (PlannerRos publishes
  // it, planner_ros.cpp:339).  With grid_map_core available the wrapper keeps the grid map too and mirrors its geometry
  // and layers (grid_map::Matrix is column-major rows x cols: the storage the C ABI takes) into the plain form.
  explicit Map(std::unique_ptr<grid_map::GridMap>&& gm) : grid_map_(std::move(gm)) {
    const auto& size = grid_map_->getSize();
    const auto& len = grid_map_->getLength();
    const auto& pos = grid_map_->getPosition();
    geom_ = Geometry{size(0), size(1), grid_map_->getResolution(), len(0), len(1), pos(0), pos(1)};
    for (const auto& name : grid_map_->getLayers()) {
      const grid_map::Matrix& m = grid_map_->get(name);
      layers_[name].assign(m.data(), m.data() + static_cast<size_t>(geom_.rows) * geom_.cols);
    }
  }
  grid_map::GridMap& getMap() { return *grid_map_; }
  const grid_map::GridMap& getMap() const { return *grid_map_; }
  bool hasGridMap() const { return static_cast<bool>(grid_map_); }
#endif

  struct Geometry {
    int rows{0}, cols{0};           // grid_map size (x, y)
    double resolution{0.0};
    double length_x{0.0}, length_y{0.0};
    double position_x{0.0}, position_y{0.0};
  };

  void setGeometry(const Geometry& g) {
    std::lock_guard<std::mutex> lock(mutex_);
    geom_ = g;
  }
  Geometry getGeometry() const {
    std::lock_guard<std::mutex> lock(mutex_);
    return geom_;
  }
  // layer = Eigen::MatrixXf storage: column-major rows x cols
  void addLayer(const std::string& name, const float* data) {
    std::lock_guard<std::mutex> lock(mutex_);
    layers_[name].assign(data, data + static_cast<size_t>(geom_.rows) * geom_.cols);
  }
  bool exists(const std::string& name) const {
    std::lock_guard<std::mutex> lock(mutex_);
    return layers_.count(name) != 0;
  }
  const std::vector<float>& getLayer(const std::string& name) const {
    std::lock_guard<std::mutex> lock(mutex_);
    return layers_.at(name);
  }
  // grid_map isInside (checkIfPositionWithinMap)
  bool isInside(double x, double y) const {
    std::lock_guard<std::mutex> lock(mutex_);
    const double tx = -((x - geom_.position_x) - 0.5 * geom_.length_x);
    const double ty = -((y - geom_.position_y) - 0.5 * geom_.length_y);
    return tx >= 0.0 && ty >= 0.0 && tx < geom_.length_x && ty < geom_.length_y;
  }

 private:
  mutable std::mutex mutex_;
  Geometry geom_;
  std::map<std::string, std::vector<float>> layers_;
#ifdef ARTP_HAVE_GRID_MAP
  std::unique_ptr<grid_map::GridMap> grid_map_;
#endif
};

using MapPtr = std::shared_ptr<Map>;

}  // namespace art_planner
