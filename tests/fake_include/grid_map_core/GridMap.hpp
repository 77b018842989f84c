// TEST SCAFFOLD, not grid_map: grid_map::GridMap as far as the host mirror's -DARTP_HAVE_GRID_MAP branch touches it
// (Planner::setMap(std::unique_ptr<grid_map::GridMap>&&), planner.h:65; art_planner::Map::getMap(), map.h), written
// from the published grid_map_core API: column-major float layers (grid_map::Matrix = Eigen::MatrixXf), size /
// length / position as 2-vectors read with operator()(i).
#pragma once
#include <map>
#include <string>
#include <vector>
#include <Eigen/Dense>
namespace grid_map {
using Matrix = Eigen::Matrix<float, Eigen::Dynamic, Eigen::Dynamic>;
template <class T>
struct Pair2 {
  T v[2]{T(), T()};
  Pair2() = default;
  Pair2(T a, T b) { v[0] = a; v[1] = b; }
  T& operator()(int i) { return v[i]; }
  const T& operator()(int i) const { return v[i]; }
  T x() const { return v[0]; }
  T y() const { return v[1]; }
};
using Size = Pair2<int>;
using Length = Pair2<double>;
using Position = Pair2<double>;
class GridMap {
 public:
  GridMap() = default;
  explicit GridMap(const std::vector<std::string>& layers) {
    for (const auto& l : layers) add(l);
  }
  void setGeometry(const Length& length, double resolution, const Position& position = Position(0.0, 0.0)) {
    length_ = length;
    resolution_ = resolution;
    position_ = position;
    size_ = Size(static_cast<int>(length(0) / resolution + 0.5), static_cast<int>(length(1) / resolution + 0.5));
    for (auto& kv : data_) kv.second.resize(size_(0), size_(1));
  }
  void add(const std::string& layer, float value = 0.0f) {
    Matrix m(size_(0), size_(1));
    for (Eigen::Index i = 0; i < m.rows() * m.cols(); ++i) m.data()[i] = value;
    add(layer, m);
  }
  void add(const std::string& layer, const Matrix& data) {
    if (!exists(layer)) layers_.push_back(layer);
    data_[layer] = data;
  }
  bool exists(const std::string& layer) const { return data_.count(layer) != 0; }
  const Matrix& get(const std::string& layer) const { return data_.at(layer); }
  Matrix& get(const std::string& layer) { return data_.at(layer); }
  const Matrix& operator[](const std::string& layer) const { return data_.at(layer); }
  Matrix& operator[](const std::string& layer) { return data_.at(layer); }
  const std::vector<std::string>& getLayers() const { return layers_; }
  const Size& getSize() const { return size_; }
  const Length& getLength() const { return length_; }
  const Position& getPosition() const { return position_; }
  double getResolution() const { return resolution_; }

 private:
  Size size_;
  Length length_;
  Position position_;
  double resolution_{0.0};
  std::vector<std::string> layers_;
  std::map<std::string, Matrix> data_;
};
}  // namespace grid_map
