// Stand-in for sensor/point_cloud.h + rangefinder_point.h: a vector of positions and
// TransformPointCloud (sensor/point_cloud.cc:56-64: every point through transform * point).
#ifndef ORACLE_REF_SHIMS_POINT_CLOUD_H_
#define ORACLE_REF_SHIMS_POINT_CLOUD_H_
#include <vector>
#include "Eigen/Core"
#include "cartographer/transform/rigid_transform.h"
#include "glog/logging.h"   // the real header pulls glog in; rotational_scan_matcher.cc relies on it
namespace cartographer {
namespace sensor {
struct RangefinderPoint {
  Eigen::Vector3f position;
};
class PointCloud {
 public:
  PointCloud() = default;
  explicit PointCloud(std::vector<RangefinderPoint> points) : points_(std::move(points)) {}
  PointCloud(std::vector<RangefinderPoint> points, std::vector<float> intensities)
      : points_(std::move(points)), intensities_(std::move(intensities)) {}
  // sensor/point_cloud.h:59-84 (no intensities in this build's fixtures).
  template <class UnaryPredicate>
  PointCloud copy_if(UnaryPredicate predicate) const {
    std::vector<RangefinderPoint> points;
    for (const RangefinderPoint& p : points_)
      if (predicate(p)) points.push_back(p);
    return PointCloud(std::move(points));
  }
  size_t size() const { return points_.size(); }
  bool empty() const { return points_.empty(); }
  const RangefinderPoint& operator[](size_t i) const { return points_[i]; }
  std::vector<RangefinderPoint>::const_iterator begin() const { return points_.begin(); }
  std::vector<RangefinderPoint>::const_iterator end() const { return points_.end(); }
  const std::vector<RangefinderPoint>& points() const { return points_; }
  const std::vector<float>& intensities() const { return intensities_; }
  void push_back(RangefinderPoint p) { points_.push_back(p); }
 private:
  std::vector<RangefinderPoint> points_;
  std::vector<float> intensities_;   // always empty in this build's fixtures
};
inline PointCloud TransformPointCloud(const PointCloud& point_cloud,
                                      const transform::Rigid3f& transform) {
  std::vector<RangefinderPoint> points;
  points.reserve(point_cloud.size());
  for (const RangefinderPoint& p : point_cloud) points.push_back({transform * p.position});
  return PointCloud(std::move(points));
}
}  // namespace sensor
}  // namespace cartographer
#endif  // ORACLE_REF_SHIMS_POINT_CLOUD_H_
