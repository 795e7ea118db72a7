// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_common.h).
// 3D correlative scan matching restated from
//   cartographer/mapping/internal/3d/scan_matching/real_time_correlative_scan_matcher_3d.cc:34-114
//   .../fast_correlative_scan_matcher_3d.{h,cc}
//   .../precomputation_grid_3d.{h,cc}
//   .../rotational_scan_matcher.cc:121-189
//   .../low_resolution_matcher.cc:23-35
// and the read accessors of mapping/3d/hybrid_grid.h.
//
// Pinned bit-for-bit against those sources compiled in place (oracle/_ref,
// tests/test_reference_ref_3d.py) over stand-in Eigen types.
//
// Eigen detail (UNPINNED): float quaternion products follow the
// SSE kernel of Eigen 3.3 (Geometry/arch/Geometry_SSE.h), which is what an
// x86-64 build of the reference uses; 4-vector squared norms reduce as
// (x^2+z^2)+(y^2+w^2) (Packet4f predux).
#ifndef ORACLE_3D_H_
#define ORACLE_3D_H_

#include <cstdint>
#include <functional>
#include <memory>
#include <vector>

#include "oracle_2d.h"
#include "oracle_common.h"

namespace oracle {

struct Cell3i { int x, y, z; };
struct Voxel { int x, y, z; uint16_t value; uint16_t pad; };

struct Qd { double w, x, y, z; };
struct Pose3d { double t[3]; Qd q; };

// Eigen 3.3 SSE float quaternion product (coeff order x,y,z,w).
inline Qf QuatMulSse(const Qf& a, const Qf& b) {
  Qf r;
  r.x = (a.x * b.w - a.z * b.y) + (a.y * b.z + a.w * b.x);
  r.y = (a.y * b.w - a.x * b.z) + (a.z * b.x + a.w * b.y);
  r.z = (a.z * b.w - a.y * b.x) + (a.x * b.y + a.w * b.z);
  r.w = (a.w * b.w - a.x * b.x) + -(a.z * b.z + a.y * b.y);
  return r;
}
inline float QuatSquaredNorm(const Qf& q) {  // Packet4f predux of (x,y,z,w)^2
  return (q.x * q.x + q.z * q.z) + (q.y * q.y + q.w * q.w);
}
inline Qf QuatNormalizedSse(const Qf& q) {
  const float z = QuatSquaredNorm(q);
  if (z > 0.f) {
    const float n = std::sqrt(z);
    return {q.w / n, q.x / n, q.y / n, q.z / n};
  }
  return q;
}
inline Qf QuatInverseSse(const Qf& q) {  // conjugate / squaredNorm
  const float n2 = QuatSquaredNorm(q);
  return {q.w / n2, -q.x / n2, -q.y / n2, -q.z / n2};
}
// Rigid3f product, transform/rigid_transform.h:183-189.
inline Rigid3f MulSse(const Rigid3f& a, const Rigid3f& b) {
  const V3f r = Rotate(a.q, b.t);
  return {{r.x + a.t.x, r.y + a.t.y, r.z + a.t.z}, QuatNormalizedSse(QuatMulSse(a.q, b.q))};
}
inline Rigid3f InverseRigid(const Rigid3f& a) {  // rigid_transform.h:151-155 (conjugate)
  const Qf c = QuatConj(a.q);
  const V3f r = Rotate(c, a.t);
  return {{-r.x, -r.y, -r.z}, c};
}
// transform/transform.h:85-99 for T = float.
Qf AngleAxisVectorToRotationQuaternion(const V3f& angle_axis);
// transform/transform.h:33-37.
float GetAngle(const Rigid3f& t);
// transform/transform.h:42-47.
float GetYaw(const Qf& q);

// Dense-brick stand-in of HybridGridBase<T> for reads: value(index) is the
// stored value or T() (hybrid_grid.h:263-279: outside / unallocated -> T()).
template <typename T>
class Brick {
 public:
  Brick() = default;
  void Reset(const Cell3i& lo, const Cell3i& hi) {   // inclusive bounds
    lo_ = lo;
    nx_ = hi.x - lo.x + 1; ny_ = hi.y - lo.y + 1; nz_ = hi.z - lo.z + 1;
    data_.assign(static_cast<size_t>(nx_) * ny_ * nz_, T());
  }
  bool empty() const { return data_.empty(); }
  T value(int x, int y, int z) const {
    const int ix = x - lo_.x, iy = y - lo_.y, iz = z - lo_.z;
    if (static_cast<unsigned>(ix) >= static_cast<unsigned>(nx_) ||
        static_cast<unsigned>(iy) >= static_cast<unsigned>(ny_) ||
        static_cast<unsigned>(iz) >= static_cast<unsigned>(nz_)) return T();
    return data_[(static_cast<size_t>(iz) * ny_ + iy) * nx_ + ix];
  }
  T* mutable_value(int x, int y, int z) {
    return &data_[(static_cast<size_t>(z - lo_.z) * ny_ + (y - lo_.y)) * nx_ + (x - lo_.x)];
  }
  Cell3i lo() const { return lo_; }
  Cell3i hi() const { return {lo_.x + nx_ - 1, lo_.y + ny_ - 1, lo_.z + nz_ - 1}; }
  template <typename F>
  void ForEachNonZero(F f) const {
    size_t i = 0;
    for (int z = 0; z < nz_; ++z)
      for (int y = 0; y < ny_; ++y)
        for (int x = 0; x < nx_; ++x, ++i)
          if (data_[i] != T()) f(Cell3i{x + lo_.x, y + lo_.y, z + lo_.z}, data_[i]);
  }
 private:
  Cell3i lo_{0, 0, 0};
  int nx_ = 0, ny_ = 0, nz_ = 0;
  std::vector<T> data_;
};

// hybrid_grid.h:414-460, read side.
class HybridGridView {
 public:
  HybridGridView(float resolution, const Voxel* voxels, int64_t n);
  float resolution() const { return resolution_; }
  int grid_size() const { return grid_size_; }   // hybrid_grid.h:259 after all writes
  Cell3i GetCellIndex(const V3f& p) const {      // hybrid_grid.h:428-433, f32 divide
    return {RoundToInt(p.x / resolution_), RoundToInt(p.y / resolution_),
            RoundToInt(p.z / resolution_)};
  }
  uint16_t value(const Cell3i& c) const { return cells_.value(c.x, c.y, c.z); }
  float GetProbability(const Cell3i& c) const { return ValueToProbability(value(c)); }
  const Brick<uint16_t>& cells() const { return cells_; }
 private:
  float resolution_;
  int grid_size_;
  Brick<uint16_t> cells_;
};

typedef std::vector<V3f> PointCloud3;

// real_time_correlative_scan_matcher_3d.cc:34-114.
float RealTimeMatch3D(const HybridGridView& grid, const Pose3d& initial, const PointCloud3& cloud,
                      double linear_window, double angular_window, double tw, double rw,
                      Pose3d* pose_estimate, int64_t* num_candidates, int num_threads = 1);

// precomputation_grid_3d.{h,cc}.
typedef Brick<uint8_t> PrecomputationGrid3D;
PrecomputationGrid3D ConvertToPrecomputationGrid(const HybridGridView& grid);
PrecomputationGrid3D PrecomputeGrid(const PrecomputationGrid3D& grid, bool half_resolution,
                                    int shift);
inline float ToProbability3D(float value) {   // precomputation_grid_3d.h:32-35
  return kMinProbability + value * ((kMaxProbability - kMinProbability) / 255.f);
}

// rotational_scan_matcher.cc:141-189.
std::vector<float> RotateHistogram(const std::vector<float>& histogram, float angle);
std::vector<float> RotationalMatch(const std::vector<float>& submap_histogram,
                                   const std::vector<float>& scan_histogram, float initial_angle,
                                   const std::vector<float>& angles);

struct Fast3DOptions {
  int branch_and_bound_depth;
  int full_resolution_depth;
  double min_rotational_score;
  double min_low_resolution_score;
  double linear_xy_search_window;
  double linear_z_search_window;
  double angular_search_window;
};

struct NodeData3D {      // TrajectoryNode::Data, mapping/trajectory_node.h:45-63
  Qd gravity_alignment;
  PointCloud3 high_resolution_point_cloud;
  PointCloud3 low_resolution_point_cloud;
  std::vector<float> rotational_scan_matcher_histogram;
};

struct Result3D {
  float score;
  Pose3d pose_estimate;
  float rotational_score;
  float low_resolution_score;
};

struct Stats3D {
  int64_t candidates_scored = 0, coarse_candidates = 0, nodes_expanded = 0, num_scans = 0;
};

// fast_correlative_scan_matcher_3d.{h,cc}.
class FastCorrelativeScanMatcher3D {
 public:
  FastCorrelativeScanMatcher3D(std::shared_ptr<HybridGridView> grid,
                               std::shared_ptr<HybridGridView> low_resolution_grid,
                               std::vector<float> histogram, const Fast3DOptions& options);
  bool Match(const Pose3d& global_node_pose, const Pose3d& global_submap_pose,
             const NodeData3D& data, float min_score, Result3D* result, Stats3D* stats) const;
  bool MatchFullSubmap(const Qd& global_node_rotation, const Qd& global_submap_rotation,
                       const NodeData3D& data, float min_score, Result3D* result,
                       Stats3D* stats) const;
  const PrecomputationGrid3D& level(int depth) const { return stack_[depth]; }
  int depth() const { return static_cast<int>(stack_.size()); }

 private:
  struct DiscreteScan3D {
    Rigid3f pose;
    std::vector<std::vector<Cell3i>> cell_indices_per_depth;
    float rotational_score;
  };
  struct Candidate3D {
    int scan_index;
    Cell3i offset;
    float score;
    float low_resolution_score;
    bool operator<(const Candidate3D& o) const { return score < o.score; }
    bool operator>(const Candidate3D& o) const { return score > o.score; }
  };
  struct Search {
    int linear_xy_window_size, linear_z_window_size;
    double angular_search_window;
    const PointCloud3* low_resolution_cloud;
  };
  bool MatchWithSearchParameters(const Search& sp, const Rigid3f& node, const Rigid3f& submap,
                                 const NodeData3D& data, float min_score, Result3D* result,
                                 Stats3D* stats) const;
  DiscreteScan3D DiscretizeScan(const Search& sp, const PointCloud3& cloud, const Rigid3f& pose,
                                float rotational_score) const;
  std::vector<DiscreteScan3D> GenerateDiscreteScans(const Search& sp, const NodeData3D& data,
                                                    const Rigid3f& node,
                                                    const Rigid3f& submap) const;
  void ScoreCandidates(int depth, const std::vector<DiscreteScan3D>& scans,
                       std::vector<Candidate3D>* candidates, Stats3D* stats) const;
  Candidate3D BranchAndBound(const Search& sp, const std::vector<DiscreteScan3D>& scans,
                             const std::vector<Candidate3D>& candidates, int depth,
                             float min_score, Stats3D* stats) const;
  Rigid3f GetPoseFromCandidate(const std::vector<DiscreteScan3D>& scans,
                               const Candidate3D& c) const;
  float LowResolutionScore(const Search& sp, const Rigid3f& pose) const;

  Fast3DOptions options_;
  float resolution_;
  int width_in_voxels_;
  std::shared_ptr<HybridGridView> grid_, low_grid_;
  std::vector<float> histogram_;
  std::vector<PrecomputationGrid3D> stack_;
};

}  // namespace oracle

#endif  // ORACLE_3D_H_
