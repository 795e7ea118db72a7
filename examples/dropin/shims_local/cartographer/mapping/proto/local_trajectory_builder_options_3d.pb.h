// Stand-in for the generated message of mapping/proto/local_trajectory_builder_options_3d.proto:
// every field local_trajectory_builder_3d.cc reads, under the generated accessors' names.
#ifndef DROPIN_SHIMS_LOCAL_LOCAL_TRAJECTORY_BUILDER_OPTIONS_3D_PB_H_
#define DROPIN_SHIMS_LOCAL_LOCAL_TRAJECTORY_BUILDER_OPTIONS_3D_PB_H_
#include <vector>
#include "cartographer/mapping/proto/motion_filter_options.pb.h"
#include "cartographer/mapping/proto/pose_extrapolator_options.pb.h"
#include "cartographer/mapping/proto/scan_matching/ceres_scan_matcher_options_3d.pb.h"
#include "cartographer/mapping/proto/scan_matching/real_time_correlative_scan_matcher_options.pb.h"
#include "cartographer/mapping/proto/submaps_options_3d.pb.h"
#include "cartographer/sensor/imu_data.h"
#include "cartographer/sensor/proto/adaptive_voxel_filter_options.pb.h"
#include "cartographer/transform/timestamped_transform.h"
namespace cartographer { namespace mapping { namespace proto {
class LocalTrajectoryBuilderOptions3D {
#define DROPIN_FIELD(type, name)                  \
 public:                                          \
  type name() const { return name##_; }           \
  void set_##name(type v) { name##_ = v; }        \
 private:                                         \
  type name##_ = type();
#define DROPIN_MESSAGE(type, name)                \
 public:                                          \
  const type& name() const { return name##_; }    \
  type* mutable_##name() { return &name##_; }     \
 private:                                         \
  type name##_;
  DROPIN_FIELD(float, min_range)
  DROPIN_FIELD(float, max_range)
  DROPIN_FIELD(int, num_accumulated_range_data)
  DROPIN_FIELD(float, voxel_filter_size)
  DROPIN_FIELD(bool, use_online_correlative_scan_matching)
  DROPIN_FIELD(bool, use_intensities)
  DROPIN_FIELD(int, rotational_histogram_size)
  DROPIN_MESSAGE(sensor::proto::AdaptiveVoxelFilterOptions,
                 high_resolution_adaptive_voxel_filter_options)
  DROPIN_MESSAGE(sensor::proto::AdaptiveVoxelFilterOptions,
                 low_resolution_adaptive_voxel_filter_options)
  DROPIN_MESSAGE(scan_matching::proto::RealTimeCorrelativeScanMatcherOptions,
                 real_time_correlative_scan_matcher_options)
  DROPIN_MESSAGE(scan_matching::proto::CeresScanMatcherOptions3D, ceres_scan_matcher_options)
  DROPIN_MESSAGE(MotionFilterOptions, motion_filter_options)
  DROPIN_MESSAGE(PoseExtrapolatorOptions, pose_extrapolator_options)
  DROPIN_MESSAGE(SubmapsOptions3D, submaps_options)
  // repeated fields, empty here: the extrapolator starts from the first IMU packet alone
  DROPIN_MESSAGE(std::vector<transform::proto::TimestampedTransform>, initial_poses)
  DROPIN_MESSAGE(std::vector<sensor::proto::ImuData>, initial_imu_data)
#undef DROPIN_FIELD
#undef DROPIN_MESSAGE
};
} } }
#endif  // DROPIN_SHIMS_LOCAL_LOCAL_TRAJECTORY_BUILDER_OPTIONS_3D_PB_H_
