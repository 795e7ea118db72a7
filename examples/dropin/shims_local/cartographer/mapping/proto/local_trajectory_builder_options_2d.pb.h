// Stand-in for the generated message of mapping/proto/local_trajectory_builder_options_2d.proto:
// every field local_trajectory_builder_2d.cc reads, under the generated accessors' names.
#ifndef DROPIN_SHIMS_LOCAL_LOCAL_TRAJECTORY_BUILDER_OPTIONS_2D_PB_H_
#define DROPIN_SHIMS_LOCAL_LOCAL_TRAJECTORY_BUILDER_OPTIONS_2D_PB_H_
#include "cartographer/mapping/proto/motion_filter_options.pb.h"
#include "cartographer/mapping/proto/pose_extrapolator_options.pb.h"
#include "cartographer/mapping/proto/scan_matching/ceres_scan_matcher_options_2d.pb.h"
#include "cartographer/mapping/proto/scan_matching/real_time_correlative_scan_matcher_options.pb.h"
#include "cartographer/mapping/proto/submaps_options_2d.pb.h"
#include "cartographer/sensor/proto/adaptive_voxel_filter_options.pb.h"
namespace cartographer { namespace mapping { namespace proto {
class LocalTrajectoryBuilderOptions2D {
 public:
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
  DROPIN_FIELD(float, min_z)
  DROPIN_FIELD(float, max_z)
  DROPIN_FIELD(float, missing_data_ray_length)
  DROPIN_FIELD(int, num_accumulated_range_data)
  DROPIN_FIELD(float, voxel_filter_size)
  DROPIN_FIELD(bool, use_online_correlative_scan_matching)
  DROPIN_FIELD(bool, use_imu_data)
  DROPIN_MESSAGE(sensor::proto::AdaptiveVoxelFilterOptions, adaptive_voxel_filter_options)
  DROPIN_MESSAGE(scan_matching::proto::RealTimeCorrelativeScanMatcherOptions,
                 real_time_correlative_scan_matcher_options)
  DROPIN_MESSAGE(scan_matching::proto::CeresScanMatcherOptions2D, ceres_scan_matcher_options)
  DROPIN_MESSAGE(MotionFilterOptions, motion_filter_options)
  DROPIN_MESSAGE(PoseExtrapolatorOptions, pose_extrapolator_options)
  DROPIN_MESSAGE(SubmapsOptions2D, submaps_options)
#undef DROPIN_FIELD
#undef DROPIN_MESSAGE
};
} } }
#endif  // DROPIN_SHIMS_LOCAL_LOCAL_TRAJECTORY_BUILDER_OPTIONS_2D_PB_H_
