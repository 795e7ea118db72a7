// Stand-in for the generated messages of mapping/proto/pose_extrapolator_options.proto: the
// fields LocalTrajectoryBuilder2D::InitializeExtrapolator reads.
#ifndef DROPIN_SHIMS_LOCAL_POSE_EXTRAPOLATOR_OPTIONS_PB_H_
#define DROPIN_SHIMS_LOCAL_POSE_EXTRAPOLATOR_OPTIONS_PB_H_
namespace cartographer { namespace mapping { namespace proto {
class ConstantVelocityPoseExtrapolatorOptions {
 public:
  double pose_queue_duration() const { return pose_queue_duration_; }
  double imu_gravity_time_constant() const { return imu_gravity_time_constant_; }
  void set_pose_queue_duration(double v) { pose_queue_duration_ = v; }
  void set_imu_gravity_time_constant(double v) { imu_gravity_time_constant_ = v; }
 private:
  double pose_queue_duration_ = 0., imu_gravity_time_constant_ = 0.;
};
class PoseExtrapolatorOptions {
 public:
  bool use_imu_based() const { return use_imu_based_; }
  void set_use_imu_based(bool v) { use_imu_based_ = v; }
  const ConstantVelocityPoseExtrapolatorOptions& constant_velocity() const {
    return constant_velocity_;
  }
  ConstantVelocityPoseExtrapolatorOptions* mutable_constant_velocity() {
    return &constant_velocity_;
  }
 private:
  bool use_imu_based_ = false;
  ConstantVelocityPoseExtrapolatorOptions constant_velocity_;
};
} } }
#endif  // DROPIN_SHIMS_LOCAL_POSE_EXTRAPOLATOR_OPTIONS_PB_H_
