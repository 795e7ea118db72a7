// Stand-in for mapping/pose_extrapolator.h in the local-trajectory-builder build.  The
// reference's extrapolator integrates an ImuTracker (Eigen geometry this image does not have);
// this one is the planar constant-velocity model it reduces to without IMU and odometry input:
// linear velocity and yaw rate from the oldest and newest pose of the queue, the last pose advanced
// by them, gravity orientation = identity.  Same public interface
// (mapping/pose_extrapolator.h:36-66); BOTH builds of the test (reference matchers, MI355X
// matchers) use it, so the comparison between them does not rest on it.
#ifndef DROPIN_SHIMS_LOCAL_POSE_EXTRAPOLATOR_H_
#define DROPIN_SHIMS_LOCAL_POSE_EXTRAPOLATOR_H_
#include <cmath>
#include <deque>
#include "Eigen/Core"
#include "Eigen/Geometry"
#include "cartographer/common/time.h"
#include "cartographer/sensor/imu_data.h"
#include "cartographer/sensor/odometry_data.h"
#include "cartographer/transform/rigid_transform.h"
#include "cartographer/transform/transform.h"
namespace cartographer { namespace mapping {
class PoseExtrapolator {
 public:
  PoseExtrapolator(common::Duration pose_queue_duration, double /*imu_gravity_time_constant*/)
      : pose_queue_duration_(pose_queue_duration) {}
  PoseExtrapolator(const PoseExtrapolator&) = delete;
  PoseExtrapolator& operator=(const PoseExtrapolator&) = delete;

  common::Time GetLastPoseTime() const {
    return queue_.empty() ? common::Time::min() : queue_.back().time;
  }
  common::Time GetLastExtrapolatedTime() const { return last_extrapolated_time_; }

  void AddPose(common::Time time, const transform::Rigid3d& pose) {
    queue_.push_back(TimedPose{time, pose});
    while (queue_.size() > 2 && queue_[1].time <= time - pose_queue_duration_) queue_.pop_front();
    last_extrapolated_time_ = time;
    if (queue_.size() < 2) return;
    const TimedPose& oldest = queue_.front();
    const double dt = common::ToSeconds(time - oldest.time);
    if (dt < common::ToSeconds(pose_queue_duration_)) return;   // too close: keep the velocities
    linear_velocity_ = (pose.translation() - oldest.pose.translation()) / dt;
    double dyaw = transform::GetYaw(pose) - transform::GetYaw(oldest.pose);
    while (dyaw > M_PI) dyaw -= 2. * M_PI;
    while (dyaw < -M_PI) dyaw += 2. * M_PI;
    yaw_rate_ = dyaw / dt;
  }
  void AddImuData(const sensor::ImuData&) {}
  void AddOdometryData(const sensor::OdometryData&) {}

  transform::Rigid3d ExtrapolatePose(common::Time time) {
    const TimedPose& newest = queue_.back();
    CHECK_GE(time, newest.time);
    last_extrapolated_time_ = time;
    const double dt = common::ToSeconds(time - newest.time);
    const Eigen::Vector3d translation = newest.pose.translation() + dt * linear_velocity_;
    const Eigen::Quaterniond turn(Eigen::AngleAxisd(dt * yaw_rate_, Eigen::Vector3d::UnitZ()));
    return transform::Rigid3d(translation, (newest.pose.rotation() * turn).normalized());
  }
  Eigen::Quaterniond EstimateGravityOrientation(common::Time) {
    return Eigen::Quaterniond::Identity();
  }

 private:
  struct TimedPose {
    common::Time time;
    transform::Rigid3d pose;
  };
  const common::Duration pose_queue_duration_;
  std::deque<TimedPose> queue_;
  common::Time last_extrapolated_time_ = common::Time::min();
  Eigen::Vector3d linear_velocity_ = Eigen::Vector3d::Zero();
  double yaw_rate_ = 0.;
};
} }
#endif  // DROPIN_SHIMS_LOCAL_POSE_EXTRAPOLATOR_H_
