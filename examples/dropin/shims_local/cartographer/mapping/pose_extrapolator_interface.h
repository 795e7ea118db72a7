// Stand-in for mapping/pose_extrapolator_interface.h (+ pose_extrapolator.cc's factory) in the
// local-trajectory-builder build: the reference's abstract interface with the same members
// (pose_extrapolator_interface.h:36-77) and, behind CreateWithImuData, a small extrapolator of
// ours instead of the reference's ImuTracker-based one (Eigen geometry this image does not have):
// linear velocity from the two newest poses, angular velocity = the newest gyro reading, gravity
// = the newest accelerometer reading.  BOTH builds of the test (reference matchers, MI355X
// matchers) use it, so the comparison between them does not rest on it.
#ifndef DROPIN_SHIMS_LOCAL_POSE_EXTRAPOLATOR_INTERFACE_H_
#define DROPIN_SHIMS_LOCAL_POSE_EXTRAPOLATOR_INTERFACE_H_
#include <cmath>
#include <deque>
#include <memory>
#include <vector>
#include "Eigen/Core"
#include "Eigen/Geometry"
#include "cartographer/common/time.h"
#include "cartographer/mapping/proto/pose_extrapolator_options.pb.h"
#include "cartographer/sensor/imu_data.h"
#include "cartographer/sensor/odometry_data.h"
#include "cartographer/transform/rigid_transform.h"
#include "cartographer/transform/timestamped_transform.h"
#include "cartographer/transform/transform.h"
namespace cartographer { namespace mapping {
class PoseExtrapolatorInterface {
 public:
  struct ExtrapolationResult {
    std::vector<transform::Rigid3f> previous_poses;   // the requested times but the last
    transform::Rigid3d current_pose;                  // the last requested time
    Eigen::Vector3d current_velocity;
    Eigen::Quaterniond gravity_from_tracking;
  };
  PoseExtrapolatorInterface(const PoseExtrapolatorInterface&) = delete;
  PoseExtrapolatorInterface& operator=(const PoseExtrapolatorInterface&) = delete;
  virtual ~PoseExtrapolatorInterface() {}

  static std::unique_ptr<PoseExtrapolatorInterface> CreateWithImuData(
      const proto::PoseExtrapolatorOptions& options, const std::vector<sensor::ImuData>& imu_data,
      const std::vector<transform::TimestampedTransform>& initial_poses);

  virtual common::Time GetLastPoseTime() const = 0;
  virtual common::Time GetLastExtrapolatedTime() const = 0;
  virtual void AddPose(common::Time time, const transform::Rigid3d& pose) = 0;
  virtual void AddImuData(const sensor::ImuData& imu_data) = 0;
  virtual void AddOdometryData(const sensor::OdometryData& odometry_data) = 0;
  virtual transform::Rigid3d ExtrapolatePose(common::Time time) = 0;
  virtual ExtrapolationResult ExtrapolatePosesWithGravity(
      const std::vector<common::Time>& times) = 0;
  virtual Eigen::Quaterniond EstimateGravityOrientation(common::Time time) = 0;

 protected:
  PoseExtrapolatorInterface() {}
};

class ConstantVelocityExtrapolatorStandIn : public PoseExtrapolatorInterface {
 public:
  explicit ConstantVelocityExtrapolatorStandIn(common::Duration pose_queue_duration)
      : pose_queue_duration_(pose_queue_duration) {}
  common::Time GetLastPoseTime() const override {
    return queue_.empty() ? common::Time::min() : queue_.back().time;
  }
  common::Time GetLastExtrapolatedTime() const override { return last_extrapolated_time_; }
  void AddPose(common::Time time, const transform::Rigid3d& pose) override {
    queue_.push_back(TimedPose{time, pose});
    while (queue_.size() > 2) queue_.pop_front();
    last_extrapolated_time_ = time;
    if (queue_.size() < 2) return;
    const double dt = common::ToSeconds(time - queue_.front().time);
    if (dt < common::ToSeconds(pose_queue_duration_)) return;
    linear_velocity_ = (pose.translation() - queue_.front().pose.translation()) / dt;
  }
  void AddImuData(const sensor::ImuData& imu_data) override {
    angular_velocity_ = imu_data.angular_velocity;
    linear_acceleration_ = imu_data.linear_acceleration;
  }
  void AddOdometryData(const sensor::OdometryData&) override {}
  transform::Rigid3d ExtrapolatePose(common::Time time) override {
    const TimedPose& newest = queue_.back();
    CHECK_GE(time, newest.time);
    last_extrapolated_time_ = time;
    const double dt = common::ToSeconds(time - newest.time);
    const Eigen::Vector3d turn = dt * angular_velocity_;
    return transform::Rigid3d(
        newest.pose.translation() + dt * linear_velocity_,
        (newest.pose.rotation() * transform::AngleAxisVectorToRotationQuaternion(turn)).normalized());
  }
  ExtrapolationResult ExtrapolatePosesWithGravity(const std::vector<common::Time>& times) override {
    std::vector<transform::Rigid3f> poses;
    for (size_t i = 0; i + 1 < times.size(); ++i)
      poses.push_back(ExtrapolatePose(times[i]).cast<float>());
    const transform::Rigid3d current = ExtrapolatePose(times.back());
    return ExtrapolationResult{poses, current, linear_velocity_,
                               EstimateGravityOrientation(times.back())};
  }
  // The rotation that takes the measured "up" (the accelerometer reading at rest) to +z.
  Eigen::Quaterniond EstimateGravityOrientation(common::Time) override {
    const Eigen::Vector3d up = linear_acceleration_.normalized();
    const Eigen::Vector3d axis(up.y(), -up.x(), 0.);                    // up x (0, 0, 1)
    const double sine = axis.norm(), cosine = up.z();
    if (sine < 1e-12) return Eigen::Quaterniond::Identity();
    return Eigen::Quaterniond(Eigen::AngleAxisd(std::atan2(sine, cosine), axis / sine));
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
  Eigen::Vector3d angular_velocity_ = Eigen::Vector3d::Zero();
  Eigen::Vector3d linear_acceleration_ = Eigen::Vector3d::UnitZ();
};

// pose_extrapolator.cc:42-53 (InitializeWithImu): the first pose is the gravity alignment at the
// newest IMU packet, at the origin.
inline std::unique_ptr<PoseExtrapolatorInterface> PoseExtrapolatorInterface::CreateWithImuData(
    const proto::PoseExtrapolatorOptions& options, const std::vector<sensor::ImuData>& imu_data,
    const std::vector<transform::TimestampedTransform>& /*initial_poses*/) {
  CHECK(!imu_data.empty());
  CHECK(!options.use_imu_based());
  auto extrapolator = std::make_unique<ConstantVelocityExtrapolatorStandIn>(
      common::FromSeconds(options.constant_velocity().pose_queue_duration()));
  extrapolator->AddImuData(imu_data.back());
  extrapolator->AddPose(imu_data.back().time,
                        transform::Rigid3d::Rotation(
                            extrapolator->EstimateGravityOrientation(imu_data.back().time)));
  return extrapolator;
}
} }
#endif  // DROPIN_SHIMS_LOCAL_POSE_EXTRAPOLATOR_INTERFACE_H_
