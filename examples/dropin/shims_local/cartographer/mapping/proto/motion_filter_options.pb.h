// Stand-in for the generated message of mapping/proto/motion_filter_options.proto.
#ifndef DROPIN_SHIMS_LOCAL_MOTION_FILTER_OPTIONS_PB_H_
#define DROPIN_SHIMS_LOCAL_MOTION_FILTER_OPTIONS_PB_H_
namespace cartographer { namespace mapping { namespace proto {
class MotionFilterOptions {
 public:
  double max_time_seconds() const { return max_time_seconds_; }
  double max_distance_meters() const { return max_distance_meters_; }
  double max_angle_radians() const { return max_angle_radians_; }
  void set_max_time_seconds(double v) { max_time_seconds_ = v; }
  void set_max_distance_meters(double v) { max_distance_meters_ = v; }
  void set_max_angle_radians(double v) { max_angle_radians_ = v; }
 private:
  double max_time_seconds_ = 0., max_distance_meters_ = 0., max_angle_radians_ = 0.;
};
} } }
#endif  // DROPIN_SHIMS_LOCAL_MOTION_FILTER_OPTIONS_PB_H_
