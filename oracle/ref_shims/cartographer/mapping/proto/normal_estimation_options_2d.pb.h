// Stand-in for the generated protobuf options message.
#ifndef ORACLE_REF_SHIMS_NORMAL_ESTIMATION_OPTIONS_2D_PB_H_
#define ORACLE_REF_SHIMS_NORMAL_ESTIMATION_OPTIONS_2D_PB_H_
namespace cartographer {
namespace mapping {
namespace proto {
class NormalEstimationOptions2D {
 public:
  int num_normal_samples() const { return num_normal_samples_; }
  double sample_radius() const { return sample_radius_; }
  void set_num_normal_samples(int v) { num_normal_samples_ = v; }
  void set_sample_radius(double v) { sample_radius_ = v; }
 private:
  int num_normal_samples_ = 0;
  double sample_radius_ = 0.;
};
}  // namespace proto
}  // namespace mapping
}  // namespace cartographer
#endif  // ORACLE_REF_SHIMS_NORMAL_ESTIMATION_OPTIONS_2D_PB_H_
