// Stand-in for the generated transform protobuf messages the grid headers name.
#ifndef ORACLE_REF_SHIMS_TRANSFORM_PB_H_
#define ORACLE_REF_SHIMS_TRANSFORM_PB_H_
namespace cartographer {
namespace transform {
namespace proto {
class Vector2d {
 public:
  double x() const { return x_; }
  double y() const { return y_; }
  void set_x(double v) { x_ = v; }
  void set_y(double v) { y_ = v; }
 private:
  double x_ = 0., y_ = 0.;
};
class Rigid3d {   // written by DrawToSubmapTexture, never read back here
 public:
  double t[3] = {0., 0., 0.};
  double q[4] = {1., 0., 0., 0.};
};
}  // namespace proto
}  // namespace transform
}  // namespace cartographer
#endif  // ORACLE_REF_SHIMS_TRANSFORM_PB_H_
