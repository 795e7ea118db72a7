// Stand-in for ceres/rotation.h: the one function the files built here call.
#ifndef ORACLE_REF_SHIMS_CERES_ROTATION_H_
#define ORACLE_REF_SHIMS_CERES_ROTATION_H_
#include "ceres/jet.h"
namespace ceres {
// zw = z * w (Hamilton product, (w, x, y, z) storage), as published in rotation.h.
template <typename T>
inline void QuaternionProduct(const T z[4], const T w[4], T zw[4]) {
  zw[0] = z[0] * w[0] - z[1] * w[1] - z[2] * w[2] - z[3] * w[3];
  zw[1] = z[0] * w[1] + z[1] * w[0] + z[2] * w[3] - z[3] * w[2];
  zw[2] = z[0] * w[2] - z[1] * w[3] + z[2] * w[0] + z[3] * w[1];
  zw[3] = z[0] * w[3] + z[1] * w[2] - z[2] * w[1] + z[3] * w[0];
}
}  // namespace ceres
#endif  // ORACLE_REF_SHIMS_CERES_ROTATION_H_
