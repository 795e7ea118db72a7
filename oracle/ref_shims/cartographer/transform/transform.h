// Shadows the reference's transform/transform.h for `make ref`: ray_to_pixel_mask.h includes it
// only to get Eigen::Array2i and glog's CHECK macros; the real header needs Eigen's geometry
// module.
#ifndef ORACLE_REF_SHIMS_CARTOGRAPHER_TRANSFORM_TRANSFORM_H_
#define ORACLE_REF_SHIMS_CARTOGRAPHER_TRANSFORM_TRANSFORM_H_
#include <algorithm>
#include <cmath>
#include "Eigen/Core"
#include "cartographer/common/math.h"   // the reference's own (int64, Clamp, ...)
#include "cartographer/common/port.h"
#include "glog/logging.h"
#endif  // ORACLE_REF_SHIMS_CARTOGRAPHER_TRANSFORM_TRANSFORM_H_
