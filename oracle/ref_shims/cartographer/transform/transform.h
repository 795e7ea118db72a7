// Shadows the reference's transform/transform.h for `make ref`: the files built there include
// it for Eigen types, the Rigid transforms and glog's CHECK macros; the real header needs
// Eigen's geometry module.
#ifndef ORACLE_REF_SHIMS_CARTOGRAPHER_TRANSFORM_TRANSFORM_H_
#define ORACLE_REF_SHIMS_CARTOGRAPHER_TRANSFORM_TRANSFORM_H_
#include <algorithm>
#include <cmath>
#include "Eigen/Core"
#include "Eigen/Geometry"
#include "cartographer/common/math.h"   // the reference's own (int64, Clamp, ...)
#include "cartographer/common/port.h"
#include "cartographer/transform/rigid_transform.h"
#include "glog/logging.h"
#endif  // ORACLE_REF_SHIMS_CARTOGRAPHER_TRANSFORM_TRANSFORM_H_
