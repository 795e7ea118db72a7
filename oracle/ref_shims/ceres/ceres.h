// Placeholder: common/math.h includes ceres for an atan2 template that is never instantiated
// by the files `make ref` builds.
#ifndef ORACLE_REF_SHIMS_CERES_H_
#define ORACLE_REF_SHIMS_CERES_H_
#include <cmath>
namespace ceres {
template <typename T>
T atan2(const T& y, const T& x) { return std::atan2(y, x); }
}  // namespace ceres
#endif  // ORACLE_REF_SHIMS_CERES_H_
