// Stand-in for ceres/ceres.h in the drop-in build: ConstraintBuilder2D only names
// ceres::Solver::Summary (an output it ignores).
#ifndef DROPIN_SHIMS_CERES_H_
#define DROPIN_SHIMS_CERES_H_
#include <cmath>
namespace ceres {
template <typename T>
T atan2(const T& y, const T& x) { return std::atan2(y, x); }
struct Solver {
  struct Summary {
    double initial_cost = 0., final_cost = 0.;
    int num_successful_steps = 0, num_unsuccessful_steps = 0;
    int termination_type = 1;
  };
};
}  // namespace ceres
#endif  // DROPIN_SHIMS_CERES_H_
