// Stand-in for the generated protobuf message (common/proto/ceres_solver_options.proto).
#ifndef ORACLE_REF_SHIMS_CERES_SOLVER_OPTIONS_PB_H_
#define ORACLE_REF_SHIMS_CERES_SOLVER_OPTIONS_PB_H_
namespace cartographer {
namespace common {
namespace proto {
class CeresSolverOptions {
 public:
  bool use_nonmonotonic_steps() const { return use_nonmonotonic_steps_; }
  int max_num_iterations() const { return max_num_iterations_; }
  int num_threads() const { return num_threads_; }
  void set_use_nonmonotonic_steps(bool v) { use_nonmonotonic_steps_ = v; }
  void set_max_num_iterations(int v) { max_num_iterations_ = v; }
  void set_num_threads(int v) { num_threads_ = v; }
 private:
  bool use_nonmonotonic_steps_ = false;
  int max_num_iterations_ = 0, num_threads_ = 0;
};
}  // namespace proto
}  // namespace common
}  // namespace cartographer
#endif  // ORACLE_REF_SHIMS_CERES_SOLVER_OPTIONS_PB_H_
