// Stand-in for the generated protobuf message mapping::proto::CellLimits.
#ifndef ORACLE_REF_SHIMS_CELL_LIMITS_2D_PB_H_
#define ORACLE_REF_SHIMS_CELL_LIMITS_2D_PB_H_
namespace cartographer {
namespace mapping {
namespace proto {
class CellLimits {
 public:
  int num_x_cells() const { return num_x_cells_; }
  int num_y_cells() const { return num_y_cells_; }
  void set_num_x_cells(int v) { num_x_cells_ = v; }
  void set_num_y_cells(int v) { num_y_cells_ = v; }
 private:
  int num_x_cells_ = 0, num_y_cells_ = 0;
};
}  // namespace proto
}  // namespace mapping
}  // namespace cartographer
#endif  // ORACLE_REF_SHIMS_CELL_LIMITS_2D_PB_H_
