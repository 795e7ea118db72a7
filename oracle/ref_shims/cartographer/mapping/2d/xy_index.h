// Stand-in for mapping/2d/xy_index.h: CellLimits only (the real header needs protobuf).
#ifndef ORACLE_REF_SHIMS_XY_INDEX_H_
#define ORACLE_REF_SHIMS_XY_INDEX_H_
namespace cartographer {
namespace mapping {
struct CellLimits {
  CellLimits() = default;
  CellLimits(int init_num_x_cells, int init_num_y_cells)
      : num_x_cells(init_num_x_cells), num_y_cells(init_num_y_cells) {}
  int num_x_cells = 0;
  int num_y_cells = 0;
};
}  // namespace mapping
}  // namespace cartographer
#endif  // ORACLE_REF_SHIMS_XY_INDEX_H_
