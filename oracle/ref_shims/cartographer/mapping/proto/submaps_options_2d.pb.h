// Stand-in for the generated options messages grid_2d.{h,cc} and range_data_inserter_interface.h
// name (their Create*Options functions are compiled, never called here).
#ifndef ORACLE_REF_SHIMS_SUBMAPS_OPTIONS_2D_PB_H_
#define ORACLE_REF_SHIMS_SUBMAPS_OPTIONS_2D_PB_H_
#include <string>
namespace cartographer {
namespace mapping {
namespace proto {
enum GridOptions2D_GridType {
  GridOptions2D_GridType_INVALID_GRID = 0,
  GridOptions2D_GridType_PROBABILITY_GRID = 1,
  GridOptions2D_GridType_TSDF = 2
};
inline bool GridOptions2D_GridType_Parse(const std::string& name, GridOptions2D_GridType* value) {
  if (name == "PROBABILITY_GRID") { *value = GridOptions2D_GridType_PROBABILITY_GRID; return true; }
  if (name == "TSDF") { *value = GridOptions2D_GridType_TSDF; return true; }
  return false;
}
class GridOptions2D {
 public:
  void set_grid_type(GridOptions2D_GridType v) { grid_type_ = v; }
  void set_resolution(double v) { resolution_ = v; }
  GridOptions2D_GridType grid_type() const { return grid_type_; }
  double resolution() const { return resolution_; }
 private:
  GridOptions2D_GridType grid_type_ = GridOptions2D_GridType_INVALID_GRID;
  double resolution_ = 0.;
};
class RangeDataInserterOptions {};
}  // namespace proto
}  // namespace mapping
}  // namespace cartographer
#endif  // ORACLE_REF_SHIMS_SUBMAPS_OPTIONS_2D_PB_H_
