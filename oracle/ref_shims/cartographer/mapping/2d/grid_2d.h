// Stand-in for mapping/2d/grid_2d.h (the real one needs protobuf): the read interface the scan
// matchers use, over caller-owned uint16 cells.  Values are converted with the REFERENCE'S OWN
// ValueConversionTables (compiled by `make ref`), asked for the same table Grid2D's constructor
// asks for (grid_2d.cc:60-73); GetCorrespondenceCost follows grid_2d.h:53-57.
#ifndef ORACLE_REF_SHIMS_GRID_2D_H_
#define ORACLE_REF_SHIMS_GRID_2D_H_
#include <vector>
#include "cartographer/mapping/2d/map_limits.h"
#include "cartographer/mapping/probability_values.h"
#include "cartographer/mapping/value_conversion_tables.h"
namespace cartographer {
namespace mapping {
enum class GridType { PROBABILITY_GRID, TSDF };
class Grid2D {
 public:
  Grid2D(const MapLimits& limits, const uint16* cells, float min_correspondence_cost,
         float max_correspondence_cost, ValueConversionTables* conversion_tables)
      : limits_(limits), cells_(cells), min_correspondence_cost_(min_correspondence_cost),
        max_correspondence_cost_(max_correspondence_cost),
        value_to_correspondence_cost_table_(conversion_tables->GetConversionTable(
            max_correspondence_cost, min_correspondence_cost, max_correspondence_cost)) {}
  virtual ~Grid2D() {}
  const MapLimits& limits() const { return limits_; }
  float GetCorrespondenceCost(const Eigen::Array2i& cell_index) const {
    if (!limits().Contains(cell_index)) return max_correspondence_cost_;
    return (*value_to_correspondence_cost_table_)[cells_[ToFlatIndex(cell_index)]];
  }
  virtual GridType GetGridType() const = 0;
  float GetMinCorrespondenceCost() const { return min_correspondence_cost_; }
  float GetMaxCorrespondenceCost() const { return max_correspondence_cost_; }
 protected:
  int ToFlatIndex(const Eigen::Array2i& cell_index) const {
    return limits_.cell_limits().num_x_cells * cell_index.y() + cell_index.x();
  }
  const uint16* cells() const { return cells_; }
 private:
  MapLimits limits_;
  const uint16* cells_;
  float min_correspondence_cost_, max_correspondence_cost_;
  const std::vector<float>* value_to_correspondence_cost_table_;
};
}  // namespace mapping
}  // namespace cartographer
#endif  // ORACLE_REF_SHIMS_GRID_2D_H_
