// Stand-in for mapping/2d/probability_grid.h: GetProbability (probability_grid.cc:78-82) with
// the reference's own inline conversions.
#ifndef ORACLE_REF_SHIMS_PROBABILITY_GRID_H_
#define ORACLE_REF_SHIMS_PROBABILITY_GRID_H_
#include "cartographer/mapping/2d/grid_2d.h"
namespace cartographer {
namespace mapping {
class ProbabilityGrid : public Grid2D {
 public:
  ProbabilityGrid(const MapLimits& limits, const uint16* cells,
                  ValueConversionTables* conversion_tables)
      : Grid2D(limits, cells, kMinCorrespondenceCost, kMaxCorrespondenceCost, conversion_tables) {}
  GridType GetGridType() const override { return GridType::PROBABILITY_GRID; }
  float GetProbability(const Eigen::Array2i& cell_index) const {
    if (!limits().Contains(cell_index)) return kMinProbability;
    return CorrespondenceCostToProbability(
        ValueToCorrespondenceCost(cells()[ToFlatIndex(cell_index)]));
  }
};
}  // namespace mapping
}  // namespace cartographer
#endif  // ORACLE_REF_SHIMS_PROBABILITY_GRID_H_
