// Stand-in for mapping/internal/2d/tsdf_2d.h: GetTSDAndWeight (tsdf_2d.cc:88-98) over two
// caller-owned planes, through the REFERENCE'S OWN TSDValueConverter (compiled by `make ref`).
#ifndef ORACLE_REF_SHIMS_TSDF_2D_H_
#define ORACLE_REF_SHIMS_TSDF_2D_H_
#include <memory>
#include <utility>
#include "cartographer/mapping/2d/grid_2d.h"
#include "cartographer/mapping/internal/2d/tsd_value_converter.h"
namespace cartographer {
namespace mapping {
class TSDF2D : public Grid2D {
 public:
  TSDF2D(const MapLimits& limits, const uint16* tsd_cells, const uint16* weight_cells,
         float truncation_distance, float max_weight, ValueConversionTables* conversion_tables)
      : Grid2D(limits, tsd_cells, -truncation_distance, truncation_distance, conversion_tables),
        value_converter_(new TSDValueConverter(truncation_distance, max_weight,
                                               conversion_tables)),
        weight_cells_(weight_cells) {}
  GridType GetGridType() const override { return GridType::TSDF; }
  std::pair<float, float> GetTSDAndWeight(const Eigen::Array2i& cell_index) const {
    if (limits().Contains(cell_index)) {
      const int flat_index = ToFlatIndex(cell_index);
      return std::make_pair(value_converter_->ValueToTSD(cells()[flat_index]),
                            value_converter_->ValueToWeight(weight_cells_[flat_index]));
    }
    return std::make_pair(value_converter_->getMinTSD(), value_converter_->getMinWeight());
  }
 private:
  std::unique_ptr<TSDValueConverter> value_converter_;
  const uint16* weight_cells_;
};
}  // namespace mapping
}  // namespace cartographer
#endif  // ORACLE_REF_SHIMS_TSDF_2D_H_
