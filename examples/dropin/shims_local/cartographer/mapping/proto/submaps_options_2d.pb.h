// Extends oracle/ref_shims' stand-in of submaps_options_2d.proto (GridOptions2D and friends) by
// the SubmapsOptions2D message ActiveSubmaps2D is configured with: num_range_data, the grid
// options and -- flattened, this build has the probability-grid inserter only -- the inserter's
// options.
#ifndef DROPIN_SHIMS_LOCAL_SUBMAPS_OPTIONS_2D_PB_H_
#define DROPIN_SHIMS_LOCAL_SUBMAPS_OPTIONS_2D_PB_H_
#include_next "cartographer/mapping/proto/submaps_options_2d.pb.h"
#include "cartographer/mapping/proto/probability_grid_range_data_inserter_options_2d.pb.h"
namespace cartographer { namespace mapping { namespace proto {
class SubmapsOptions2D {
 public:
  int num_range_data() const { return num_range_data_; }
  void set_num_range_data(int v) { num_range_data_ = v; }
  const GridOptions2D& grid_options_2d() const { return grid_options_2d_; }
  GridOptions2D* mutable_grid_options_2d() { return &grid_options_2d_; }
  const ProbabilityGridRangeDataInserterOptions2D&
  probability_grid_range_data_inserter_options_2d() const { return inserter_; }
  ProbabilityGridRangeDataInserterOptions2D*
  mutable_probability_grid_range_data_inserter_options_2d() { return &inserter_; }
 private:
  int num_range_data_ = 0;
  GridOptions2D grid_options_2d_;
  ProbabilityGridRangeDataInserterOptions2D inserter_;
};
} } }
#endif  // DROPIN_SHIMS_LOCAL_SUBMAPS_OPTIONS_2D_PB_H_
