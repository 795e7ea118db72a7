// Stand-in for the generated message of mapping/proto/submaps_options_3d.proto.
#ifndef DROPIN_SHIMS_LOCAL_SUBMAPS_OPTIONS_3D_PB_H_
#define DROPIN_SHIMS_LOCAL_SUBMAPS_OPTIONS_3D_PB_H_
#include "cartographer/mapping/proto/range_data_inserter_options_3d.pb.h"
namespace cartographer { namespace mapping { namespace proto {
class SubmapsOptions3D {
 public:
  double high_resolution() const { return high_resolution_; }
  double high_resolution_max_range() const { return high_resolution_max_range_; }
  double low_resolution() const { return low_resolution_; }
  int num_range_data() const { return num_range_data_; }
  void set_high_resolution(double v) { high_resolution_ = v; }
  void set_high_resolution_max_range(double v) { high_resolution_max_range_ = v; }
  void set_low_resolution(double v) { low_resolution_ = v; }
  void set_num_range_data(int v) { num_range_data_ = v; }
  const RangeDataInserterOptions3D& range_data_inserter_options() const { return inserter_; }
  RangeDataInserterOptions3D* mutable_range_data_inserter_options() { return &inserter_; }
 private:
  double high_resolution_ = 0., high_resolution_max_range_ = 0., low_resolution_ = 0.;
  int num_range_data_ = 0;
  RangeDataInserterOptions3D inserter_;
};
} } }
#endif  // DROPIN_SHIMS_LOCAL_SUBMAPS_OPTIONS_3D_PB_H_
