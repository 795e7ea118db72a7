// Stand-in for the generated protobuf options message.
#ifndef ORACLE_REF_SHIMS_PG_INSERTER_OPTIONS_2D_PB_H_
#define ORACLE_REF_SHIMS_PG_INSERTER_OPTIONS_2D_PB_H_
namespace cartographer {
namespace mapping {
namespace proto {
class ProbabilityGridRangeDataInserterOptions2D {
 public:
  double hit_probability() const { return hit_probability_; }
  double miss_probability() const { return miss_probability_; }
  bool insert_free_space() const { return insert_free_space_; }
  void set_hit_probability(double v) { hit_probability_ = v; }
  void set_miss_probability(double v) { miss_probability_ = v; }
  void set_insert_free_space(bool v) { insert_free_space_ = v; }
 private:
  double hit_probability_ = 0., miss_probability_ = 0.;
  bool insert_free_space_ = false;
};
}  // namespace proto
}  // namespace mapping
}  // namespace cartographer
#endif  // ORACLE_REF_SHIMS_PG_INSERTER_OPTIONS_2D_PB_H_
