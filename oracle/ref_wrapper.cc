// ORACLE — TEST INFRASTRUCTURE ONLY.  C wrapper around five reference translation units that
// `make -C oracle ref` compiles unmodified from /root/reference (see ref_shims/README.md).
#include <cstdint>
#include <cstring>
#include <vector>

#include "cartographer/common/fixed_ratio_sampler.h"
#include "cartographer/mapping/internal/2d/ray_to_pixel_mask.h"
#include "cartographer/mapping/internal/2d/tsd_value_converter.h"
#include "cartographer/mapping/probability_values.h"
#include "cartographer/mapping/value_conversion_tables.h"

extern "C" {

// RayToPixelMask: returns the number of pixels; out gets min(count, capacity) (x, y) pairs.
int ref_ray_to_pixel_mask(int bx, int by, int ex, int ey, int subpixel_scale, int32_t* out_xy,
                          int capacity) {
  const std::vector<Eigen::Array2i> mask = cartographer::mapping::RayToPixelMask(
      Eigen::Array2i(bx, by), Eigen::Array2i(ex, ey), subpixel_scale);
  const int n = static_cast<int>(mask.size());
  for (int i = 0; i < n && i < capacity; ++i) {
    out_xy[2 * i] = mask[i].x();
    out_xy[2 * i + 1] = mask[i].y();
  }
  return n;
}

// kValueToProbability / kValueToCorrespondenceCost, 65536 entries each.
void ref_value_tables(float* value_to_probability, float* value_to_correspondence_cost) {
  for (int v = 0; v != 65536; ++v) {
    value_to_probability[v] = cartographer::mapping::ValueToProbability(static_cast<uint16_t>(v));
    value_to_correspondence_cost[v] =
        cartographer::mapping::ValueToCorrespondenceCost(static_cast<uint16_t>(v));
  }
}

int ref_probability_to_value(float p) { return cartographer::mapping::ProbabilityToValue(p); }
int ref_correspondence_cost_to_value(float c) {
  return cartographer::mapping::CorrespondenceCostToValue(c);
}

// ComputeLookupTableToApplyCorrespondenceCostOdds(Odds(probability)) / ...ApplyOdds, 32768 each.
void ref_odds_tables(float probability, uint16_t* correspondence_cost_table,
                     uint16_t* probability_table) {
  const float odds = cartographer::mapping::Odds(probability);
  const std::vector<uint16_t> cc =
      cartographer::mapping::ComputeLookupTableToApplyCorrespondenceCostOdds(odds);
  const std::vector<uint16_t> pr = cartographer::mapping::ComputeLookupTableToApplyOdds(odds);
  std::memcpy(correspondence_cost_table, cc.data(), 32768 * sizeof(uint16_t));
  std::memcpy(probability_table, pr.data(), 32768 * sizeof(uint16_t));
}

// ValueConversionTables::GetConversionTable(unknown_result, lower, upper), 65536 entries.
void ref_conversion_table(float unknown_result, float lower_bound, float upper_bound, float* out) {
  cartographer::mapping::ValueConversionTables tables;
  const std::vector<float>* t = tables.GetConversionTable(unknown_result, lower_bound, upper_bound);
  std::memcpy(out, t->data(), 65536 * sizeof(float));
}

// TSDValueConverter(max_tsd, max_weight): kind 0 = TSDToValue / ValueToTSD, 1 = weights.
int ref_tsd_float_to_value(int kind, float max_tsd, float max_weight, float x) {
  cartographer::mapping::ValueConversionTables tables;
  const cartographer::mapping::TSDValueConverter c(max_tsd, max_weight, &tables);
  return kind == 0 ? c.TSDToValue(x) : c.WeightToValue(x);
}
float ref_tsd_value_to_float(int kind, float max_tsd, float max_weight, int value) {
  cartographer::mapping::ValueConversionTables tables;
  const cartographer::mapping::TSDValueConverter c(max_tsd, max_weight, &tables);
  return kind == 0 ? c.ValueToTSD(static_cast<uint16_t>(value))
                   : c.ValueToWeight(static_cast<uint16_t>(value));
}

// FixedRatioSampler(ratio): the outcome of `count` consecutive Pulse() calls.
void ref_fixed_ratio_sampler(double ratio, int count, uint8_t* out) {
  cartographer::common::FixedRatioSampler sampler(ratio);
  for (int i = 0; i != count; ++i) out[i] = sampler.Pulse() ? 1 : 0;
}

}  // extern "C"
