// Host-side lookup table of the 3D range-data inserter: what
// ComputeLookupTableToApplyOdds(Odds(probability)) returns
// (cartographer/mapping/probability_values.cc:76-89 with probability_values.h:32-61 and the
// kValueToProbability expression of probability_values.cc:33-41).  Plain C++ (no HIP), so that a
// CPU test can compile it on its own and compare it with the reference's table
// (tests/test_device_formulas.py).
#ifndef CMX_ODDS_TABLE_H_
#define CMX_ODDS_TABLE_H_

#include <cmath>
#include <cstdint>

namespace cmx {

constexpr uint16_t kUpdateMarker = 1u << 15;

// out[v] = value of a voxel holding v after the update, with the update marker set; v = 0 is an
// unknown voxel, which simply takes the update's own probability.
inline void ProbabilityOddsTable(float probability, uint16_t* out /*[32768]*/) {
  const float min_p = 0.1f, max_p = 1.f - min_p;
  const auto to_value = [&](float p) {                    // ProbabilityToValue
    const float clamped = p > max_p ? max_p : (p < min_p ? min_p : p);
    return static_cast<uint16_t>(std::lround((clamped - min_p) * (32766.f / (max_p - min_p))) + 1);
  };
  const auto from_odds = [](float o) { return o / (o + 1.f); };
  const float odds = probability / (1.f - probability);
  const float scale = (max_p - min_p) / (32768 - 2.f);    // kValueToProbability's expression
  out[0] = static_cast<uint16_t>(to_value(from_odds(odds)) + kUpdateMarker);
  for (int cell = 1; cell != 32768; ++cell) {
    const float p = cell * scale + (min_p - scale);
    out[cell] = static_cast<uint16_t>(to_value(from_odds(odds * (p / (1.f - p)))) + kUpdateMarker);
  }
}

}  // namespace cmx

#endif  // CMX_ODDS_TABLE_H_
