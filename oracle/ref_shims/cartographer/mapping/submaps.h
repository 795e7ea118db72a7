// Stand-in for mapping/submaps.h: probability_grid.cc includes it for the log-odds byte its
// DrawToSubmapTexture writes (compiled, never called here).
#ifndef ORACLE_REF_SHIMS_SUBMAPS_H_
#define ORACLE_REF_SHIMS_SUBMAPS_H_
#include <cmath>
#include "cartographer/common/math.h"
#include "cartographer/common/port.h"
#include "cartographer/mapping/probability_values.h"
#include "glog/logging.h"
namespace cartographer { namespace mapping {
// [kMinProbability, kMaxProbability] in log odds, mapped linearly onto [1, 255].
inline uint8 ProbabilityToLogOddsInteger(const float probability) {
  const auto log_odds = [](float p) { return std::log(p / (1.f - p)); };
  const float lo = log_odds(kMinProbability), hi = log_odds(kMaxProbability);
  const int byte = 1 + common::RoundToInt((log_odds(probability) - lo) * 254.f / (hi - lo));
  CHECK(byte >= 1 && byte <= 255);
  return static_cast<uint8>(byte);
}
} }
#endif  // ORACLE_REF_SHIMS_SUBMAPS_H_
