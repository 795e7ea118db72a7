// Stand-in for mapping/submaps.h: probability_grid.cc includes it for one helper its
// DrawToSubmapTexture uses (submaps.h:36-55; compiled, never called here).
#ifndef ORACLE_REF_SHIMS_SUBMAPS_H_
#define ORACLE_REF_SHIMS_SUBMAPS_H_
#include <cmath>
#include "cartographer/common/math.h"
#include "cartographer/common/port.h"
#include "cartographer/mapping/probability_values.h"
#include "glog/logging.h"
namespace cartographer {
namespace mapping {
inline float Logit(float probability) { return std::log(probability / (1.f - probability)); }
inline uint8 ProbabilityToLogOddsInteger(const float probability) {
  const float max_log_odds = Logit(kMaxProbability), min_log_odds = Logit(kMinProbability);
  const int value = common::RoundToInt((Logit(probability) - min_log_odds) * 254.f /
                                       (max_log_odds - min_log_odds)) + 1;
  CHECK_LE(1, value);
  CHECK_GE(255, value);
  return value;
}
}  // namespace mapping
}  // namespace cartographer
#endif  // ORACLE_REF_SHIMS_SUBMAPS_H_
