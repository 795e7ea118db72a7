// Stand-in for common/histogram.cc (the reference's formats with absl::StrFormat): the
// constraint builder only logs it when log_matches is set.
#include "cartographer/common/histogram.h"

#include <sstream>

namespace cartographer {
namespace common {
void Histogram::Add(const float value) { values_.push_back(value); }
std::string Histogram::ToString(const int buckets) const {
  std::ostringstream out;
  out << "Count: " << values_.size() << " (" << buckets << " buckets)";
  return out.str();
}
}  // namespace common
}  // namespace cartographer
