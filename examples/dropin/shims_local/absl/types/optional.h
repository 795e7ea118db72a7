// Stand-in for absl/types/optional.h: absl::optional is std::optional under C++17.
#ifndef DROPIN_SHIMS_LOCAL_ABSL_OPTIONAL_H_
#define DROPIN_SHIMS_LOCAL_ABSL_OPTIONAL_H_
#include <optional>
namespace absl {
template <typename T>
using optional = std::optional<T>;
}  // namespace absl
#endif  // DROPIN_SHIMS_LOCAL_ABSL_OPTIONAL_H_
