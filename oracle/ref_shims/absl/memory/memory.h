#ifndef ORACLE_REF_SHIMS_ABSL_MEMORY_H_
#define ORACLE_REF_SHIMS_ABSL_MEMORY_H_
#include <memory>
#include <utility>
namespace absl {
template <typename T, typename... Args>
std::unique_ptr<T> make_unique(Args&&... args) {
  return std::unique_ptr<T>(new T(std::forward<Args>(args)...));
}
}  // namespace absl
#endif  // ORACLE_REF_SHIMS_ABSL_MEMORY_H_
