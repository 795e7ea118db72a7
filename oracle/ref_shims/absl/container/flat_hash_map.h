// Stand-in for absl::flat_hash_map: sensor/internal/voxel_filter.cc only inserts through
// operator[] and iterates to set order-independent flags, so any hash map serves.
#ifndef ORACLE_REF_SHIMS_ABSL_FLAT_HASH_MAP_H_
#define ORACLE_REF_SHIMS_ABSL_FLAT_HASH_MAP_H_
#include <unordered_map>
namespace absl {
template <class K, class V>
using flat_hash_map = std::unordered_map<K, V>;
}  // namespace absl
#endif  // ORACLE_REF_SHIMS_ABSL_FLAT_HASH_MAP_H_
