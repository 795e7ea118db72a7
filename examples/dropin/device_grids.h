// How a submap whose grid lives in HBM (cmx_grid2d / cmx_grid3d, include/cartographer_mi355x.h)
// tells the matcher adapters so: the matchers' interfaces take `const Grid2D&` / `const
// HybridGrid&` and stay as they are.
//   * Grid2D is polymorphic: a grid class that also derives from DeviceGrid2DView is found by
//     dynamic_cast (resident/.../submap_2d.h's DeviceGrid2D).
//   * HybridGrid has no virtual function: the submap registers the address of the (empty) host
//     object it hands out with the device grid behind it.
// Without either, the adapters upload the host grid per call as before.
#ifndef DROPIN_DEVICE_GRIDS_H_
#define DROPIN_DEVICE_GRIDS_H_
#include <map>
#include <mutex>

#include "cartographer_mi355x.h"

namespace dropin {

class DeviceGrid2DView {
 public:
  virtual const cmx_grid2d* device_grid() const = 0;
 protected:
  ~DeviceGrid2DView() = default;
};

namespace internal {
struct Registry3D {
  std::mutex mutex;
  std::map<const void*, const cmx_grid3d*> grids;
};
inline Registry3D& TheRegistry3D() {
  static Registry3D registry;
  return registry;
}
}  // namespace internal

inline void RegisterDeviceGrid(const void* hybrid_grid, const cmx_grid3d* grid) {
  std::lock_guard<std::mutex> lock(internal::TheRegistry3D().mutex);
  internal::TheRegistry3D().grids[hybrid_grid] = grid;
}
inline void UnregisterDeviceGrid(const void* hybrid_grid) {
  std::lock_guard<std::mutex> lock(internal::TheRegistry3D().mutex);
  internal::TheRegistry3D().grids.erase(hybrid_grid);
}
inline const cmx_grid3d* DeviceGridOf(const void* hybrid_grid) {
  std::lock_guard<std::mutex> lock(internal::TheRegistry3D().mutex);
  const auto it = internal::TheRegistry3D().grids.find(hybrid_grid);
  return it == internal::TheRegistry3D().grids.end() ? nullptr : it->second;
}

}  // namespace dropin
#endif  // DROPIN_DEVICE_GRIDS_H_
