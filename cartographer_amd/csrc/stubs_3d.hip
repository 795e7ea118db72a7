// Placeholder 3D entry points (replaced by rt_3d.hip / fast_3d.hip).
#include "cmx_common.h"
extern "C" {
cmx_status cmx_rt3d_match(const cmx_rt_options*, float, const cmx_voxel*, int64_t, const cmx_pose3d*,
                          const float*, int32_t, int32_t, float*, cmx_pose3d*, cmx_match_stats*) {
  cmx::SetLastError("cmx_rt3d_match: not built yet");
  return CMX_UNSUPPORTED;
}
cmx_status cmx_fast3d_create(const cmx_fast3d_options*, float, int32_t, const cmx_voxel*, int64_t, float,
                             const cmx_voxel*, int64_t, const float*, int32_t, int32_t, cmx_fast3d**) {
  cmx::SetLastError("cmx_fast3d_create: not built yet");
  return CMX_UNSUPPORTED;
}
void cmx_fast3d_destroy(cmx_fast3d*) {}
cmx_status cmx_fast3d_match(const cmx_fast3d*, const cmx_pose3d*, const cmx_pose3d*, const cmx_node_data3d*,
                            float, int32_t*, cmx_result3d*, cmx_match_stats*) {
  return CMX_UNSUPPORTED;
}
cmx_status cmx_fast3d_match_full_submap(const cmx_fast3d*, const double*, const double*,
                                        const cmx_node_data3d*, float, int32_t*, cmx_result3d*,
                                        cmx_match_stats*) {
  return CMX_UNSUPPORTED;
}
}
