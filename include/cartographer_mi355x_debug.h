/* Test and tool switches of libcartographer_mi355x -- NOT part of the drop-in boundary.
 *
 * The parity tests run every device path against its partner (the tile path of the real-time
 * 2D matcher against the one-thread-per-candidate kernels, verification modes of the 3D
 * matchers, ...) and the profiling tools override tuning choices.  Those selections go through
 * this call; the product reads no environment variable on a call path.  Names are the fields of
 * cmx::DebugOptions (cartographer_amd/csrc/cmx_common.h); everything is 0 by default.
 * Process-wide and not synchronised: set switches before the calls they should affect. */
#ifndef CARTOGRAPHER_MI355X_DEBUG_H_
#define CARTOGRAPHER_MI355X_DEBUG_H_

#include "cartographer_mi355x.h"

#ifdef __cplusplus
extern "C" {
#endif

/* CMX_INVALID_ARGUMENT for an unknown name. */
cmx_status cmx_debug_set(const char* name, int32_t value);
/* Every switch back to 0. */
void cmx_debug_reset(void);

#ifdef __cplusplus
}
#endif

#endif  /* CARTOGRAPHER_MI355X_DEBUG_H_ */
