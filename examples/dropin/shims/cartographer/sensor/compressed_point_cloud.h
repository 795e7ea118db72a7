// Stand-in for sensor/compressed_point_cloud.h: constraint_builder_3d.h includes it without
// using it (the real header needs the generated sensor.pb.h).
#ifndef DROPIN_SHIMS_COMPRESSED_POINT_CLOUD_H_
#define DROPIN_SHIMS_COMPRESSED_POINT_CLOUD_H_
#include "cartographer/sensor/point_cloud.h"
#endif  // DROPIN_SHIMS_COMPRESSED_POINT_CLOUD_H_
