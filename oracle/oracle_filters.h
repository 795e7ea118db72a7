// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_common.h).
// Upstream point preparation restated:
//   cartographer/sensor/internal/voxel_filter.cc:30-36 (FilterByMaxRange), :38-75
//   (AdaptivelyVoxelFiltered), :79-86 (GetVoxelCellIndex), :88-115 (the randomised reservoir
//   filter), :193-198 (AdaptiveVoxelFilter)
//   cartographer/mapping/internal/3d/scan_matching/rotational_scan_matcher.cc:30-120,164-177
//   (ComputeHistogram with SortSlice / AddPointCloudSliceToHistogram / AddValueToHistogram)
// Pinned on the reference's own voxel_filter.cc and rotational_scan_matcher.cc compiled in place
// (oracle/_ref): identical point sets / histograms on random clouds (tests/test_filters.py).
// The voxel filter keeps a RANDOM point per voxel: std::minstd_rand0 default-seeded per call and
// std::uniform_int_distribution -- whose algorithm is the standard library's, so "identical"
// means identical to a reference built with this libstdc++ (GCC 11), like std::sort's tie order
// in the branch and bound.
#ifndef ORACLE_FILTERS_H_
#define ORACLE_FILTERS_H_

#include <cstdint>
#include <vector>

#include "oracle_2d.h"

namespace oracle {

// points_used flags of RandomizedVoxelFilterIndices.
std::vector<uint8_t> VoxelFilterFlags(const PointCloud& cloud, float resolution);
PointCloud VoxelFilter(const PointCloud& cloud, float resolution);
PointCloud AdaptiveVoxelFilter(const PointCloud& cloud, float max_length, float min_num_points,
                               float max_range);
std::vector<float> ComputeHistogram(const PointCloud& cloud, int histogram_size);

}  // namespace oracle

#endif  // ORACLE_FILTERS_H_
