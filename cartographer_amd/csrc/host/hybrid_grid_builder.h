// Host-side HybridGrid producer for fixtures, benchmarks and callers that do
// not already own a cartographer grid.  Sparse voxel map with the write-side
// arithmetic of
//   cartographer/mapping/3d/hybrid_grid.h:414-460   (SetProbability / ApplyLookupTable / FinishUpdate)
//   cartographer/mapping/3d/range_data_inserter_3d.cc:27-114   (hits, then the last
//       `num_free_space_voxels` voxels of every ray as misses)
//   cartographer/mapping/probability_values.cc:76-89  (ComputeLookupTableToApplyOdds)
// and the DynamicGrid growth rule (hybrid_grid.h:259,381-398) that defines
// grid_size().  The matchers consume its flattened voxel list, exactly what
// HybridGrid::Iterator (hybrid_grid.h:304-372) yields.
#ifndef CARTOGRAPHER_AMD_HOST_HYBRID_GRID_BUILDER_H_
#define CARTOGRAPHER_AMD_HOST_HYBRID_GRID_BUILDER_H_

#include <cstdint>
#include <unordered_map>
#include <vector>

namespace cartographer_amd {
namespace host {

struct VoxelRecord { int32_t x, y, z; uint16_t value; uint16_t pad; };

std::vector<uint16_t> ComputeLookupTableToApplyOdds(float odds);   // probability values
uint16_t ProbabilityToValue(float probability);
float ValueToProbability(uint16_t value);

class HybridGridBuilder {
 public:
  explicit HybridGridBuilder(float resolution) : resolution_(resolution) {}
  float resolution() const { return resolution_; }
  int grid_size() const { return grid_size_; }
  void GetCellIndex(const float p[3], int out[3]) const;   // lround(p / resolution), f32
  void SetProbability(int x, int y, int z, float probability);
  float GetProbability(int x, int y, int z) const;
  // One range-data insertion; points in the map frame, xyz stride 3.
  void Insert(const float origin[3], const float* returns_xyz, int num_returns,
              const std::vector<uint16_t>& hit_table, const std::vector<uint16_t>& miss_table,
              int num_free_space_voxels);
  std::vector<VoxelRecord> Voxels() const;   // non-zero cells, sorted (z, y, x)

 private:
  static uint64_t Key(int x, int y, int z) {
    return (static_cast<uint64_t>(static_cast<uint32_t>(x + (1 << 20))) << 42) |
           (static_cast<uint64_t>(static_cast<uint32_t>(y + (1 << 20))) << 21) |
           static_cast<uint64_t>(static_cast<uint32_t>(z + (1 << 20)));
  }
  uint16_t* Mutable(int x, int y, int z);
  bool ApplyLookupTable(int x, int y, int z, const std::vector<uint16_t>& table);
  float resolution_;
  int grid_size_ = 128;   // DynamicGrid starts at 2 x 64 cells per axis
  std::unordered_map<uint64_t, uint16_t> cells_;
  std::vector<uint64_t> update_keys_;
};

}  // namespace host
}  // namespace cartographer_amd

#endif  // CARTOGRAPHER_AMD_HOST_HYBRID_GRID_BUILDER_H_
