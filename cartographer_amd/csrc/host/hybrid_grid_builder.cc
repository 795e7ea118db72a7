#include "hybrid_grid_builder.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace cartographer_amd {
namespace host {
namespace {
constexpr float kMinProbability = 0.1f;
constexpr float kMaxProbability = 1.f - kMinProbability;
constexpr uint16_t kUpdateMarker = 1u << 15;
float Clamp(float v, float lo, float hi) { return v > hi ? hi : (v < lo ? lo : v); }
float Odds3(float p) { return p / (1.f - p); }
float ProbabilityFromOdds(float odds) { return odds / (odds + 1.f); }
const std::vector<float>& ProbabilityTable() {
  static const std::vector<float> table = [] {
    std::vector<float> t(65536);
    const float scale = (kMaxProbability - kMinProbability) / (32768 - 2.f);
    for (int i = 0; i != 65536; ++i) {
      const int v = i & 32767;
      t[i] = v == 0 ? kMinProbability : v * scale + (kMinProbability - scale);
    }
    return t;
  }();
  return table;
}
}  // namespace

uint16_t ProbabilityToValue(float probability) {
  const float lo = kMinProbability, hi = kMaxProbability;
  return static_cast<uint16_t>(
      std::lround((Clamp(probability, lo, hi) - lo) * (32766.f / (hi - lo))) + 1);
}
float ValueToProbability(uint16_t value) { return ProbabilityTable()[value]; }

std::vector<uint16_t> ComputeLookupTableToApplyOdds(float odds) {
  std::vector<uint16_t> table;
  table.reserve(32768);
  table.push_back(ProbabilityToValue(ProbabilityFromOdds(odds)) + kUpdateMarker);
  for (int cell = 1; cell != 32768; ++cell)
    table.push_back(ProbabilityToValue(ProbabilityFromOdds(odds * Odds3(ProbabilityTable()[cell]))) +
                    kUpdateMarker);
  return table;
}

void HybridGridBuilder::GetCellIndex(const float p[3], int out[3]) const {
  for (int k = 0; k != 3; ++k) out[k] = static_cast<int>(std::lround(p[k] / resolution_));
}

uint16_t* HybridGridBuilder::Mutable(int x, int y, int z) {
  // DynamicGrid::mutable_value grows until the shifted index is in range.
  auto fits = [&](int v) { return v >= -(grid_size_ / 2) && v < grid_size_ / 2; };
  while (!(fits(x) && fits(y) && fits(z))) grid_size_ *= 2;
  return &cells_[Key(x, y, z)];
}

void HybridGridBuilder::SetProbability(int x, int y, int z, float probability) {
  *Mutable(x, y, z) = ProbabilityToValue(probability);
}

float HybridGridBuilder::GetProbability(int x, int y, int z) const {
  auto it = cells_.find(Key(x, y, z));
  return ValueToProbability(it == cells_.end() ? 0 : it->second);
}

bool HybridGridBuilder::ApplyLookupTable(int x, int y, int z, const std::vector<uint16_t>& table) {
  uint16_t* cell = Mutable(x, y, z);
  if (*cell >= kUpdateMarker) return false;
  update_keys_.push_back(Key(x, y, z));
  *cell = table[*cell];
  return true;
}

void HybridGridBuilder::Insert(const float origin[3], const float* returns_xyz, int num_returns,
                               const std::vector<uint16_t>& hit_table,
                               const std::vector<uint16_t>& miss_table,
                               int num_free_space_voxels) {
  int hit[3];
  for (int i = 0; i != num_returns; ++i) {
    GetCellIndex(returns_xyz + 3 * i, hit);
    ApplyLookupTable(hit[0], hit[1], hit[2], hit_table);
  }
  int o[3];
  GetCellIndex(origin, o);
  for (int i = 0; i != num_returns; ++i) {
    GetCellIndex(returns_xyz + 3 * i, hit);
    const int d[3] = {hit[0] - o[0], hit[1] - o[1], hit[2] - o[2]};
    const int num_samples = std::max(std::abs(d[0]), std::max(std::abs(d[1]), std::abs(d[2])));
    for (int position = std::max(0, num_samples - num_free_space_voxels); position < num_samples;
         ++position) {
      // origin_cell + delta * position / num_samples (integer, truncating).
      ApplyLookupTable(o[0] + d[0] * position / num_samples, o[1] + d[1] * position / num_samples,
                       o[2] + d[2] * position / num_samples, miss_table);
    }
  }
  for (uint64_t key : update_keys_) cells_[key] -= kUpdateMarker;
  update_keys_.clear();
}

std::vector<VoxelRecord> HybridGridBuilder::Voxels() const {
  std::vector<VoxelRecord> out;
  out.reserve(cells_.size());
  for (const auto& kv : cells_) {
    if (kv.second == 0) continue;
    VoxelRecord r;
    r.x = static_cast<int>((kv.first >> 42) & 0x1fffff) - (1 << 20);
    r.y = static_cast<int>((kv.first >> 21) & 0x1fffff) - (1 << 20);
    r.z = static_cast<int>(kv.first & 0x1fffff) - (1 << 20);
    r.value = kv.second;
    r.pad = 0;
    out.push_back(r);
  }
  std::sort(out.begin(), out.end(), [](const VoxelRecord& a, const VoxelRecord& b) {
    if (a.z != b.z) return a.z < b.z;
    if (a.y != b.y) return a.y < b.y;
    return a.x < b.x;
  });
  return out;
}

}  // namespace host
}  // namespace cartographer_amd
