// Device-resident HybridGrid with range-data insertion on gfx950: the step
// on the near side of the real-time 3D matcher (LocalTrajectoryBuilder3D inserts every scan into
// the active submaps' high- and low-resolution grids).
//
// Reference: mapping/3d/range_data_inserter_3d.cc:27-114 (hits, then the last
// `num_free_space_voxels` voxels of every ray as misses, FinishUpdate), mapping/3d/hybrid_grid.h
// :428-433 (GetCellIndex), :471-487 (ApplyLookupTable / FinishUpdate), :259,381-398
// (DynamicGrid::grid_size / Grow), mapping/probability_values.cc:76-89
// (ComputeLookupTableToApplyOdds).
//
// Layout.  The reference's sparse tree (2^bits meta cells of 8^3 x 8^3 voxels) becomes a dense
// uint16 brick over the bounding box of every voxel written so far, x fastest -- the order
// HybridGrid::Iterator walks within a block and the order the matchers' voxel lists use.  The
// brick is re-allocated when a scan reaches outside it; grid_size() follows the reference's
// doubling rule separately (it only feeds the loop-closure matcher's full-submap window).
//
// Parallel form.  As in 2D, a cell is updated at most once per Insert (update marker): first
// every hit, then every miss sample.  All hits apply one table to the pre-insert value and so do
// all misses, so "first writer wins" is order-independent inside a phase: one kernel per phase
// (a thread per return / per (return, sample)), one that clears the markers.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <vector>

#include "cmx_common.h"
#include "cmx_device.h"
#include "cmx_odds_table.h"
#include "scan_matching_3d.h"

struct cmx_grid3d {
  int device = 0;
  float resolution = 0.f;
  int grid_size = 128;                       // DynamicGrid starts at 2 x 64 voxels per axis
  int lo[3] = {0, 0, 0}, dims[3] = {0, 0, 0};   // brick bounds; empty while dims[0] == 0
  uint16_t* cells = nullptr;                 // device, dims[0] * dims[1] * dims[2]
  std::map<uint32_t, uint16_t*> tables;      // odds tables by float bits, device
};

// IntensityHybridGrid (mapping/3d/hybrid_grid.h:543-571): AverageIntensityData {sum, count} per
// voxel, here two dense bricks over the bounding box of the voxels written so far (x fastest),
// plus -- derived on demand, per version -- the f32 brick of sum / count the intensity cost
// function interpolates (GetIntensity: 0 where count == 0 or the cell is absent).
struct cmx_intensity_grid3d {
  int device = 0;
  float resolution = 0.f;
  int lo[3] = {0, 0, 0}, dims[3] = {0, 0, 0};   // brick bounds; empty while dims[0] == 0
  float* sum = nullptr;                      // device
  int* count = nullptr;                      // device
  float* average = nullptr;                  // device, valid for `average_version`
  unsigned long long version = 1, average_version = 0;
};

namespace cmx {
namespace {

const uint16_t* DeviceTable(cmx_grid3d* g, float probability) {
  CMX_REQUIRE(probability > 0.f && probability < 1.f, "probability must be in (0, 1)");
  uint32_t bits;
  std::memcpy(&bits, &probability, sizeof(bits));
  auto it = g->tables.find(bits);
  if (it != g->tables.end()) return it->second;
  std::vector<uint16_t> host(32768);
  ProbabilityOddsTable(probability, host.data());
  uint16_t* table = nullptr;
  CMX_HIP(hipMalloc(reinterpret_cast<void**>(&table), 32768 * sizeof(uint16_t)));
  CMX_HIP(hipMemcpy(table, host.data(), 32768 * sizeof(uint16_t), hipMemcpyHostToDevice));
  g->tables[bits] = table;
  return table;
}

// ---- device ------------------------------------------------------------------------------------
struct BrickView {
  uint16_t* cells;
  int lo_x, lo_y, lo_z, nx, ny, nz;
};

// HybridGridBase::GetCellIndex (hybrid_grid.h:428-433): f32 divide, lround.
__device__ __forceinline__ int3 CellOf(const float* __restrict__ p, float resolution) {
  return make_int3(LRoundF32(p[0] / resolution), LRoundF32(p[1] / resolution),
                   LRoundF32(p[2] / resolution));
}

// The sample `position` of ray origin -> hit (range_data_inserter_3d.cc:37-52):
// origin_cell + delta * position / num_samples, integer arithmetic truncating towards zero.
__device__ __forceinline__ int3 MissCell(int3 origin, int3 delta, int position, int num_samples) {
  return make_int3(origin.x + delta.x * position / num_samples,
                   origin.y + delta.y * position / num_samples,
                   origin.z + delta.z * position / num_samples);
}

// HybridGrid::ApplyLookupTable (hybrid_grid.h:471-481) for one voxel.
__device__ __forceinline__ void Apply(const BrickView& b, int3 c,
                                      const uint16_t* __restrict__ table, int* error) {
  const int x = c.x - b.lo_x, y = c.y - b.lo_y, z = c.z - b.lo_z;
  if (static_cast<unsigned>(x) >= static_cast<unsigned>(b.nx) ||
      static_cast<unsigned>(y) >= static_cast<unsigned>(b.ny) ||
      static_cast<unsigned>(z) >= static_cast<unsigned>(b.nz)) {
    *error = 1;                                          // the extent pass sized the brick
    return;
  }
  uint16_t* cell = b.cells + (static_cast<size_t>(z) * b.ny + y) * b.nx + x;
  const uint16_t old = *cell;
  if (old >= kUpdateMarker) return;
  *cell = table[old];
}

// Pass 0: the bounding box of every voxel this insertion touches (hits and miss samples), so
// that the host can grow the brick and grid_size() before anything is written.
// box = {min x, y, z, max x, y, z}; error bit 2: a ray longer than 2^15 - 1 voxels (CHECK_LT at
// range_data_inserter_3d.cc:39).
__global__ void __launch_bounds__(256)
Grid3DExtentKernel(const float* __restrict__ origin, const float* __restrict__ returns, int n,
                   float resolution, int num_free_space_voxels, int* __restrict__ box,
                   int* __restrict__ error) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[3] = {-0x7fffffff - 1, -0x7fffffff - 1,
                                                             -0x7fffffff - 1};
  if (i < n) {
    const int3 o = CellOf(origin, resolution);
    const int3 h = CellOf(returns + 3 * static_cast<size_t>(i), resolution);
    const auto extend = [&](int3 c) {
      lo[0] = min(lo[0], c.x); lo[1] = min(lo[1], c.y); lo[2] = min(lo[2], c.z);
      hi[0] = max(hi[0], c.x); hi[1] = max(hi[1], c.y); hi[2] = max(hi[2], c.z);
    };
    extend(h);
    const int3 d = make_int3(h.x - o.x, h.y - o.y, h.z - o.z);
    const int num_samples = max(abs(d.x), max(abs(d.y), abs(d.z)));
    if (num_samples >= (1 << 15)) {
      *error = 2;
    } else {
      // Samples are monotone along the ray, so the first and the last touched sample bound
      // every one in between, component by component.
      const int first = max(0, num_samples - num_free_space_voxels);
      if (first < num_samples) {
        extend(MissCell(o, d, first, num_samples));
        extend(MissCell(o, d, num_samples - 1, num_samples));
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int wlo = WaveMin(lo[k]), whi = WaveMax(hi[k]);
    if ((threadIdx.x & 63) == 0 && wlo <= whi) {
      atomicMin(&box[k], wlo);
      atomicMax(&box[3 + k], whi);
    }
  }
}

__global__ void Grid3DHitKernel(BrickView b, const float* __restrict__ returns, int n,
                                float resolution, const uint16_t* __restrict__ hit_table,
                                int* __restrict__ error) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) Apply(b, CellOf(returns + 3 * static_cast<size_t>(i), resolution), hit_table, error);
}

// One thread per (return, k): sample position num_samples - num_free_space_voxels + k.
__global__ void Grid3DMissKernel(BrickView b, const float* __restrict__ origin,
                                 const float* __restrict__ returns, int n, float resolution,
                                 int num_free_space_voxels,
                                 const uint16_t* __restrict__ miss_table,
                                 int* __restrict__ error) {
  const long long t = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= static_cast<long long>(n) * num_free_space_voxels) return;
  const int i = static_cast<int>(t / num_free_space_voxels);
  const int k = static_cast<int>(t - static_cast<long long>(i) * num_free_space_voxels);
  const int3 o = CellOf(origin, resolution);
  const int3 h = CellOf(returns + 3 * static_cast<size_t>(i), resolution);
  const int3 d = make_int3(h.x - o.x, h.y - o.y, h.z - o.z);
  const int num_samples = max(abs(d.x), max(abs(d.y), abs(d.z)));
  const int position = num_samples - num_free_space_voxels + k;
  if (position < 0 || position >= num_samples) return;
  Apply(b, MissCell(o, d, position, num_samples), miss_table, error);
}

// HybridGrid::FinishUpdate (hybrid_grid.h:463-469) over the whole brick.
__global__ void Grid3DFinishKernel(uint16_t* __restrict__ cells, size_t count) {
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < count && cells[i] >= kUpdateMarker) cells[i] -= kUpdateMarker;
}

// The old brick copied into its place inside the new one (one block row per (y, z)).
__global__ void Grid3DCopyKernel(const uint16_t* __restrict__ old_cells, int onx, int ony,
                                 uint16_t* __restrict__ grown, int nnx, int nny, int off_x,
                                 int off_y, int off_z) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y, z = blockIdx.z;
  if (x < onx)
    grown[(static_cast<size_t>(z + off_z) * nny + (y + off_y)) * nnx + x + off_x] =
        old_cells[(static_cast<size_t>(z) * ony + y) * onx + x];
}

__global__ void Grid3DCountKernel(const uint16_t* __restrict__ cells, size_t count,
                                  unsigned long long* __restrict__ total) {
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int known = i < count && cells[i] != 0;
  const int in_wave = WaveSum(known);
  if ((threadIdx.x & 63) == 0 && in_wave) atomicAdd(total, static_cast<unsigned long long>(in_wave));
}

int FloorTo(int v, int m) { return v >= 0 ? v / m * m : -((-v + m - 1) / m * m); }

// Makes the brick cover [lo, hi] (inclusive), keeping its contents.
void EnsureBrick(cmx_grid3d* g, Workspace& ws, const int lo[3], const int hi[3]) {
  int nlo[3], nhi[3];
  bool change = g->dims[0] == 0;
  for (int k = 0; k < 3; ++k) {
    const int cur_lo = g->lo[k], cur_hi = g->lo[k] + g->dims[k] - 1;
    // Grow in steps of 16 voxels so that a slowly widening scene does not re-allocate per scan.
    nlo[k] = g->dims[0] == 0 ? FloorTo(lo[k], 16) : std::min(cur_lo, FloorTo(lo[k], 16));
    nhi[k] = g->dims[0] == 0 ? FloorTo(hi[k], 16) + 15 : std::max(cur_hi, FloorTo(hi[k], 16) + 15);
    change |= nlo[k] != cur_lo || nhi[k] != cur_hi;
  }
  if (!change) return;
  const int ndims[3] = {nhi[0] - nlo[0] + 1, nhi[1] - nlo[1] + 1, nhi[2] - nlo[2] + 1};
  const long long count = static_cast<long long>(ndims[0]) * ndims[1] * ndims[2];   // each < 2^21
  CMX_REQUIRE(count < (1ll << 30), "dense voxel brick of %d x %d x %d is too large", ndims[0],
              ndims[1], ndims[2]);
  uint16_t* grown = nullptr;
  CMX_HIP(hipMalloc(reinterpret_cast<void**>(&grown), static_cast<size_t>(count) * sizeof(uint16_t)));
  hipError_t err = hipMemsetAsync(grown, 0, static_cast<size_t>(count) * sizeof(uint16_t), ws.stream);
  if (err == hipSuccess && g->dims[0] != 0) {
    Grid3DCopyKernel<<<dim3(DivUp(g->dims[0], 256), g->dims[1], g->dims[2]), 256, 0, ws.stream>>>(
        g->cells, g->dims[0], g->dims[1], grown, ndims[0], ndims[1], g->lo[0] - nlo[0],
        g->lo[1] - nlo[1], g->lo[2] - nlo[2]);
    err = hipGetLastError();
  }
  if (err == hipSuccess) err = hipStreamSynchronize(ws.stream);
  if (err != hipSuccess) {
    (void)hipFree(grown);
    CMX_HIP(err);
  }
  if (g->cells) CMX_HIP(hipFree(g->cells));
  g->cells = grown;
  for (int k = 0; k < 3; ++k) { g->lo[k] = nlo[k]; g->dims[k] = ndims[k]; }
}

// ---- IntensityHybridGrid -----------------------------------------------------------------------
// InsertIntensitiesIntoGrid (range_data_inserter_3d.cc:54-70): returns whose intensity exceeds
// the threshold are skipped (`>`: a NaN intensity is inserted, as in the reference); the others
// add to their voxel count += 1, sum += intensity -- an f32 sum IN POINT ORDER.  Parallel form:
// the included returns are stably sorted by voxel (radix sort keeps equal keys in index order),
// and the first return of every voxel walks its run, adding to the voxel's old sum in that order:
// bit-identical sums.
__device__ __forceinline__ bool IntensityIncluded(float intensity, float threshold) {
  return !(intensity > threshold);
}

__global__ void __launch_bounds__(256)
IntensityExtentKernel(const float* __restrict__ returns, const float* __restrict__ intensities,
                      int n, float resolution, float threshold, int* __restrict__ box) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[3] = {-0x7fffffff - 1, -0x7fffffff - 1,
                                                             -0x7fffffff - 1};
  if (i < n && IntensityIncluded(intensities[i], threshold)) {
    const int3 c = CellOf(returns + 3 * static_cast<size_t>(i), resolution);
    lo[0] = hi[0] = c.x; lo[1] = hi[1] = c.y; lo[2] = hi[2] = c.z;
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int wlo = WaveMin(lo[k]), whi = WaveMax(hi[k]);
    if ((threadIdx.x & 63) == 0 && wlo <= whi) {
      atomicMin(&box[k], wlo);
      atomicMax(&box[3 + k], whi);
    }
  }
}

struct IntensityView {
  float* sum;
  int* count;
  int lo_x, lo_y, lo_z, nx, ny, nz;
};

// keys[i] = linear voxel index of return i, 0xffffffff for a skipped return (sorts last).
__global__ void IntensityKeyKernel(IntensityView b, const float* __restrict__ returns,
                                   const float* __restrict__ intensities, int n, float resolution,
                                   float threshold, unsigned* __restrict__ keys,
                                   int* __restrict__ index, int* __restrict__ error) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned key = 0xffffffffu;
  if (IntensityIncluded(intensities[i], threshold)) {
    const int3 c = CellOf(returns + 3 * static_cast<size_t>(i), resolution);
    const int x = c.x - b.lo_x, y = c.y - b.lo_y, z = c.z - b.lo_z;
    if (static_cast<unsigned>(x) >= static_cast<unsigned>(b.nx) ||
        static_cast<unsigned>(y) >= static_cast<unsigned>(b.ny) ||
        static_cast<unsigned>(z) >= static_cast<unsigned>(b.nz)) {
      *error = 1;                                          // the extent pass sized the brick
    } else {
      key = static_cast<unsigned>((z * b.ny + y) * b.nx + x);
    }
  }
  keys[i] = key;
  index[i] = i;
}

// IntensityHybridGrid::AddIntensity (hybrid_grid.h:552-556) for every return of a voxel's run.
__global__ void IntensityApplyKernel(IntensityView b, const unsigned* __restrict__ keys_sorted,
                                     const int* __restrict__ index_sorted, int n,
                                     const float* __restrict__ intensities) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const unsigned key = keys_sorted[j];
  if (key == 0xffffffffu || (j > 0 && keys_sorted[j - 1] == key)) return;
  float sum = b.sum[key];
  int count = b.count[key];
  // The run's head walks it: the f32 sum must be taken in point order (bit parity with the
  // reference's loop), but its LOADS need not wait for one another -- eight returns are fetched
  // (key, index, then intensity: independent chains) before the eight dependent additions, so a
  // long run (dense near-range returns in one voxel) costs an eighth of the round trips.
  constexpr int kAhead = 8;
  for (int k = j; k < n; k += kAhead) {
    float value[kAhead];
    bool same[kAhead];
#pragma unroll
    for (int u = 0; u < kAhead; ++u) {
      const int at = k + u;
      same[u] = at < n && keys_sorted[min(at, n - 1)] == key;
      value[u] = intensities[index_sorted[same[u] ? at : j]];
    }
    bool ended = false;
#pragma unroll
    for (int u = 0; u < kAhead; ++u) {
      ended = ended || !same[u];
      if (!ended) {
        count += 1;
        sum += value[u];
      }
    }
    if (ended) break;
  }
  b.sum[key] = sum;
  b.count[key] = count;
}

// IntensityHybridGrid::GetIntensity (hybrid_grid.h:558-565) for every voxel of the brick.
__global__ void IntensityAverageKernel(const float* __restrict__ sum, const int* __restrict__ count,
                                       size_t cells, float* __restrict__ average) {
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= cells) return;
  const int c = count[i];
  average[i] = c == 0 ? 0.f : sum[i] / static_cast<float>(c);
}

template <typename T>
__global__ void BrickCopyKernel(const T* __restrict__ old_cells, int onx, int ony,
                                T* __restrict__ grown, int nnx, int nny, int off_x, int off_y,
                                int off_z) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y, z = blockIdx.z;
  if (x < onx)
    grown[(static_cast<size_t>(z + off_z) * nny + (y + off_y)) * nnx + x + off_x] =
        old_cells[(static_cast<size_t>(z) * ony + y) * onx + x];
}

// Makes the two bricks cover [lo, hi] (inclusive), keeping their contents.
void EnsureIntensityBrick(cmx_intensity_grid3d* g, Workspace& ws, const int lo[3], const int hi[3]) {
  int nlo[3], nhi[3];
  bool change = g->dims[0] == 0;
  for (int k = 0; k < 3; ++k) {
    const int cur_lo = g->lo[k], cur_hi = g->lo[k] + g->dims[k] - 1;
    nlo[k] = g->dims[0] == 0 ? FloorTo(lo[k], 16) : std::min(cur_lo, FloorTo(lo[k], 16));
    nhi[k] = g->dims[0] == 0 ? FloorTo(hi[k], 16) + 15 : std::max(cur_hi, FloorTo(hi[k], 16) + 15);
    change |= nlo[k] != cur_lo || nhi[k] != cur_hi;
  }
  if (!change) return;
  const int ndims[3] = {nhi[0] - nlo[0] + 1, nhi[1] - nlo[1] + 1, nhi[2] - nlo[2] + 1};
  const long long cells = static_cast<long long>(ndims[0]) * ndims[1] * ndims[2];
  CMX_REQUIRE(cells < (1ll << 30), "dense intensity brick of %d x %d x %d is too large", ndims[0],
              ndims[1], ndims[2]);
  float* sum = nullptr;
  int* count = nullptr;
  CMX_HIP(hipMalloc(reinterpret_cast<void**>(&sum), static_cast<size_t>(cells) * sizeof(float)));
  hipError_t err = hipMalloc(reinterpret_cast<void**>(&count), static_cast<size_t>(cells) * sizeof(int));
  if (err == hipSuccess) err = hipMemsetAsync(sum, 0, static_cast<size_t>(cells) * sizeof(float), ws.stream);
  if (err == hipSuccess) err = hipMemsetAsync(count, 0, static_cast<size_t>(cells) * sizeof(int), ws.stream);
  if (err == hipSuccess && g->dims[0] != 0) {
    const dim3 grid(DivUp(g->dims[0], 256), g->dims[1], g->dims[2]);
    BrickCopyKernel<float><<<grid, 256, 0, ws.stream>>>(g->sum, g->dims[0], g->dims[1], sum, ndims[0],
                                                        ndims[1], g->lo[0] - nlo[0],
                                                        g->lo[1] - nlo[1], g->lo[2] - nlo[2]);
    BrickCopyKernel<int><<<grid, 256, 0, ws.stream>>>(g->count, g->dims[0], g->dims[1], count, ndims[0],
                                                      ndims[1], g->lo[0] - nlo[0],
                                                      g->lo[1] - nlo[1], g->lo[2] - nlo[2]);
    err = hipGetLastError();
  }
  if (err == hipSuccess) err = hipStreamSynchronize(ws.stream);
  if (err != hipSuccess) {
    (void)hipFree(sum);
    if (count) (void)hipFree(count);
    CMX_HIP(err);
  }
  if (g->sum) (void)hipFree(g->sum);
  if (g->count) (void)hipFree(g->count);
  if (g->average) (void)hipFree(g->average);
  g->sum = sum;
  g->count = count;
  g->average = nullptr;
  g->average_version = 0;
  for (int k = 0; k < 3; ++k) { g->lo[k] = nlo[k]; g->dims[k] = ndims[k]; }
}

// InsertIntensitiesIntoGrid for device arrays of returns / intensities (n >= 1).
void InsertIntensities(cmx_intensity_grid3d* g, Workspace& ws, const float* d_returns,
                       const float* d_intensities, int n, float threshold) {
  int* d_box = ws.dev[11].ReserveAs<int>(8);
  int* h_box = ws.pinned[2].ReserveAs<int>(8);
  const int preset[8] = {0x7fffffff, 0x7fffffff, 0x7fffffff, -0x7fffffff - 1, -0x7fffffff - 1,
                         -0x7fffffff - 1, 0, 0};
  std::memcpy(h_box, preset, sizeof(preset));
  CMX_HIP(hipMemcpyAsync(d_box, h_box, sizeof(preset), hipMemcpyHostToDevice, ws.stream));
  IntensityExtentKernel<<<DivUp(n, 256), 256, 0, ws.stream>>>(d_returns, d_intensities, n,
                                                             g->resolution, threshold, d_box);
  CMX_HIP(hipGetLastError());
  CMX_HIP(hipMemcpyAsync(h_box, d_box, sizeof(preset), hipMemcpyDeviceToHost, ws.stream));
  CMX_HIP(hipStreamSynchronize(ws.stream));
  if (h_box[0] > h_box[3]) return;                       // every return above the threshold
  const int lo[3] = {h_box[0], h_box[1], h_box[2]}, hi[3] = {h_box[3], h_box[4], h_box[5]};
  for (int k = 0; k < 3; ++k)
    CMX_REQUIRE(lo[k] > -(1 << 20) && hi[k] < (1 << 20), "voxel index out of range");
  EnsureIntensityBrick(g, ws, lo, hi);
  const IntensityView view{g->sum, g->count, g->lo[0], g->lo[1], g->lo[2],
                           g->dims[0], g->dims[1], g->dims[2]};
  unsigned* keys = ws.dev[12].ReserveAs<unsigned>(2 * static_cast<size_t>(n));
  int* index = ws.dev[13].ReserveAs<int>(2 * static_cast<size_t>(n));
  int* d_error = d_box + 7;                              // still zero
  IntensityKeyKernel<<<DivUp(n, 256), 256, 0, ws.stream>>>(view, d_returns, d_intensities, n,
                                                          g->resolution, threshold, keys, index,
                                                          d_error);
  StableSortPairs32(ws, 14, keys, keys + n, index, index + n, n);
  IntensityApplyKernel<<<DivUp(n, 256), 256, 0, ws.stream>>>(view, keys + n, index + n, n,
                                                            d_intensities);
  CMX_HIP(hipGetLastError());
  // The brick of averages the intensity cost function interpolates, built HERE, on the insert's
  // stream and under its synchronisation: a match never writes to the grid handle (two matches
  // on one grid from different threads used to race on a lazily built brick).
  const size_t cells = static_cast<size_t>(g->dims[0]) * g->dims[1] * g->dims[2];
  if (g->average == nullptr)
    CMX_HIP(hipMalloc(reinterpret_cast<void**>(&g->average), cells * sizeof(float) + 16));
  IntensityAverageKernel<<<DivUp(cells, 256), 256, 0, ws.stream>>>(g->sum, g->count, cells, g->average);
  CMX_HIP(hipGetLastError());
  CMX_HIP(hipMemcpyAsync(h_box, d_box, sizeof(preset), hipMemcpyDeviceToHost, ws.stream));
  CMX_HIP(hipStreamSynchronize(ws.stream));
  CMX_REQUIRE(h_box[7] == 0, "internal error: a voxel fell outside the intensity brick");
  ++g->version;
  g->average_version = g->version;
}

}  // namespace

// The grid's f32 brick of average intensities for the cost function that interpolates it in place
// (ceres_3d.hip); false while the grid is empty (every cell reads 0).  Read-only: the brick is
// built by the insertion that changed the grid (InsertIntensities), so concurrent matches on one
// grid share it like any other resident grid.
bool IntensityGrid3DBrick(cmx_intensity_grid3d* g, hipStream_t /*stream*/, Brick* brick,
                          float* resolution, int* device) {
  *resolution = g->resolution;
  *device = g->device;
  if (g->dims[0] == 0) return false;
  CMX_REQUIRE(g->average != nullptr && g->average_version == g->version,
              "internal error: intensity grid without its brick of averages");
  brick->cells = g->average;
  brick->lo_x = g->lo[0]; brick->lo_y = g->lo[1]; brick->lo_z = g->lo[2];
  brick->nx = g->dims[0]; brick->ny = g->dims[1]; brick->nz = g->dims[2];
  return true;
}

}  // namespace cmx

using cmx::Guard;

extern "C" cmx_status cmx_grid3d_create(float resolution, int32_t device, cmx_grid3d** out) {
  return Guard([&] {
    CMX_REQUIRE(out != nullptr, "null argument");
    CMX_REQUIRE(resolution > 0.f, "bad resolution");
    cmx::UseDevice(device);
    std::unique_ptr<cmx_grid3d> g(new cmx_grid3d);
    g->device = device;
    g->resolution = resolution;
    *out = g.release();
  });
}

extern "C" void cmx_grid3d_destroy(cmx_grid3d* grid) {
  if (!grid) return;
  (void)hipSetDevice(grid->device);
  if (grid->cells) (void)hipFree(grid->cells);
  for (auto& kv : grid->tables) (void)hipFree(kv.second);
  delete grid;
}

namespace cmx {
namespace {
// RangeDataInserter3D::Insert (range_data_inserter_3d.cc:93-114); `intensity_grid` may be null.
void InsertRangeData(cmx_grid3d* grid, cmx_intensity_grid3d* intensity_grid, const float* origin_xyz,
                     const float* returns_xyz, const float* intensities, int32_t num_returns,
                     float hit_probability, float miss_probability, int32_t num_free_space_voxels,
                     float intensity_threshold) {
  {
    CMX_REQUIRE(grid && origin_xyz, "null argument");
    CMX_REQUIRE(num_returns >= 0 && (num_returns == 0 || returns_xyz), "bad range data");
    CMX_REQUIRE(num_free_space_voxels >= 0, "num_free_space_voxels must not be negative");
    if (intensity_grid) {
      CMX_REQUIRE(intensity_grid->device == grid->device, "the grids live on different devices");
      CMX_REQUIRE(intensity_grid->resolution == grid->resolution,
                  "the intensity grid must have the hybrid grid's resolution");
    }
    WorkspaceLease ws(grid->device);
    const uint16_t* hit_table = DeviceTable(grid, hit_probability);
    const uint16_t* miss_table = DeviceTable(grid, miss_probability);
    if (num_returns == 0) return;                        // nothing is written, FinishUpdate no-op
    const int n = num_returns;
    // (`returns.intensities().size() > 0`, :57: a cloud without intensities inserts none)
    const bool with_intensities = intensity_grid != nullptr && intensities != nullptr;
    // Points (origin first), then the intensities -> device.
    const size_t floats = 3 * static_cast<size_t>(n + 1) + (with_intensities ? n : 0);
    float* h_points = ws->pinned[0].ReserveAs<float>(floats);
    std::memcpy(h_points, origin_xyz, 3 * sizeof(float));
    std::memcpy(h_points + 3, returns_xyz, 3 * sizeof(float) * n);
    if (with_intensities)
      std::memcpy(h_points + 3 * static_cast<size_t>(n + 1), intensities, sizeof(float) * n);
    float* d_points = ws->dev[0].ReserveAs<float>(floats);
    CMX_HIP(hipMemcpyAsync(d_points, h_points, sizeof(float) * floats,
                           hipMemcpyHostToDevice, ws->stream));
    const float* d_origin = d_points;
    const float* d_returns = d_points + 3;
    // Pass 0: extent of the touched voxels.
    int* d_box = ws->dev[1].ReserveAs<int>(8);           // 6 bounds, error, pad
    int* h_box = ws->pinned[1].ReserveAs<int>(8);
    const int preset[8] = {0x7fffffff, 0x7fffffff, 0x7fffffff, -0x7fffffff - 1, -0x7fffffff - 1,
                           -0x7fffffff - 1, 0, 0};
    std::memcpy(h_box, preset, sizeof(preset));
    CMX_HIP(hipMemcpyAsync(d_box, h_box, sizeof(preset), hipMemcpyHostToDevice, ws->stream));
    Grid3DExtentKernel<<<DivUp(n, 256), 256, 0, ws->stream>>>(
        d_origin, d_returns, n, grid->resolution, num_free_space_voxels, d_box, d_box + 6);
    CMX_HIP(hipGetLastError());
    CMX_HIP(hipMemcpyAsync(h_box, d_box, sizeof(preset), hipMemcpyDeviceToHost, ws->stream));
    CMX_HIP(hipStreamSynchronize(ws->stream));
    CMX_REQUIRE(h_box[6] == 0, "a ray spans 2^15 voxels or more");   // CHECK_LT at :39
    const int lo[3] = {h_box[0], h_box[1], h_box[2]}, hi[3] = {h_box[3], h_box[4], h_box[5]};
    for (int k = 0; k < 3; ++k)
      CMX_REQUIRE(lo[k] > -(1 << 20) && hi[k] < (1 << 20), "voxel index out of range");
    // DynamicGrid::mutable_value grows until every written index fits (hybrid_grid.h:282-287).
    const auto fits = [&](int v) { return v >= -(grid->grid_size / 2) && v < grid->grid_size / 2; };
    for (int k = 0; k < 3; ++k)
      while (!(fits(lo[k]) && fits(hi[k]))) grid->grid_size *= 2;
    EnsureBrick(grid, *ws, lo, hi);

    const BrickView view{grid->cells, grid->lo[0], grid->lo[1], grid->lo[2], grid->dims[0],
                         grid->dims[1], grid->dims[2]};
    int* d_error = d_box + 7;                            // still zero
    Grid3DHitKernel<<<DivUp(n, 256), 256, 0, ws->stream>>>(view, d_returns, n, grid->resolution,
                                                          hit_table, d_error);
    if (num_free_space_voxels > 0) {
      const long long samples = static_cast<long long>(n) * num_free_space_voxels;
      CMX_REQUIRE(samples < (1ll << 38), "too many free-space samples");
      Grid3DMissKernel<<<DivUp(samples, 256), 256, 0, ws->stream>>>(
          view, d_origin, d_returns, n, grid->resolution, num_free_space_voxels, miss_table,
          d_error);
    }
    const size_t count = static_cast<size_t>(view.nx) * view.ny * view.nz;
    Grid3DFinishKernel<<<DivUp(count, 256), 256, 0, ws->stream>>>(grid->cells, count);
    CMX_HIP(hipGetLastError());
    CMX_HIP(hipMemcpyAsync(h_box, d_box, sizeof(preset), hipMemcpyDeviceToHost, ws->stream));
    CMX_HIP(hipStreamSynchronize(ws->stream));
    CMX_REQUIRE(h_box[7] == 0, "internal error: a voxel fell outside the brick");
    if (with_intensities)
      InsertIntensities(intensity_grid, *ws, d_returns, d_points + 3 * static_cast<size_t>(n + 1), n,
                        intensity_threshold);
  }
}
}  // namespace
}  // namespace cmx

extern "C" cmx_status cmx_grid3d_insert(cmx_grid3d* grid, const float* origin_xyz,
                                        const float* returns_xyz, int32_t num_returns,
                                        float hit_probability, float miss_probability,
                                        int32_t num_free_space_voxels) {
  return Guard([&] {
    cmx::InsertRangeData(grid, nullptr, origin_xyz, returns_xyz, nullptr, num_returns,
                         hit_probability, miss_probability, num_free_space_voxels, 0.f);
  });
}

extern "C" cmx_status cmx_grid3d_insert_with_intensities(
    cmx_grid3d* grid, cmx_intensity_grid3d* intensity_grid, const float* origin_xyz,
    const float* returns_xyz, const float* intensities, int32_t num_returns, float hit_probability,
    float miss_probability, int32_t num_free_space_voxels, float intensity_threshold) {
  return Guard([&] {
    CMX_REQUIRE(intensity_grid != nullptr, "null argument");
    cmx::InsertRangeData(grid, intensity_grid, origin_xyz, returns_xyz, intensities, num_returns,
                         hit_probability, miss_probability, num_free_space_voxels,
                         intensity_threshold);
  });
}

extern "C" cmx_status cmx_intensity_grid3d_create(float resolution, int32_t device,
                                                  cmx_intensity_grid3d** out) {
  return Guard([&] {
    CMX_REQUIRE(out != nullptr, "null argument");
    CMX_REQUIRE(resolution > 0.f, "bad resolution");
    cmx::UseDevice(device);
    std::unique_ptr<cmx_intensity_grid3d> g(new cmx_intensity_grid3d);
    g->device = device;
    g->resolution = resolution;
    *out = g.release();
  });
}

extern "C" void cmx_intensity_grid3d_destroy(cmx_intensity_grid3d* grid) {
  if (!grid) return;
  (void)hipSetDevice(grid->device);
  if (grid->sum) (void)hipFree(grid->sum);
  if (grid->count) (void)hipFree(grid->count);
  if (grid->average) (void)hipFree(grid->average);
  delete grid;
}

// The voxels with count > 0 in (z, y, x) order -- what HybridGridBase's Iterator yields -- at most
// `capacity` of them; *num_voxels receives the full count.
extern "C" cmx_status cmx_intensity_grid3d_download(const cmx_intensity_grid3d* grid,
                                                    cmx_intensity_voxel* voxels, int64_t capacity,
                                                    int64_t* num_voxels) {
  using namespace cmx;
  return Guard([&] {
    CMX_REQUIRE(grid && num_voxels && (capacity == 0 || voxels), "null argument");
    *num_voxels = 0;
    if (grid->dims[0] == 0) return;
    UseDevice(grid->device);
    const size_t cells = static_cast<size_t>(grid->dims[0]) * grid->dims[1] * grid->dims[2];
    std::vector<float> sum(cells);
    std::vector<int> count(cells);
    CMX_HIP(hipMemcpy(sum.data(), grid->sum, cells * sizeof(float), hipMemcpyDeviceToHost));
    CMX_HIP(hipMemcpy(count.data(), grid->count, cells * sizeof(int), hipMemcpyDeviceToHost));
    int64_t k = 0;
    size_t i = 0;
    for (int z = 0; z < grid->dims[2]; ++z)
      for (int y = 0; y < grid->dims[1]; ++y)
        for (int x = 0; x < grid->dims[0]; ++x, ++i) {
          if (count[i] == 0) continue;
          if (k < capacity)
            voxels[k] = cmx_intensity_voxel{x + grid->lo[0], y + grid->lo[1], z + grid->lo[2],
                                            count[i], sum[i]};
          ++k;
        }
    *num_voxels = k;
  });
}

namespace cmx {
// The grid's dense uint16 brick for the matchers that read it in place (ceres_3d.hip); false
// while the grid is empty (nothing was inserted yet: every cell reads 0).
bool Grid3DBrick(const cmx_grid3d* g, Brick* brick, float* resolution, int* device) {
  *resolution = g->resolution;
  *device = g->device;
  if (g->dims[0] == 0) return false;
  brick->cells = g->cells;
  brick->lo_x = g->lo[0]; brick->lo_y = g->lo[1]; brick->lo_z = g->lo[2];
  brick->nx = g->dims[0]; brick->ny = g->dims[1]; brick->nz = g->dims[2];
  return true;
}
}  // namespace cmx

extern "C" cmx_status cmx_grid3d_info(const cmx_grid3d* grid, float* resolution,
                                      int32_t* grid_size, int64_t* num_voxels) {
  using namespace cmx;
  return Guard([&] {
    CMX_REQUIRE(grid != nullptr, "null argument");
    if (resolution) *resolution = grid->resolution;
    if (grid_size) *grid_size = grid->grid_size;
    if (num_voxels) {
      *num_voxels = 0;
      if (grid->dims[0] == 0) return;
      WorkspaceLease ws(grid->device);
      unsigned long long* d_total = ws->dev[1].ReserveAs<unsigned long long>(1);
      CMX_HIP(hipMemsetAsync(d_total, 0, sizeof(unsigned long long), ws->stream));
      const size_t count = static_cast<size_t>(grid->dims[0]) * grid->dims[1] * grid->dims[2];
      Grid3DCountKernel<<<DivUp(count, 256), 256, 0, ws->stream>>>(grid->cells, count, d_total);
      CMX_HIP(hipGetLastError());
      unsigned long long total = 0;
      CMX_HIP(hipMemcpyAsync(&total, d_total, sizeof(total), hipMemcpyDeviceToHost, ws->stream));
      CMX_HIP(hipStreamSynchronize(ws->stream));
      *num_voxels = static_cast<int64_t>(total);
    }
  });
}

// The non-zero voxels in (z, y, x) order -- what HybridGrid::Iterator yields, sorted the way the
// matchers' voxel lists are -- at most `capacity` of them; *num_voxels receives the full count.
extern "C" cmx_status cmx_grid3d_download(const cmx_grid3d* grid, cmx_voxel* voxels,
                                          int64_t capacity, int64_t* num_voxels) {
  using namespace cmx;
  return Guard([&] {
    CMX_REQUIRE(grid && num_voxels && (capacity == 0 || voxels), "null argument");
    *num_voxels = 0;
    if (grid->dims[0] == 0) return;
    UseDevice(grid->device);
    const size_t count = static_cast<size_t>(grid->dims[0]) * grid->dims[1] * grid->dims[2];
    std::vector<uint16_t> host(count);
    CMX_HIP(hipMemcpy(host.data(), grid->cells, count * sizeof(uint16_t), hipMemcpyDeviceToHost));
    int64_t k = 0;
    size_t i = 0;
    for (int z = 0; z < grid->dims[2]; ++z)
      for (int y = 0; y < grid->dims[1]; ++y)
        for (int x = 0; x < grid->dims[0]; ++x, ++i) {
          if (host[i] == 0) continue;
          if (k < capacity)
            voxels[k] = cmx_voxel{x + grid->lo[0], y + grid->lo[1], z + grid->lo[2], host[i], 0};
          ++k;
        }
    *num_voxels = k;
  });
}
