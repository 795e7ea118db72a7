// Device-resident ProbabilityGrid with range-data insertion on gfx950:
// the step on the near side of the real-time 2D matcher.  LocalTrajectoryBuilder2D
// alternates Match(active grid) and InsertRangeData(active grid) for every scan
// (mapping/internal/2d/local_trajectory_builder_2d.cc:78-80, :288-289); with the grid in HBM
// neither step moves it across PCIe.
//
// Reference: mapping/2d/probability_grid_range_data_inserter_2d.cc:33-96 (CastRays,
// GrowAsNeeded, Insert), mapping/internal/2d/ray_to_pixel_mask.cc:34-156 (RayToPixelMask),
// mapping/2d/probability_grid.cc:58-82 (ApplyLookupTable), mapping/2d/grid_2d.cc:118-164
// (FinishUpdate, GrowLimits), mapping/probability_values.cc:76-105 (odds tables).
//
// Parallel form.  The reference updates each cell at most once per Insert (the update
// marker, probability_grid.cc:64-66): first every hit, then every ray cell.  All hits
// apply the same table to the pre-insert value and so do all misses, so "first writer
// wins" is order-independent inside a phase: one kernel for the hits, one for the rays
// (one wavefront per ray, lanes over pixel columns), one that clears the markers.  A racing
// second writer either still sees the unmarked value and stores the same result, or sees
// the marker and skips: the outcome is bit-identical to the sequential loops.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <vector>

#include "scan_matching_2d.h"

struct cmx_grid2d {
  int device = 0;
  double resolution = 0., max_x = 0., max_y = 0.;
  int nx = 0, ny = 0;
  uint16_t* cells = nullptr;                             // device, nx * ny
  std::map<uint32_t, uint16_t*> tables;                  // odds tables by float bits, device
  unsigned long long version = 1;                        // bumped whenever the cells change
  mutable cmx::Rt2DImageCache rt_image;                  // the real-time matcher's staged image
};

namespace cmx {
namespace {

constexpr int kSubpixelScale = 1000;                     // ..._inserter_2d.cc:33
constexpr uint16_t kUpdateMarker = 1u << 15;

// ---- odds tables (host; mapping/probability_values.{h,cc}) -------------------
float ClampF(float v, float lo, float hi) { return v > hi ? hi : (v < lo ? lo : v); }

struct ValueTables {
  float min_cc, max_cc;
  float cost[32768];
  ValueTables() {
    const float min_p = 0.1f, max_p = 1.f - min_p;
    min_cc = 1.f - max_p;
    max_cc = 1.f - min_p;
    const float scale = (max_cc - min_cc) / (32768 - 2.f);
    cost[0] = max_cc;
    for (int v = 1; v != 32768; ++v) cost[v] = v * scale + (min_cc - scale);
  }
  uint16_t CostToValue(float c) const {                  // CorrespondenceCostToValue
    return static_cast<uint16_t>(
        std::lround((ClampF(c, min_cc, max_cc) - min_cc) * (32766.f / (max_cc - min_cc))) + 1);
  }
};

// ComputeLookupTableToApplyCorrespondenceCostOdds (probability_values.cc:91-105).
void OddsTable(float probability, uint16_t* out /*[32768]*/) {
  static const ValueTables t;
  const float odds = probability / (1.f - probability);
  const auto from_odds = [](float o) { return o / (o + 1.f); };
  out[0] = t.CostToValue(1.f - from_odds(odds)) + kUpdateMarker;
  for (int cell = 1; cell != 32768; ++cell) {
    const float p = 1.f - t.cost[cell];
    out[cell] = t.CostToValue(1.f - from_odds(odds * (p / (1.f - p)))) + kUpdateMarker;
  }
}

// ---- device ----------------------------------------------------------------------
struct GridView {
  uint16_t* cells;
  int nx, ny;
  double fine_resolution, max_x, max_y;                  // superscaled limits (:58-62)
};

// MapLimits::GetCellIndex on the superscaled limits (mapping/2d/map_limits.h:69-76).
__device__ __forceinline__ int2 FineIndex(const GridView& g, float px, float py) {
  return make_int2(LRoundF64((g.max_y - static_cast<double>(py)) / g.fine_resolution - 0.5),
                   LRoundF64((g.max_x - static_cast<double>(px)) / g.fine_resolution - 0.5));
}

// ProbabilityGrid::ApplyLookupTable (probability_grid.cc:58-72) for one cell.
__device__ __forceinline__ void Apply(const GridView& g, int cx, int cy,
                                      const uint16_t* __restrict__ table, int* error) {
  if (static_cast<unsigned>(cx) >= static_cast<unsigned>(g.nx) ||
      static_cast<unsigned>(cy) >= static_cast<unsigned>(g.ny)) {
    *error = 1;                                          // DCHECK(limits().Contains(cell_index))
    return;
  }
  uint16_t* cell = g.cells + static_cast<size_t>(g.nx) * cy + cx;
  const uint16_t old = *cell;
  if (old >= kUpdateMarker) return;
  *cell = table[old];
}

// Hits: ends[i] = superscaled cell of return i; ApplyLookupTable(ends[i] / scale, hit_table)
// (:66-73).  The same kernel computes the superscaled cells of the misses.
__global__ void GridHitKernel(GridView g, const float* __restrict__ points, int num_returns,
                              int num_points, const uint16_t* __restrict__ hit_table,
                              int2* __restrict__ ends, int* __restrict__ error) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= num_points) return;
  const int2 fine = FineIndex(g, points[3 * i], points[3 * i + 1]);
  ends[i] = fine;
  if (i < num_returns) Apply(g, fine.x / kSubpixelScale, fine.y / kSubpixelScale, hit_table, error);
}

__device__ __forceinline__ long long FloorDiv(long long a, long long b) {
  return a >= 0 ? a / b : -((-a + b - 1) / b);
}
__device__ __forceinline__ long long CeilDiv(long long a, long long b) { return -FloorDiv(-a, b); }

// Rays: one wavefront per ray from the origin to ends[i]; the cells RayToPixelMask returns
// (ray_to_pixel_mask.cc:34-156), column by column: the ray enters pixel column `col` at
// height y_in and leaves it at y_out (exact integers in half-sub-pixel units scaled by dx);
// the column contributes the pixels between them, a corner touched exactly adding none.
__global__ void __launch_bounds__(256)
GridMissKernel(GridView g, float origin_x, float origin_y, const int2* __restrict__ ends,
               int num_rays, const uint16_t* __restrict__ miss_table, int* __restrict__ error) {
  const int lane = threadIdx.x & 63;
  const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= num_rays) return;
  int2 begin = FineIndex(g, origin_x, origin_y);
  int2 end = ends[ray];
  if (begin.x > end.x) { const int2 t = begin; begin = end; end = t; }
  const int scale = kSubpixelScale;
  const int col0 = begin.x / scale, col1 = end.x / scale;
  if (col0 == col1) {                                    // stays inside one pixel column
    const int lo = min(begin.y, end.y) / scale, hi = max(begin.y, end.y) / scale;
    for (int y = lo + lane; y <= hi; y += 64) Apply(g, col0, y, miss_table, error);
    return;
  }
  const long long dx = static_cast<long long>(end.x) - begin.x;
  const long long dy = static_cast<long long>(end.y) - begin.y;
  const long long x2_begin = 2ll * begin.x + 1, x2_end = 2ll * end.x + 1;
  const long long y2_begin = 2ll * begin.y + 1;
  const long long pixel = 2ll * scale * dx;
  for (int col = col0 + lane; col <= col1; col += 64) {
    const long long left = max(2ll * scale * col, x2_begin);
    const long long right = min(2ll * scale * (col + 1), x2_end);
    const long long y_in = y2_begin * dx + (left - x2_begin) * dy;
    const long long y_out = y2_begin * dx + (right - x2_begin) * dy;
    long long first, last;
    if (dy > 0) {
      first = FloorDiv(y_in, pixel);
      last = CeilDiv(y_out, pixel) - 1;
    } else {
      last = CeilDiv(y_in, pixel) - 1;
      first = FloorDiv(y_out, pixel);
    }
    for (long long y = first; y <= last; ++y) Apply(g, col, static_cast<int>(y), miss_table, error);
  }
}

// Grid2D::FinishUpdate (grid_2d.cc:118-125) over the whole grid.
__global__ void GridFinishKernel(uint16_t* __restrict__ cells, size_t count) {
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < count && cells[i] >= kUpdateMarker) cells[i] -= kUpdateMarker;
}

// Grid2D::GrowLimits (grid_2d.cc:130-164): the old grid lands in the middle of one twice
// as large.
__global__ void GridGrowKernel(const uint16_t* __restrict__ old_cells, int nx, int ny,
                               uint16_t* __restrict__ grown, int x_offset, int y_offset) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  if (x < nx) grown[static_cast<size_t>(y + y_offset) * (2 * nx) + x + x_offset] =
      old_cells[static_cast<size_t>(y) * nx + x];
}

// Grid2D::ComputeCroppedLimits (grid_2d.cc:104-114): bounding box of the known cells
// (known_cells_box_ is extended by every SetProbability / ApplyLookupTable, and a known cell never
// becomes unknown again, so it is the box of the non-zero cells).  box = {min_x, min_y, max_x,
// max_y}, preset to {INT_MAX, INT_MAX, -1, -1}; one set of atomics per wavefront.
__global__ void __launch_bounds__(256)
GridKnownBoxKernel(const uint16_t* __restrict__ cells, int nx, int ny, int* __restrict__ box) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  const bool known = x < nx && cells[static_cast<size_t>(y) * nx + x] != 0;
  int lo = known ? x : 0x7fffffff, hi = known ? x : -1;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    lo = min(lo, __shfl_xor(lo, off, 64));
    hi = max(hi, __shfl_xor(hi, off, 64));
  }
  if ((threadIdx.x & 63) == 0 && hi >= 0) {
    atomicMin(&box[0], lo);
    atomicMax(&box[2], hi);
    atomicMin(&box[1], y);
    atomicMax(&box[3], y);
  }
}

// ProbabilityGrid::ComputeCroppedGrid (probability_grid.cc:90-106): the known box copied into a
// grid of its own.  SetProbability(GetProbability(v)) is the identity on every value
// (tests/test_device_formulas.py), so the cells are copied as they are.
__global__ void GridCropKernel(const uint16_t* __restrict__ cells, int nx, int off_x, int off_y,
                               uint16_t* __restrict__ cropped, int cnx) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  if (x < cnx) cropped[static_cast<size_t>(y) * cnx + x] =
      cells[static_cast<size_t>(y + off_y) * nx + x + off_x];
}

bool Contains(const cmx_grid2d& g, float px, float py) {
  const long ix = std::lround((g.max_y - py) / g.resolution - 0.5);
  const long iy = std::lround((g.max_x - px) / g.resolution - 0.5);
  return ix >= 0 && ix < g.nx && iy >= 0 && iy < g.ny;
}

void GrowLimits(cmx_grid2d* g, Workspace& ws, float px, float py) {
  while (!Contains(*g, px, py)) {
    CMX_REQUIRE(static_cast<long long>(g->nx) * g->ny < (1ll << 28), "grid grows beyond 2^30 cells");
    const int x_offset = g->nx / 2, y_offset = g->ny / 2;
    const size_t new_count = 4 * static_cast<size_t>(g->nx) * g->ny;
    uint16_t* grown = nullptr;
    CMX_HIP(hipMalloc(reinterpret_cast<void**>(&grown), new_count * sizeof(uint16_t)));
    CMX_HIP(hipMemsetAsync(grown, 0, new_count * sizeof(uint16_t), ws.stream));   // unknown
    GridGrowKernel<<<dim3(DivUp(g->nx, 256), g->ny), 256, 0, ws.stream>>>(g->cells, g->nx, g->ny,
                                                                       grown, x_offset, y_offset);
    CMX_HIP(hipGetLastError());
    CMX_HIP(hipStreamSynchronize(ws.stream));
    CMX_HIP(hipFree(g->cells));
    g->cells = grown;
    ++g->version;
    g->max_x += g->resolution * y_offset;
    g->max_y += g->resolution * x_offset;
    g->nx *= 2;
    g->ny *= 2;
  }
}

const uint16_t* DeviceTable(cmx_grid2d* g, float probability) {
  CMX_REQUIRE(probability > 0.f && probability < 1.f, "probability must be in (0, 1)");
  uint32_t bits;
  std::memcpy(&bits, &probability, sizeof(bits));
  auto it = g->tables.find(bits);
  if (it != g->tables.end()) return it->second;
  std::vector<uint16_t> host(32768);
  OddsTable(probability, host.data());
  uint16_t* table = nullptr;
  CMX_HIP(hipMalloc(reinterpret_cast<void**>(&table), 32768 * sizeof(uint16_t)));
  CMX_HIP(hipMemcpy(table, host.data(), 32768 * sizeof(uint16_t), hipMemcpyHostToDevice));
  g->tables[bits] = table;
  return table;
}

}  // namespace
}  // namespace cmx

using cmx::Guard;

extern "C" cmx_status cmx_grid2d_create(const cmx_grid2d_limits* limits, const uint16_t* cells,
                                        int32_t device, cmx_grid2d** out) {
  return Guard([&] {
    CMX_REQUIRE(limits && out, "null argument");
    CMX_REQUIRE(limits->resolution > 0. && limits->num_x_cells >= 1 && limits->num_y_cells >= 1,
                "bad map limits");
    cmx::UseDevice(device);
    std::unique_ptr<cmx_grid2d> g(new cmx_grid2d);
    g->device = device;
    g->resolution = limits->resolution;
    g->max_x = limits->max_x;
    g->max_y = limits->max_y;
    g->nx = limits->num_x_cells;
    g->ny = limits->num_y_cells;
    const size_t bytes = static_cast<size_t>(g->nx) * g->ny * sizeof(uint16_t);
    CMX_HIP(hipMalloc(reinterpret_cast<void**>(&g->cells), bytes));
    if (cells) CMX_HIP(hipMemcpy(g->cells, cells, bytes, hipMemcpyHostToDevice));
    else CMX_HIP(hipMemset(g->cells, 0, bytes));         // kUnknownCorrespondenceValue
    *out = g.release();
  });
}

extern "C" void cmx_grid2d_destroy(cmx_grid2d* grid) {
  if (!grid) return;
  (void)hipSetDevice(grid->device);
  if (grid->cells) (void)hipFree(grid->cells);
  for (auto& kv : grid->tables) (void)hipFree(kv.second);
  delete grid;
}

extern "C" cmx_status cmx_grid2d_get_limits(const cmx_grid2d* grid, cmx_grid2d_limits* limits) {
  return Guard([&] {
    CMX_REQUIRE(grid && limits, "null argument");
    limits->resolution = grid->resolution;
    limits->max_x = grid->max_x;
    limits->max_y = grid->max_y;
    limits->num_x_cells = grid->nx;
    limits->num_y_cells = grid->ny;
    const float min_p = 0.1f, max_p = 1.f - min_p;
    limits->min_correspondence_cost = 1.f - max_p;
    limits->max_correspondence_cost = 1.f - min_p;
  });
}

extern "C" cmx_status cmx_grid2d_download(const cmx_grid2d* grid, uint16_t* cells) {
  return Guard([&] {
    CMX_REQUIRE(grid && cells, "null argument");
    cmx::UseDevice(grid->device);
    CMX_HIP(hipMemcpy(cells, grid->cells, static_cast<size_t>(grid->nx) * grid->ny * 2,
                      hipMemcpyDeviceToHost));
  });
}

extern "C" cmx_status cmx_grid2d_crop(cmx_grid2d* grid) {
  using namespace cmx;
  return Guard([&] {
    CMX_REQUIRE(grid != nullptr, "null argument");
    WorkspaceLease ws(grid->device);
    int* d_box = ws->dev[2].ReserveAs<int>(4);
    const int preset[4] = {0x7fffffff, 0x7fffffff, -1, -1};
    int* h_box = ws->pinned[0].ReserveAs<int>(4);
    std::memcpy(h_box, preset, sizeof(preset));
    CMX_HIP(hipMemcpyAsync(d_box, h_box, sizeof(preset), hipMemcpyHostToDevice, ws->stream));
    GridKnownBoxKernel<<<dim3(DivUp(grid->nx, 256), grid->ny), 256, 0, ws->stream>>>(
        grid->cells, grid->nx, grid->ny, d_box);
    CMX_HIP(hipGetLastError());
    CMX_HIP(hipMemcpyAsync(h_box, d_box, sizeof(preset), hipMemcpyDeviceToHost, ws->stream));
    CMX_HIP(hipStreamSynchronize(ws->stream));
    const bool empty = h_box[2] < 0;
    // ComputeCroppedLimits: no known cell -> offset 0, CellLimits(1, 1) (grid_2d.cc:106-110).
    const int off_x = empty ? 0 : h_box[0], off_y = empty ? 0 : h_box[1];
    const int cnx = empty ? 1 : h_box[2] - h_box[0] + 1, cny = empty ? 1 : h_box[3] - h_box[1] + 1;
    uint16_t* cropped = nullptr;
    const size_t bytes = static_cast<size_t>(cnx) * cny * sizeof(uint16_t);
    CMX_HIP(hipMalloc(reinterpret_cast<void**>(&cropped), bytes));
    if (empty) {
      CMX_HIP(hipMemsetAsync(cropped, 0, bytes, ws->stream));       // one unknown cell
    } else {
      GridCropKernel<<<dim3(DivUp(cnx, 256), cny), 256, 0, ws->stream>>>(
          grid->cells, grid->nx, off_x, off_y, cropped, cnx);
    }
    hipError_t err = hipGetLastError();
    if (err == hipSuccess) err = hipStreamSynchronize(ws->stream);
    if (err != hipSuccess) {
      (void)hipFree(cropped);
      CMX_HIP(err);
    }
    CMX_HIP(hipFree(grid->cells));
    grid->cells = cropped;
    ++grid->version;
    // max = limits().max() - resolution * (offset.y, offset.x) (probability_grid.cc:95-96).
    grid->max_x = grid->max_x - grid->resolution * off_y;
    grid->max_y = grid->max_y - grid->resolution * off_x;
    grid->nx = cnx;
    grid->ny = cny;
  });
}

extern "C" cmx_status cmx_grid2d_insert(cmx_grid2d* grid, const float* origin_xy,
                                        const float* returns_xyz, int32_t num_returns,
                                        const float* misses_xyz, int32_t num_misses,
                                        float hit_probability, float miss_probability,
                                        int32_t insert_free_space) {
  using namespace cmx;
  return Guard([&] {
    CMX_REQUIRE(grid && origin_xy, "null argument");
    CMX_REQUIRE(num_returns >= 0 && num_misses >= 0 && (num_returns == 0 || returns_xyz) &&
                    (num_misses == 0 || misses_xyz),
                "bad range data");
    WorkspaceLease ws(grid->device);
    // GrowAsNeeded (:35-50): bounding box of origin, returns and misses, padded.
    float lo_x = origin_xy[0], hi_x = origin_xy[0], lo_y = origin_xy[1], hi_y = origin_xy[1];
    const auto extend = [&](const float* p) {
      lo_x = std::min(lo_x, p[0]); hi_x = std::max(hi_x, p[0]);
      lo_y = std::min(lo_y, p[1]); hi_y = std::max(hi_y, p[1]);
    };
    for (int i = 0; i != num_returns; ++i) extend(returns_xyz + 3 * i);
    for (int i = 0; i != num_misses; ++i) extend(misses_xyz + 3 * i);
    constexpr float kPadding = 1e-6f;
    GrowLimits(grid, *ws, lo_x - kPadding * 1.f, lo_y - kPadding * 1.f);
    GrowLimits(grid, *ws, hi_x + kPadding * 1.f, hi_y + kPadding * 1.f);

    ++grid->version;                                     // the cells change below
    const uint16_t* hit_table = DeviceTable(grid, hit_probability);
    const uint16_t* miss_table = DeviceTable(grid, miss_probability);
    const int num_points = num_returns + num_misses;
    const size_t count = static_cast<size_t>(grid->nx) * grid->ny;
    int* d_error = ws->dev[2].ReserveAs<int>(1);
    CMX_HIP(hipMemsetAsync(d_error, 0, sizeof(int), ws->stream));
    if (num_points > 0) {
      float* h_points = ws->pinned[0].ReserveAs<float>(3 * static_cast<size_t>(num_points));
      if (num_returns) std::memcpy(h_points, returns_xyz, 3 * sizeof(float) * num_returns);
      if (num_misses)
        std::memcpy(h_points + 3 * static_cast<size_t>(num_returns), misses_xyz,
                    3 * sizeof(float) * num_misses);
      float* d_points = ws->dev[0].ReserveAs<float>(3 * static_cast<size_t>(num_points));
      int2* d_ends = ws->dev[1].ReserveAs<int2>(num_points);
      CMX_HIP(hipMemcpyAsync(d_points, h_points, 3 * sizeof(float) * num_points,
                             hipMemcpyHostToDevice, ws->stream));
      GridView view{grid->cells, grid->nx, grid->ny, grid->resolution / kSubpixelScale,
                    grid->max_x, grid->max_y};
      GridHitKernel<<<DivUp(num_points, 256), 256, 0, ws->stream>>>(
          view, d_points, num_returns, num_points, hit_table, d_ends, d_error);
      if (insert_free_space)
        GridMissKernel<<<DivUp(num_points, 4), 256, 0, ws->stream>>>(
            view, origin_xy[0], origin_xy[1], d_ends, num_points, miss_table, d_error);
      GridFinishKernel<<<DivUp(count, 256), 256, 0, ws->stream>>>(grid->cells, count);
      CMX_HIP(hipGetLastError());
    }
    int h_error = 0;
    CMX_HIP(hipMemcpyAsync(&h_error, d_error, sizeof(int), hipMemcpyDeviceToHost, ws->stream));
    CMX_HIP(hipStreamSynchronize(ws->stream));
    CMX_REQUIRE(!h_error, "internal error: range data left the grid after GrowLimits");
  });
}

namespace cmx {
// For the other translation units that read a resident grid (ceres_2d.hip).
const uint16_t* Grid2DDeviceCells(const cmx_grid2d* grid, cmx_grid2d_limits* limits, int* device) {
  limits->resolution = grid->resolution;
  limits->max_x = grid->max_x;
  limits->max_y = grid->max_y;
  limits->num_x_cells = grid->nx;
  limits->num_y_cells = grid->ny;
  // ProbabilityGrid(limits, tables): Grid2D(limits, kMinCorrespondenceCost, kMaxCorrespondenceCost)
  limits->min_correspondence_cost = 1.f - (1.f - 0.1f);
  limits->max_correspondence_cost = 1.f - 0.1f;
  *device = grid->device;
  return grid->cells;
}
}  // namespace cmx

extern "C" cmx_status cmx_rt2d_match_grid(const cmx_rt_options* options, const cmx_grid2d* grid,
                                          const cmx_pose2d* initial_pose_estimate,
                                          const float* point_cloud_xyz, int32_t num_points,
                                          double* score, cmx_pose2d* pose_estimate,
                                          cmx_match_stats* stats) {
  return Guard([&] {
    CMX_REQUIRE(grid != nullptr, "null argument");
    cmx_grid2d_limits limits;
    limits.resolution = grid->resolution;
    limits.max_x = grid->max_x;
    limits.max_y = grid->max_y;
    limits.num_x_cells = grid->nx;
    limits.num_y_cells = grid->ny;
    limits.min_correspondence_cost = 0.f;
    limits.max_correspondence_cost = 0.f;
    cmx::Rt2DItem item{};
    item.limits = &limits;
    item.device_cells = grid->cells;
    item.initial = initial_pose_estimate;
    item.xyz = point_cloud_xyz;
    item.n = num_points;
    item.score = score;
    item.pose = pose_estimate;
    item.image_cache = &grid->rt_image;
    item.grid_version = grid->version;
    CMX_REQUIRE(options != nullptr, "null argument");
    cmx::Rt2DMatchBatch(options, &item, 1, grid->device, stats);
  });
}

extern "C" cmx_status cmx_rt2d_match_grid_batch(const cmx_rt_options* options,
                                                const cmx_grid2d* const* grids,
                                                int32_t num_matches,
                                                const cmx_pose2d* initial_pose_estimates,
                                                const float* const* point_clouds_xyz,
                                                const int32_t* num_points, double* scores,
                                                cmx_pose2d* pose_estimates,
                                                cmx_match_stats* stats) {
  return Guard([&] {
    CMX_REQUIRE(grids && initial_pose_estimates && point_clouds_xyz && num_points && scores &&
                    pose_estimates && num_matches >= 1,
                "null argument");
    std::vector<cmx_grid2d_limits> limits(num_matches);
    std::vector<cmx::Rt2DItem> items(num_matches);
    for (int m = 0; m < num_matches; ++m) {
      const cmx_grid2d* g = grids[m];
      CMX_REQUIRE(g != nullptr, "null grid");
      CMX_REQUIRE(g->device == grids[0]->device, "all grids of a batch must live on one device");
      limits[m] = cmx_grid2d_limits{g->resolution, g->max_x, g->max_y, g->nx, g->ny, 0.f, 0.f};
      cmx::Rt2DItem item{};
      item.limits = &limits[m];
      item.device_cells = g->cells;
      item.initial = &initial_pose_estimates[m];
      item.xyz = point_clouds_xyz[m];
      item.n = num_points[m];
      item.score = &scores[m];
      item.pose = &pose_estimates[m];
      item.image_cache = &g->rt_image;
      item.grid_version = g->version;
      items[m] = item;
    }
    cmx::Rt2DMatchBatch(options, items.data(), num_matches, grids[0]->device, stats);
  });
}

extern "C" cmx_status cmx_rt2d_match_grid_batch_resident(const cmx_rt_options* options,
                                                         const cmx_grid2d* const* grids,
                                                         int32_t num_matches,
                                                         const cmx_pose2d* initial_pose_estimates,
                                                         const cmx_cloud* const* clouds,
                                                         double* scores,
                                                         cmx_pose2d* pose_estimates,
                                                         cmx_match_stats* stats) {
  return Guard([&] {
    CMX_REQUIRE(grids && initial_pose_estimates && clouds && scores && pose_estimates &&
                    num_matches >= 1,
                "null argument");
    std::vector<cmx_grid2d_limits> limits(num_matches);
    std::vector<cmx::Rt2DItem> items(num_matches);
    for (int m = 0; m < num_matches; ++m) {
      const cmx_grid2d* g = grids[m];
      const cmx_cloud* c = clouds[m];
      CMX_REQUIRE(g != nullptr && c != nullptr, "null grid or cloud");
      CMX_REQUIRE(g->device == grids[0]->device && c->device == g->device,
                  "all grids and clouds of a batch must live on one device");
      limits[m] = cmx_grid2d_limits{g->resolution, g->max_x, g->max_y, g->nx, g->ny, 0.f, 0.f};
      cmx::Rt2DItem item{};
      item.limits = &limits[m];
      item.device_cells = g->cells;
      item.initial = &initial_pose_estimates[m];
      item.xyz = c->host_xyz.data();          // the range scan of SearchParameters runs on the host
      item.device_xyz = c->xyz;
      if (!c->far_points.empty()) {
        item.far_points = c->far_points.data();
        item.num_far_points = static_cast<int>(c->far_points.size());
      }
      item.n = c->num_points;
      item.score = &scores[m];
      item.pose = &pose_estimates[m];
      item.image_cache = &g->rt_image;
      item.grid_version = g->version;
      items[m] = item;
    }
    cmx::Rt2DMatchBatch(options, items.data(), num_matches, grids[0]->device, stats);
  });
}

extern "C" cmx_status cmx_fast2d_create_from_grid(const cmx_fast2d_options* options,
                                                  const cmx_grid2d* grid, cmx_fast2d** out) {
  return Guard([&] {
    CMX_REQUIRE(options && grid && out, "null argument");
    // The finished submap's cells come back once; the stack is built on the device.
    std::vector<uint16_t> cells(static_cast<size_t>(grid->nx) * grid->ny);
    cmx::UseDevice(grid->device);
    CMX_HIP(hipMemcpy(cells.data(), grid->cells, cells.size() * 2, hipMemcpyDeviceToHost));
    cmx_grid2d_limits limits;
    cmx_status st = cmx_grid2d_get_limits(grid, &limits);
    if (st == CMX_OK) st = cmx_fast2d_create(options, &limits, cells.data(), grid->device, out);
    if (st != CMX_OK) throw cmx::HipError{st};           // last error already set
  });
}
