// RealTimeCorrelativeScanMatcher2D::Match on gfx950 (probability-grid branch).
//
// Reference: SM2/real_time_correlative_scan_matcher_2d.cc:61-75 (ComputeCandidateScore),
// :83-115 (GenerateExhaustiveSearchCandidates), :117-149 (Match), :151-176 (ScoreCandidates).
//
// Parity notes
//   * The reference accumulates the N probabilities of a candidate in f32, in
//     point order.  One thread per candidate performs exactly that sequence,
//     so the unweighted score is bit-identical (parallelism comes from the
//     thousands of candidates, not from splitting a sum).
//   * The exp(-(hypot*wt + |theta|*wr)^2) weight and the final first-maximum
//     rule are applied on the host with libm for the few candidates whose
//     device-weighted score is within 1e-5 (relative) of the device maximum,
//     so the returned score/pose cannot depend on the device's exp().
#include <algorithm>
#include <chrono>
#include <cmath>
#include <string>
#include <cstring>

#include "rt_2d_device.h"
#include "scan_matching_2d.h"

namespace cmx {
namespace {

// Design (MI355X)
//   The grid is first expanded on the device into a *padded f32 score grid*:
//   element (x, y) holds the value a point falling into cell (x, y) adds to a
//   candidate's running sum (the probability; for a TSDF the pair
//   (normalised tsd score * weight, weight)), and a border of `pad` = 2*nl + 1
//   cells on every side holds the out-of-bounds value.  A rotated scan then
//   is one int32 per point: the linear offset of its (clamped) cell in that
//   grid.  A wavefront scores 64 candidates of one rotation: lanes run along
//   x offsets first, so one gather instruction reads a few contiguous row
//   segments, the point offset is wave-uniform (scalar loads), and the
//   per-candidate f32 sum is the reference's sequential chain.  Loads are
//   software pipelined kBatch deep so the chain never waits on memory.

struct Rt2DParams {
  const uint16_t* cells;     // device grid (probability values / tsd values)
  const uint16_t* weights;   // TSDF weight cells (nullptr for a probability grid)
  int nx, ny;
  double res, max_x, max_y;
  double inv_res;            // RN(1 / res), for the division-free cell index (cmx_device.h)
  float tx, ty, init_qw, init_qz;
  int nl, num_scans, num_angular;
  double step, wt, wr;
  float max_tsd, max_weight;  // TSDF only
  const float2* scan_rot;
  int* offsets;              // [num_scans][n_pad] byte offsets into `padded` (see PrepKernel)
  int n_pad;                 // n rounded up to a multiple of 64
  void* padded;              // float[rows][stride] or float2[rows][stride]
  int pad, stride, rows;
  unsigned* misc;            // [0] max weighted score bits, [1] finalist count, then pairs
  unsigned* overflow;        // finalist pairs beyond kFinalistHead
  const float* xyz;          // device point cloud
  int n;
  float* unweighted;         // [num_candidates]
  float* weighted;
  int num_candidates;
  int prep_blocks;           // num_scans + blocks of the grid expansion
};

// ValueConversionTables (mapping/value_conversion_tables.cc:29-52): value 0 ->
// `unknown`, [1, 32767] -> [lower, upper]; bit 15 (update marker) is masked.
__device__ __forceinline__ float BoundedValue(unsigned raw, float unknown, float lower,
                                              float upper) {
  const unsigned v = raw & 32767u;
  if (v == 0) return unknown;
  const float scale = (upper - lower) / 32766.f;
  return static_cast<float>(v) * scale + (lower - scale);
}

// The (term, weight) a TSDF cell contributes (real_time_..._2d.cc:38-59,
// mapping/internal/2d/tsdf_2d.cc:88-98, tsd_value_converter.cc:22-33).
__device__ __forceinline__ float2 TsdfTerm(float tsd, float weight, float max_tsd) {
  const float normalized = (max_tsd - fabsf(tsd)) / max_tsd;
  return make_float2(normalized * weight, weight);
}

// Every kernel serves a batch of independent matches: blockIdx.z picks the match, blocks
// beyond a match's own extent return at once.
template <bool kTsdf>
__global__ void __launch_bounds__(256)
Rt2DPrepKernel(const Rt2DParams* __restrict__ params) {
  const Rt2DParams& P = params[blockIdx.z];
  if (static_cast<int>(blockIdx.x) >= P.prep_blocks) return;
  const float* __restrict__ xyz = P.xyz;
  const int n = P.n;
  if (blockIdx.x == 0 && threadIdx.x == 0) { P.misc[0] = 0u; P.misc[1] = 0u; }
  if (blockIdx.x < P.num_scans) {
    const int s = blockIdx.x;
    const Quat q0{P.init_qw, 0.f, 0.f, P.init_qz};
    const float2 r = P.scan_rot[s];
    const Quat qs{r.x, 0.f, 0.f, r.y};
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      const F3 p{xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
      F3 a = Rotate(q0, p);
      a.x += 0.f; a.y += 0.f; a.z += 0.f;
      F3 b = Rotate(qs, a);
      b.x += 0.f; b.y += 0.f;
      const float x = (1.f * b.x + 0.f * b.y) + P.tx;
      const float y = (0.f * b.x + 1.f * b.y) + P.ty;
      int ix = LRoundF64((P.max_y - static_cast<double>(y)) / P.res - 0.5);
      int iy = LRoundF64((P.max_x - static_cast<double>(x)) / P.res - 0.5);
      // A coordinate further than nl outside the grid is out of bounds for every
      // offset; clamping it to nl + 1 outside keeps it so and inside the border.
      ix = min(max(ix, -(P.nl + 1)), P.nx + P.nl);
      iy = min(max(iy, -(P.nl + 1)), P.ny + P.nl);
      // Byte offset of cell (ix - nl, iy - nl): lanes add their (dx + nl, dy + nl).
      const int element = (iy + P.pad - P.nl) * P.stride + (ix + P.pad - P.nl);
      P.offsets[static_cast<size_t>(s) * P.n_pad + i] =
          element * static_cast<int>(kTsdf ? sizeof(float2) : sizeof(float));
    }
    // Padding points read the zero cell that follows the grid.
    for (int i = n + threadIdx.x; i < P.n_pad; i += blockDim.x)
      P.offsets[static_cast<size_t>(s) * P.n_pad + i] =
          P.stride * P.rows * static_cast<int>(kTsdf ? sizeof(float2) : sizeof(float));
    return;
  }
  // Remaining blocks expand the grid into the padded score grid.
  const int total = P.stride * P.rows;
  if (blockIdx.x == P.num_scans && threadIdx.x == 0) {   // the zero cell
    if constexpr (kTsdf) static_cast<float2*>(P.padded)[total] = make_float2(0.f, 0.f);
    else static_cast<float*>(P.padded)[total] = 0.f;
  }
  const int per_block = 256 * 4;
  const int base = (blockIdx.x - P.num_scans) * per_block;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int e = base + k * 256 + threadIdx.x;
    if (e >= total) break;
    const int y = e / P.stride - P.pad, x = e % P.stride - P.pad;
    const bool inside = static_cast<unsigned>(x) < static_cast<unsigned>(P.nx) &&
                        static_cast<unsigned>(y) < static_cast<unsigned>(P.ny);
    const int flat = inside ? P.nx * y + x : 0;
    if constexpr (kTsdf) {
      const float min_tsd = -P.max_tsd;
      float tsd = min_tsd, weight = 0.f;             // getMinTSD / getMinWeight outside
      if (inside) {
        tsd = BoundedValue(P.cells[flat], min_tsd, min_tsd, P.max_tsd);
        weight = BoundedValue(P.weights[flat], 0.f, 0.f, P.max_weight);
      }
      static_cast<float2*>(P.padded)[e] = TsdfTerm(tsd, weight, P.max_tsd);
    } else {
      static_cast<float*>(P.padded)[e] = inside ? CellProbability(P.cells[flat]) : 0.1f;
    }
  }
}

template <bool kTsdf>
struct Acc;
template <>
struct Acc<false> {
  using Cell = float;
  float sum = 0.f;
  __device__ __forceinline__ void Add(float v) { sum += v; }
  __device__ __forceinline__ float Finish(int n) const { return sum / static_cast<float>(n); }
};
template <>
struct Acc<true> {
  using Cell = float2;
  float sum = 0.f, weight = 0.f;
  __device__ __forceinline__ void Add(float2 v) { sum += v.x; weight += v.y; }
  __device__ __forceinline__ float Finish(int) const {
    return weight == 0.f ? 0.f : sum / weight;
  }
};

// grid (ceil(side^2 / 64), num_scans), one wavefront per block.
template <bool kTsdf>
__global__ void __launch_bounds__(64)
Rt2DScoreKernel(const Rt2DParams* __restrict__ params) {
  using Cell = typename Acc<kTsdf>::Cell;
  const Rt2DParams& P = params[blockIdx.z];
  const int s = blockIdx.y;
  const int side = 2 * P.nl + 1;
  if (s >= P.num_scans || static_cast<int>(blockIdx.x) * 64 >= side * side) return;
  const int n = P.n;
  float* __restrict__ unweighted = P.unweighted;
  float* __restrict__ weighted = P.weighted;
  const int rem = blockIdx.x * 64 + threadIdx.x;
  const bool valid = rem < side * side;
  const int r = valid ? rem : 0;
  const int dyi = r / side, dxi = r - dyi * side;     // lanes run along x offsets
  // Buffer addressing: descriptor (SGPRs) + per-lane byte offset (one VGPR, constant) +
  // wave-uniform point offset (SGPR): no vector address arithmetic per gather.
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      P.padded, 0, (P.stride * P.rows + 1) * static_cast<int>(sizeof(Cell)), 0x00020000);
  const int lane_off = (dyi * P.stride + dxi) * static_cast<int>(sizeof(Cell));
  // Point offsets: n_pad (a multiple of 64) per rotation; the padding points at the
  // grid's trailing zero cell, and x + 0.f == x keeps the sums bit-exact.
  const int* __restrict__ offs = P.offsets + static_cast<size_t>(s) * P.n_pad;
  const auto gather = [&](int off) -> Cell {
    if constexpr (kTsdf) {
      const auto v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, lane_off, off, 0);
      return make_float2(__uint_as_float(v[0]), __uint_as_float(v[1]));
    } else {
      return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, lane_off, off, 0));
    }
  };

  // A super-chunk = kChunks x 64 points whose offsets sit in kChunks VGPRs (lane l of
  // register c holds point 64 c + l); v_readlane turns them into the scalar offset of
  // each gather.  The next super-chunk's offsets are fetched while this one is summed,
  // and gathers run kBatch ahead of the sequential adds in two alternating banks.
  constexpr int kChunks = 16;
  constexpr int kBatch = kTsdf ? 16 : 32;
  constexpr int kPerChunk = 64 / kBatch;
  const int lane = threadIdx.x;
  const int chunks = P.n_pad / 64;
  Acc<kTsdf> acc;
  int ov[kChunks], ovn[kChunks];
#pragma unroll
  for (int c = 0; c < kChunks; ++c) ov[c] = c < chunks ? offs[c * 64 + lane] : 0;
  for (int base = 0; base < chunks; base += kChunks) {
#pragma unroll
    for (int c = 0; c < kChunks; ++c)
      ovn[c] = base + kChunks + c < chunks ? offs[(base + kChunks + c) * 64 + lane] : 0;
    const int live = min(kChunks, chunks - base);
    Cell a[kBatch], b[kBatch];
#pragma unroll
    for (int k = 0; k < kBatch; ++k) a[k] = gather(__builtin_amdgcn_readlane(ov[0], k));
#pragma unroll
    for (int c = 0; c < kChunks; ++c) {
      if (c < live) {
#pragma unroll
        for (int h = 0; h < kPerChunk; ++h) {
          Cell* cur = (h & 1) ? b : a;
          Cell* nxt = (h & 1) ? a : b;
          if (h + 1 < kPerChunk) {
#pragma unroll
            for (int k = 0; k < kBatch; ++k)
              nxt[k] = gather(__builtin_amdgcn_readlane(ov[c], (h + 1) * kBatch + k));
          } else if (c + 1 < kChunks) {
            if (c + 1 < live) {
#pragma unroll
              for (int k = 0; k < kBatch; ++k)
                nxt[k] = gather(__builtin_amdgcn_readlane(ov[c + 1], k));
            }
          }
#pragma unroll
          for (int k = 0; k < kBatch; ++k) acc.Add(cur[k]);
        }
      }
    }
#pragma unroll
    for (int c = 0; c < kChunks; ++c) ov[c] = ovn[c];
  }

  float w = 0.f;
  if (valid) {
    const float score = acc.Finish(n);
    const int dx = dxi - P.nl, dy = dyi - P.nl;
    const int c = (s * side + (dx + P.nl)) * side + (dy + P.nl);   // x outer, y inner (:99-113)
    unweighted[c] = score;
    const double cx = -dy * P.res, cy = -dx * P.res;
    const double theta = (s - P.num_angular) * P.step;
    const double t = hypot(cx, cy) * P.wt + fabs(theta) * P.wr;
    w = static_cast<float>(static_cast<double>(score) * exp(-(t * t)));
    weighted[c] = w;
  }
  unsigned bits = __float_as_uint(w);   // scores are >= 0
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) bits = max(bits, __shfl_xor(bits, off, 64));
  if (threadIdx.x == 0) atomicMax(&P.misc[0], bits);
}

// Probability grid, four candidates per lane: a lane owns the x offsets 4g .. 4g+3 of one
// y offset and reads their four cells with ONE dwordx4 gather (the cells are adjacent in a
// row of the padded grid).  A rotation's 13 x 13 window then is a single wavefront
// (13 rows x 4 groups = 52 lanes), a quarter of the gather instructions of the scalar
// variant; the four f32 sums per lane stay sequential in point order.
// grid (ceil(side * ceil(side/4) / 64), num_scans, matches), one wavefront per block.
__global__ void __launch_bounds__(64)
Rt2DScoreX4Kernel(const Rt2DParams* __restrict__ params) {
  const Rt2DParams& P = params[blockIdx.z];
  const int s = blockIdx.y;
  const int side = 2 * P.nl + 1;
  const int groups = (side + 3) / 4;
  if (s >= P.num_scans || static_cast<int>(blockIdx.x) * 64 >= side * groups) return;
  const int n = P.n;
  const int slot = blockIdx.x * 64 + threadIdx.x;
  const bool lane_valid = slot < side * groups;
  const int r = lane_valid ? slot : 0;
  const int dyi = r / groups, g = r - dyi * groups;
  // The last group of a row may reach up to three cells past the window (and, in the
  // very last row, past the grid: the buffer descriptor returns 0 there); those sums
  // belong to no candidate and are dropped.
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      P.padded, 0, (P.stride * P.rows + 1) * 4, 0x00020000);
  const int lane_off = (dyi * P.stride + 4 * g) * 4;
  const int* __restrict__ offs = P.offsets + static_cast<size_t>(s) * P.n_pad;
  typedef float float4v __attribute__((ext_vector_type(4)));
  const auto gather = [&](int off) -> float4v {
    const auto v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_off, off, 0);
    return float4v{__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]),
                   __uint_as_float(v[3])};
  };

  constexpr int kChunks = 16;
  constexpr int kBatch = 8;                 // x 16 bytes per lane in flight
  constexpr int kPerChunk = 64 / kBatch;
  const int lane = threadIdx.x;
  const int chunks = P.n_pad / 64;
  float4v acc = {0.f, 0.f, 0.f, 0.f};
  int ov[kChunks], ovn[kChunks];
#pragma unroll
  for (int c = 0; c < kChunks; ++c) ov[c] = c < chunks ? offs[c * 64 + lane] : 0;
  for (int base = 0; base < chunks; base += kChunks) {
#pragma unroll
    for (int c = 0; c < kChunks; ++c)
      ovn[c] = base + kChunks + c < chunks ? offs[(base + kChunks + c) * 64 + lane] : 0;
    const int live = min(kChunks, chunks - base);
    float4v a[kBatch], b[kBatch];
#pragma unroll
    for (int k = 0; k < kBatch; ++k) a[k] = gather(__builtin_amdgcn_readlane(ov[0], k));
#pragma unroll
    for (int c = 0; c < kChunks; ++c) {
      if (c < live) {
#pragma unroll
        for (int h = 0; h < kPerChunk; ++h) {
          float4v* cur = (h & 1) ? b : a;
          float4v* nxt = (h & 1) ? a : b;
          if (h + 1 < kPerChunk) {
#pragma unroll
            for (int k = 0; k < kBatch; ++k)
              nxt[k] = gather(__builtin_amdgcn_readlane(ov[c], (h + 1) * kBatch + k));
          } else if (c + 1 < kChunks) {
            if (c + 1 < live) {
#pragma unroll
              for (int k = 0; k < kBatch; ++k)
                nxt[k] = gather(__builtin_amdgcn_readlane(ov[c + 1], k));
            }
          }
#pragma unroll
          for (int k = 0; k < kBatch; ++k) acc += cur[k];   // four independent f32 chains
        }
      }
    }
#pragma unroll
    for (int c = 0; c < kChunks; ++c) ov[c] = ovn[c];
  }

  float w_max = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int dxi = 4 * g + j;
    if (lane_valid && dxi < side) {
      const float score = acc[j] / static_cast<float>(n);
      const int dx = dxi - P.nl, dy = dyi - P.nl;
      const int c = (s * side + dxi) * side + dyi;            // x outer, y inner (:99-113)
      P.unweighted[c] = score;
      const double cx = -dy * P.res, cy = -dx * P.res;
      const double theta = (s - P.num_angular) * P.step;
      const double t = hypot(cx, cy) * P.wt + fabs(theta) * P.wr;
      const float w = static_cast<float>(static_cast<double>(score) * exp(-(t * t)));
      P.weighted[c] = w;
      w_max = fmaxf(w_max, w);
    }
  }
  unsigned bits = __float_as_uint(w_max);   // scores are >= 0
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) bits = max(bits, __shfl_xor(bits, off, 64));
  if (threadIdx.x == 0) atomicMax(&P.misc[0], bits);
}

// Candidates whose device-weighted score is within 1e-5 of the maximum, with their exact
// unweighted score: (index, score bits) pairs (kFinalistCap / kFinalistHead, rt_2d_device.h).
__global__ void Rt2DCollectKernel(const Rt2DParams* __restrict__ params) {
  const Rt2DParams& P = params[blockIdx.z];
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= P.num_candidates) return;
  const float threshold = __uint_as_float(P.misc[0]) * (1.f - 1e-5f);
  if (P.weighted[c] >= threshold) {
    const unsigned slot = atomicAdd(&P.misc[1], 1u);
    if (slot < static_cast<unsigned>(kFinalistCap)) {
      unsigned* pair = slot < static_cast<unsigned>(kFinalistHead)
                           ? P.misc + 2 + 2 * slot
                           : P.overflow + 2 * (slot - kFinalistHead);
      pair[0] = static_cast<unsigned>(c);
      pair[1] = __float_as_uint(P.unweighted[c]);
    }
  }
}

size_t Align16(size_t v) { return (v + 15) & ~static_cast<size_t>(15); }

}  // namespace

// A batch of independent matches (one per trajectory / robot) in one set of launches, on the
// one-thread-per-candidate kernels: TSDFs, flat score landscapes and whatever the tile path
// (rt_2d_tiles.hip) does not take.  Per item `cells` is a host buffer, or -- when
// `device_cells` is given -- ignored in favour of a grid that already lives in HBM (cmx_grid2d).
namespace {
void Rt2DLegacyBatch(const cmx_rt_options* options, const Rt2DItem* items, const Rt2DSearch* search,
                     int num, int32_t device, cmx_match_stats* stats) {
  struct Plan {
    int n, nx, ny, nl, na, num_scans, n_pad, pad;
    long long side, num_candidates, stride, rows;
    size_t off_xyz, off_rot, off_cells, off_weights;          // in the staging buffer
    size_t off_offsets, off_padded, off_scores;                // element offsets, device
  };
  std::vector<Plan> plan(num);
  const bool tsdf = items[0].weight_cells != nullptr;
  size_t in_bytes = Align16(sizeof(Rt2DParams) * num);
  size_t offsets_total = 0, padded_bytes = 0, scores_total = 0;
  unsigned max_prep = 0, max_tiles = 0, max_tiles4 = 0, max_scans = 0, max_collect = 0;
  for (int m = 0; m < num; ++m) {
    const Rt2DItem& it = items[m];
    Plan& pl = plan[m];
    pl.n = it.n; pl.nx = it.limits->num_x_cells; pl.ny = it.limits->num_y_cells;
    pl.na = search[m].na; pl.num_scans = search[m].num_scans; pl.nl = search[m].nl;
    const int n = pl.n;
    pl.side = 2ll * pl.nl + 1;
    pl.num_candidates = pl.side * pl.side * pl.num_scans;
    pl.pad = 2 * pl.nl + 1;
    pl.stride = pl.nx + 2ll * pl.pad;
    pl.rows = pl.ny + 2ll * pl.pad;
    CMX_REQUIRE(pl.stride * pl.rows < (1ll << 27), "grid plus search window too large");
    CMX_REQUIRE(static_cast<long long>(pl.num_scans) * n < (1ll << 30), "too many rotated points");
    pl.n_pad = (n + 63) / 64 * 64;

    // Staging buffer: [params | per item: xyz | rotations | cells | weight cells].
    const size_t cell_count = static_cast<size_t>(pl.nx) * pl.ny;
    pl.off_xyz = in_bytes;
    pl.off_rot = pl.off_xyz + (it.device_xyz ? 0 : Align16(3 * sizeof(float) * n));
    pl.off_cells = pl.off_rot + Align16(sizeof(float2) * pl.num_scans);
    pl.off_weights =
        pl.off_cells + (it.device_cells ? 0 : Align16(sizeof(uint16_t) * cell_count));
    in_bytes = pl.off_weights + (tsdf ? Align16(sizeof(uint16_t) * cell_count) : 0);
    pl.off_offsets = offsets_total;
    offsets_total += static_cast<size_t>(pl.num_scans) * pl.n_pad;
    pl.off_padded = padded_bytes;
    padded_bytes += Align16(static_cast<size_t>(pl.stride * pl.rows + 1) *
                            (tsdf ? sizeof(float2) : sizeof(float)));
    pl.off_scores = scores_total;
    scores_total += static_cast<size_t>(pl.num_candidates);
    max_prep = std::max<unsigned>(max_prep, pl.num_scans + DivUp(pl.stride * pl.rows, 1024));
    max_tiles = std::max<unsigned>(max_tiles, DivUp(pl.side * pl.side, 64));
    max_tiles4 = std::max<unsigned>(max_tiles4, DivUp(pl.side * ((pl.side + 3) / 4), 64));
    max_scans = std::max<unsigned>(max_scans, pl.num_scans);
    max_collect = std::max<unsigned>(max_collect, DivUp(pl.num_candidates, 256));
  }
  CMX_REQUIRE(num <= 65535, "too many matches in one batch");
  // The per-match result words ride in the upload (zeroed) so that no kernel has to clear them.
  const size_t off_misc = in_bytes;
  in_bytes += Align16(sizeof(unsigned) * 128 * static_cast<size_t>(num));

  WorkspaceLease ws(device);
  char* h_in = ws->pinned[0].ReserveAs<char>(in_bytes);
  char* d_in = ws->dev[0].ReserveAs<char>(in_bytes);
  int* d_offsets = ws->dev[1].ReserveAs<int>(offsets_total);
  char* d_padded = ws->dev[2].ReserveAs<char>(padded_bytes);
  float* d_unweighted = ws->dev[3].ReserveAs<float>(scores_total);
  float* d_weighted = ws->dev[4].ReserveAs<float>(scores_total);
  static_assert(2 + 2 * kFinalistHead <= 128, "a match's head must fit its 128-word slot");
  unsigned* d_misc = reinterpret_cast<unsigned*>(d_in + off_misc);
  std::memset(h_in + off_misc, 0, sizeof(unsigned) * 128 * static_cast<size_t>(num));
  unsigned* d_overflow = ws->dev[6].ReserveAs<unsigned>(static_cast<size_t>(num) * 2 *
                                                        (kFinalistCap - kFinalistHead));
  unsigned* h_misc = ws->pinned[1].ReserveAs<unsigned>(static_cast<size_t>(num) * 128);

  Rt2DParams* h_params = reinterpret_cast<Rt2DParams*>(h_in);
  ParallelFor(num, 8, [&](int m) {
    const Rt2DItem& it = items[m];
    const Plan& pl = plan[m];
    const Rt2DSearch& sr = search[m];
    const size_t cell_count = static_cast<size_t>(pl.nx) * pl.ny;
    if (!it.device_xyz) std::memcpy(h_in + pl.off_xyz, it.xyz, 3 * sizeof(float) * pl.n);
    const auto table = HostRotationTable(sr.step, sr.na);
    std::memcpy(h_in + pl.off_rot, table->data(), sizeof(float2) * pl.num_scans);
    if (!it.device_cells)
      std::memcpy(h_in + pl.off_cells, it.cells, sizeof(uint16_t) * cell_count);
    if (tsdf) std::memcpy(h_in + pl.off_weights, it.weight_cells, sizeof(uint16_t) * cell_count);

    Rt2DParams P{};
    P.cells = it.device_cells ? it.device_cells
                              : reinterpret_cast<const uint16_t*>(d_in + pl.off_cells);
    P.weights = tsdf ? reinterpret_cast<const uint16_t*>(d_in + pl.off_weights) : nullptr;
    P.nx = pl.nx; P.ny = pl.ny;
    P.res = it.limits->resolution; P.max_x = it.limits->max_x; P.max_y = it.limits->max_y;
    P.inv_res = 1.0 / P.res;
    P.tx = static_cast<float>(it.initial->x);
    P.ty = static_cast<float>(it.initial->y);
    P.init_qw = sr.q0w; P.init_qz = sr.q0z;
    P.nl = pl.nl; P.num_scans = pl.num_scans; P.num_angular = pl.na;
    P.step = sr.step;
    P.wt = options->translation_delta_cost_weight;
    P.wr = options->rotation_delta_cost_weight;
    P.max_tsd = it.max_tsd; P.max_weight = it.max_weight;
    P.scan_rot = reinterpret_cast<const float2*>(d_in + pl.off_rot);
    P.offsets = d_offsets + pl.off_offsets;
    P.n_pad = pl.n_pad;
    P.padded = d_padded + pl.off_padded;
    P.pad = pl.pad; P.stride = static_cast<int>(pl.stride); P.rows = static_cast<int>(pl.rows);
    P.misc = d_misc + static_cast<size_t>(m) * 128;
    P.overflow = d_overflow + static_cast<size_t>(m) * 2 * (kFinalistCap - kFinalistHead);
    P.xyz = it.device_xyz ? it.device_xyz : reinterpret_cast<const float*>(d_in + pl.off_xyz);
    P.n = pl.n;
    P.unweighted = d_unweighted + pl.off_scores;
    P.weighted = d_weighted + pl.off_scores;
    P.num_candidates = static_cast<int>(pl.num_candidates);
    P.prep_blocks = pl.num_scans + static_cast<int>(DivUp(pl.stride * pl.rows, 1024));
    h_params[m] = P;
  });
  SmallCopyAsync(d_in, h_in, in_bytes, /*to_device=*/true, ws->stream);
  const Rt2DParams* d_params = reinterpret_cast<const Rt2DParams*>(d_in);

  RecordEvent(ws->ev_begin, ws->stream);
  const dim3 prep_grid(max_prep, 1, num), score_grid(max_tiles, max_scans, num),
      collect_grid(max_collect, 1, num);
  if (tsdf) {
    Rt2DPrepKernel<true><<<prep_grid, 256, 0, ws->stream>>>(d_params);
    RecordEvent(ws->ev_k0, ws->stream);
    Rt2DScoreKernel<true><<<score_grid, 64, 0, ws->stream>>>(d_params);
  } else {
    Rt2DPrepKernel<false><<<prep_grid, 256, 0, ws->stream>>>(d_params);
    RecordEvent(ws->ev_k0, ws->stream);
    // One match (81 waves) is bound by the latency of its sequential sums: the scalar
    // variant keeps 32 gathers in flight per wave (C1: 19 us vs 34 us).  From ~16
    // concurrent matches on, the gather path is the limit and four candidates per
    // gather win (128 matches: 169 us vs 293 us, 3.5e9 candidates/s).
    if (num >= 16)
      Rt2DScoreX4Kernel<<<dim3(max_tiles4, max_scans, num), 64, 0, ws->stream>>>(d_params);
    else
      Rt2DScoreKernel<false><<<score_grid, 64, 0, ws->stream>>>(d_params);
  }
  RecordEvent(ws->ev_k1, ws->stream);
  Rt2DCollectKernel<<<collect_grid, 256, 0, ws->stream>>>(d_params);
  CMX_HIP(hipGetLastError());
  RecordEvent(ws->ev_end, ws->stream);
  SmallCopyAsync(h_misc, d_misc, sizeof(unsigned) * 128 * num, /*to_device=*/false, ws->stream);
  CMX_HIP(hipStreamSynchronize(ws->stream));

  cmx_match_stats total{};
  std::vector<std::pair<int, float>> finalists;
  std::vector<unsigned> extra;
  std::vector<float> all;
  for (int m = 0; m < num; ++m) {
    const Plan& pl = plan[m];
    const unsigned* head = h_misc + static_cast<size_t>(m) * 128;
    const long long count = head[1];
    finalists.clear();
    if (count <= kFinalistCap) {
      finalists.resize(count);
      const long long in_head = std::min<long long>(count, kFinalistHead);
      if (count > kFinalistHead) {
        extra.resize(2 * (count - kFinalistHead));
        CMX_HIP(hipMemcpyAsync(extra.data(),
                               d_overflow + static_cast<size_t>(m) * 2 *
                                                (kFinalistCap - kFinalistHead),
                               sizeof(unsigned) * extra.size(), hipMemcpyDeviceToHost,
                               ws->stream));
        CMX_HIP(hipStreamSynchronize(ws->stream));
      }
      for (long long i = 0; i < count; ++i) {
        const unsigned* pair = i < in_head ? head + 2 + 2 * i : extra.data() + 2 * (i - in_head);
        float v;
        std::memcpy(&v, &pair[1], sizeof(float));
        finalists[i] = {static_cast<int>(pair[0]), v};
      }
      std::sort(finalists.begin(), finalists.end());
    } else {  // flat score landscape: take everything
      all.resize(pl.num_candidates);
      CMX_HIP(hipMemcpyAsync(all.data(), d_unweighted + pl.off_scores,
                             sizeof(float) * pl.num_candidates, hipMemcpyDeviceToHost,
                             ws->stream));
      CMX_HIP(hipStreamSynchronize(ws->stream));
      finalists.resize(pl.num_candidates);
      for (long long c = 0; c < pl.num_candidates; ++c)
        finalists[c] = {static_cast<int>(c), all[c]};
    }
    CMX_REQUIRE(!finalists.empty(), "internal error: no candidate collected");
    Rt2DFinishOnHost(options, items[m], search[m], finalists.data(), finalists.size());
    total.candidates_scored += pl.num_candidates;
    total.coarse_candidates += pl.num_candidates;
    total.num_scans += pl.num_scans;
    total.finalists += pl.num_candidates;        // every score is the reference's own f32 sum here
  }
  if (stats) {
    float ms = 0.f;
    ms = ElapsedMs(ws->ev_begin, ws->ev_end);
    total.device_ms = ms;
    ms = ElapsedMs(ws->ev_k0, ws->ev_k1);
    total.dominant_kernel_ms = ms;
    *stats = total;
  }
}

// ScoreCandidates (SM2/real_time_correlative_scan_matcher_2d.cc:147-175), the method the
// reference keeps "visible for testing": ANY list of (scan, x offset, y offset) over discrete
// scans handed in by the caller.  One thread per candidate, the f32 sums in point order
// (ComputeCandidateScore, :38-73); cells outside the limits read kMinProbability / (min tsd,
// weight 0) (probability_grid.cc:78-82, tsdf_2d.cc:88-98).
template <bool kTsdf>
__global__ void Rt2DScoreCandidatesKernel(const uint16_t* __restrict__ cells,
                                          const uint16_t* __restrict__ weights, int nx, int ny,
                                          float max_tsd, float max_weight,
                                          const int* __restrict__ scans_xy,
                                          const int* __restrict__ scan_begin,
                                          const int4* __restrict__ candidates, int num,
                                          float* __restrict__ unweighted) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= num) return;
  const int4 cand = candidates[c];                       // scan_index, x offset, y offset
  const int begin = scan_begin[cand.x], end = scan_begin[cand.x + 1];
  Acc<kTsdf> acc;
  for (int i = begin; i < end; ++i) {
    const int x = scans_xy[2 * i] + cand.y, y = scans_xy[2 * i + 1] + cand.z;
    const bool inside = static_cast<unsigned>(x) < static_cast<unsigned>(nx) &&
                        static_cast<unsigned>(y) < static_cast<unsigned>(ny);
    const int flat = inside ? nx * y + x : 0;
    if constexpr (kTsdf) {
      const float min_tsd = -max_tsd;
      float tsd = min_tsd, weight = 0.f;
      if (inside) {
        tsd = BoundedValue(cells[flat], min_tsd, min_tsd, max_tsd);
        weight = BoundedValue(weights[flat], 0.f, 0.f, max_weight);
      }
      acc.Add(TsdfTerm(tsd, weight, max_tsd));
    } else {
      acc.Add(inside ? CellProbability(cells[flat]) : 0.1f);
    }
  }
  unweighted[c] = acc.Finish(end - begin);
}
}  // namespace

void Rt2DMatchBatch(const cmx_rt_options* options, const Rt2DItem* items, int num, int32_t device,
                    cmx_match_stats* stats) {
  CMX_REQUIRE(options && items && num >= 1, "null argument");
  // Debug switch host_trace: wall clock of the host phases (tools only).
  const bool host_trace = Debug().host_trace != 0;
  auto t_last = std::chrono::steady_clock::now();
  const auto t_call = t_last;
  std::string host_report;
  const auto lap = [&](const char* name) {
    if (!host_trace) return;
    const auto now = std::chrono::steady_clock::now();
    char buf[64];
    snprintf(buf, sizeof buf, " %s=%.0f", name,
             std::chrono::duration<double, std::micro>(now - t_last).count());
    host_report += buf;
    t_last = now;
  };
  const bool tsdf = items[0].weight_cells != nullptr;
  for (int m = 0; m < num; ++m) {
    const Rt2DItem& it = items[m];
    CMX_REQUIRE(it.limits && (it.cells || it.device_cells) && it.initial && it.xyz,
                "null argument");
    CMX_REQUIRE(it.pose != nullptr && it.score != nullptr,
                "pose_estimate must not be null");            // CHECK at :121
    CMX_REQUIRE(it.n >= 1 && it.n <= (1 << 24), "bad point count");
    CMX_REQUIRE(it.limits->resolution > 0. && it.limits->num_x_cells >= 1 &&
                    it.limits->num_y_cells >= 1,
                "bad map limits");
    CMX_REQUIRE((it.weight_cells != nullptr) == tsdf, "mixed grid types in one batch");
    if (tsdf) CMX_REQUIRE(it.max_tsd > 0.f && it.max_weight > 0.f, "bad TSDF ranges");
  }
  // SearchParameters of every item (a range scan over its cloud, acos): on the host pool, part
  // by part (below), so that the first part's kernels start under the planning of the others.
  std::vector<Rt2DSearch> search(num);
  lap("args");
  const auto plan_search = [&](int begin, int end) {
    ParallelFor(end - begin, Debug().rt2d_host_par > 0 ? Debug().rt2d_host_par : 4096,   // (0.03 us per match; a pool dispatch costs ~25 us)
                [&](int k) { Rt2DComputeSearch(options, items[begin + k], &search[begin + k]); });
    for (int m = begin; m < end; ++m) {
      const Rt2DSearch& sr = search[m];
      CMX_REQUIRE(sr.num_scans >= 1 && sr.num_scans < (1 << 16) && sr.nl >= 0 && sr.nl < (1 << 12),
                  "unsupported search window");
      const long long side = 2ll * sr.nl + 1;
      CMX_REQUIRE(side * side * sr.num_scans < (1ll << 30), "search window too large");
    }
  };
  UseDevice(device);
  if (tsdf || Debug().rt2d_legacy) {
    plan_search(0, num);
    Rt2DLegacyBatch(options, items, search.data(), num, device, stats);
    return;
  }
  // A large batch goes out in PARTS, each with its own workspace and stream, all from the calling
  // thread: part k is planned, uploaded and launched while the kernels of the parts before it run,
  // then the parts are collected in order.  (Until round 4 the parts ran on pool threads: a
  // dispatch to the pool costs ~25 us, a cold worker ran its part three times slower than the
  // caller, and eight streams share four hardware queues.)  How many parts, measured on C1's
  // shape at 256 ... 2048 matches (round 4, one box): EQUAL parts of roughly
  // 800 matches, two at least -- the first schedule of the round, small first part then growing
  // by half (128, 192, 320, 384 for 1024 matches), started the device 30 us earlier and then
  // spread a match's rotations over more, smaller work items in every small part: 536 against
  // 450 us at 1024 matches, 899 against 830 at 2048.  Below ~900 matches the parts also SHARE
  // the CUs (each sizes its persistent tile grid and its work items for 1 / parts of them: the
  // parts then run side by side from the start instead of one part's workgroups waiting for
  // the other's to leave their CUs): 185 -> 169 us at 256 matches, 448 -> 368 at 768; above,
  // the second part is still on the host while the first would idle half the chip, and the
  // parts take all CUs as they come.  One part when the caller ordered the work on a stream of
  // its own (cmx_set_stream is per thread).
  std::vector<int> part_end;
  int num_parts = 1;
  bool decreasing = false;
  if (OverrideStream(device) != nullptr) {
    num_parts = 1;
  } else if (Debug().rt2d_parts > 0) {
    num_parts = std::max(1, std::min(std::min(16, Debug().rt2d_parts), num / 32));
  } else if (num >= 256) {
    num_parts = std::max(2, std::min(8, (num + 400) / 800));
    // (round 5, with the bound kernel taking such batches: 1024 matches in three parts 325 us, in
    // two 340, in four 422 -- the first part reaches the device 20 us earlier)
    if (num >= 900 && num_parts == 2) num_parts = 3;
    // (round 6: the kernels of a part now take less than its host work, and a call ends with
    // the whole device latency of its LAST part -- two launches, the stragglers of the tail
    // kernel: 126 us for 341 matches.  Parts of DECREASING size, weights k, k - 1, ... 1: the
    // host's time is the same, the last part a sixth of the call.)
    // Three at most: the streams of a StreamSetLease sit on three hardware queues, a fourth
    // part shares one and waits behind its neighbour (dispatch timeline, profiles/r06b_*).
    decreasing = true;
  }
  if (decreasing) {
    // (and a SMALL first part: with the helper lane the host prepares parts faster than the
    // device takes them, so what counts is how soon the device starts -- the first part an eighth
    // of the call, the rest with weights k - 1, ... 1)
    int first = Debug().rt2d_first_part > 0 ? std::min(Debug().rt2d_first_part, num / 2) : 0;
    if (Debug().rt2d_first_part == 0 && num >= 512 && num_parts >= 3) first = num / 10;
    const int rest_parts = first > 0 ? num_parts - 1 : num_parts;
    if (first > 0) part_end.push_back(first);
    // (weights k + 2, k + 1, ... 3: 4 : 3 for two parts -- neither of 1024 matches' later parts
    // beyond the 512 workgroups of the tail kernel the chip holds at once)
    long long total_weight = 0, weight = 0;
    for (int h = 0; h < rest_parts; ++h) total_weight += rest_parts - h + 2;
    for (int h = 0; h < rest_parts; ++h) {
      weight += rest_parts - h + 2;
      part_end.push_back(first + static_cast<int>(static_cast<long long>(num - first) * weight / total_weight));
    }
  } else {
    for (int h = 0; h < num_parts; ++h)
      part_end.push_back(static_cast<int>(static_cast<long long>(num) * (h + 1) / num_parts));
  }
  const bool share_cus = Debug().rt2d_grid_share ? Debug().rt2d_grid_share == 1
                                                 : (num_parts > 1 && num < 896);
  const int parts = static_cast<int>(part_end.size());
  struct Part {
    int begin, end;
    std::unique_ptr<Rt2DTileCall> call;
    bool eligible = false, enqueued = false;
    cmx_match_stats stats{};
    cmx_status status = CMX_OK;
    std::string error;
  };
  // (the parts must overlap on the device: streams that sit on different hardware queues.
  // Declared BEFORE the parts: a part abandoned by an exception synchronises its stream in its
  // destructor, and that must happen before the streams go back to the pool)
  std::unique_ptr<StreamSetLease> part_streams;
  if (parts > 1) part_streams.reset(new StreamSetLease(device));
  std::vector<Part> part(parts);
  const auto since_call = [&]() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_call).count();
  };
  for (int h = 0; h < parts; ++h) {
    part[h].begin = h ? part_end[h - 1] : 0;
    part[h].end = part_end[h];
  }
  // (round 6, debug switch rt2d_parts_pool: the parts planned and enqueued by host pool threads,
  // one each, instead of one after the other by the caller)
  // A part in two steps: its host-only preparation (search parameters, plan: no device call, no
  // shared lock) and its issue (staging buffer, grid images, upload, launches).  Round 6: the
  // preparation of part h + 1 runs on the helper lane (HostLane) while the calling thread issues
  // part h -- a third of a part's host time off the critical path of every part but the first.
  const auto prepare_part = [&](int h) {
    Part& p = part[h];
    p.status = Guard([&] {
      const double t0 = since_call();
      plan_search(p.begin, p.end);
      const double t_search = since_call();
      p.call.reset(new Rt2DTileCall(options, items + p.begin, search.data() + p.begin, p.end - p.begin, device,
                                    share_cus ? parts : 1, num));
      p.eligible = p.call->Plan();
      if (host_trace)
        fprintf(stderr, "[cmx host] rt2d part %d (%d matches), us since the call's start: begins %.0f, search "
                        "%.0f, plan %.0f\n", h, p.end - p.begin, t0, t_search, since_call());
    });
    if (p.status != CMX_OK) p.error = LastError();
  };
  const auto enqueue_part = [&](int h) {
    Part& p = part[h];
    if (p.status != CMX_OK || !p.eligible) return;
    UseDevice(device);
    p.status = Guard([&] {
      p.call->Enqueue(part_streams ? part_streams->stream(h) : nullptr);
      p.enqueued = true;
      if (host_trace)
        fprintf(stderr, "[cmx host] rt2d part %d enqueued %.0f us since the call's start\n", h, since_call());
    });
    if (p.status != CMX_OK) p.error = LastError();
  };
  if (Debug().rt2d_parts_pool && parts > 1) {
    ParallelFor(parts, 2, [&](int h) { prepare_part(h); enqueue_part(h); });
  } else if (parts > 1 && !Debug().rt2d_no_lane) {
    HostLane lane;
    prepare_part(0);
    for (int h = 0; h < parts; ++h) {
      if (h + 1 < parts) lane.Run([&prepare_part, h] { prepare_part(h + 1); });
      enqueue_part(h);
      lane.Wait();
    }
  } else {
    for (int h = 0; h < parts; ++h) { prepare_part(h); enqueue_part(h); }
  }
  bool failed = false;
  for (int h = 0; h < parts; ++h) failed = failed || part[h].status != CMX_OK;
  for (int h = 0; h < parts; ++h) {
    Part& p = part[h];
    if (p.status != CMX_OK) { p.call.reset(); continue; }    // (waits for what it had in flight)
    if (failed) { p.call.reset(); continue; }                // (the call fails: no reruns)
    p.status = Guard([&] {
      std::vector<int> redo;
      const bool done = p.enqueued && p.call->Collect(&p.stats, &redo);
      // (not eligible: the part runs on the per-candidate kernels; so do the matches of it whose
      // score landscape was too flat for the lists, or that had a point outside the predicted box)
      if (!done) {
        Rt2DLegacyBatch(options, items + p.begin, search.data() + p.begin, p.end - p.begin, device,
                        &p.stats);
      } else if (!redo.empty()) {
        std::vector<Rt2DItem> again_items;
        std::vector<Rt2DSearch> again_search;
        for (int m : redo) {
          again_items.push_back(items[p.begin + m]);
          again_search.push_back(search[p.begin + m]);
        }
        cmx_match_stats again{};
        Rt2DLegacyBatch(options, again_items.data(), again_search.data(), static_cast<int>(redo.size()), device,
                        &again);
        p.stats.candidates_scored += again.candidates_scored;
        p.stats.coarse_candidates += again.coarse_candidates;
        p.stats.num_scans += again.num_scans;
        p.stats.refined_candidates += again.refined_candidates;
        p.stats.finalists += again.finalists;
      }
      if (host_trace)
        fprintf(stderr, "[cmx host] rt2d part %d collected %.0f us since the call's start\n", h, since_call());
    });
    if (p.status != CMX_OK) p.error = LastError();
    p.call.reset();
  }
  lap("parts");
  cmx_match_stats total{};
  for (int h = 0; h < parts; ++h) {
    const Part& p = part[h];
    if (p.status != CMX_OK) {
      SetLastError("%s", p.error.c_str());
      throw HipError{p.status};
    }
    total.candidates_scored += p.stats.candidates_scored;
    total.coarse_candidates += p.stats.coarse_candidates;
    total.num_scans += p.stats.num_scans;
    total.device_ms = std::max(total.device_ms, p.stats.device_ms);      // the parts overlap
    total.dominant_kernel_ms += p.stats.dominant_kernel_ms;
    total.refined_candidates += p.stats.refined_candidates;
    total.finalists += p.stats.finalists;
  }
  if (stats) *stats = total;
  if (host_trace)
    fprintf(stderr, "[cmx host] rt2d batch(%d, %d parts):%s us\n", num, parts, host_report.c_str());
}

void Rt2DMatch(const cmx_rt_options* options, const cmx_grid2d_limits* limits,
               const uint16_t* cells, const uint16_t* weight_cells, float max_tsd,
               float max_weight, const cmx_pose2d* initial_pose_estimate,
               const float* point_cloud_xyz, int32_t num_points, int32_t device, double* score,
               cmx_pose2d* pose_estimate, cmx_match_stats* stats,
               const uint16_t* device_cells) {
  Rt2DItem item{};
  item.limits = limits; item.cells = cells; item.weight_cells = weight_cells;
  item.max_tsd = max_tsd; item.max_weight = max_weight; item.device_cells = device_cells;
  item.initial = initial_pose_estimate; item.xyz = point_cloud_xyz; item.n = num_points;
  item.score = score; item.pose = pose_estimate;
  CMX_REQUIRE(options != nullptr, "null argument");
  Rt2DMatchBatch(options, &item, 1, device, stats);
}

}  // namespace cmx

extern "C" cmx_status cmx_rt2d_match(const cmx_rt_options* options,
                                     const cmx_grid2d_limits* limits, const uint16_t* cells,
                                     const cmx_pose2d* initial_pose_estimate,
                                     const float* point_cloud_xyz, int32_t num_points,
                                     int32_t device, double* score, cmx_pose2d* pose_estimate,
                                     cmx_match_stats* stats) {
  return cmx::Guard([&] {
    cmx::Rt2DMatch(options, limits, cells, nullptr, 0.f, 0.f, initial_pose_estimate,
                   point_cloud_xyz, num_points, device, score, pose_estimate, stats, nullptr);
  });
}

extern "C" cmx_status cmx_rt2d_match_tsdf(const cmx_rt_options* options,
                                          const cmx_grid2d_limits* limits,
                                          const uint16_t* tsd_cells,
                                          const uint16_t* weight_cells,
                                          float truncation_distance, float max_weight,
                                          const cmx_pose2d* initial_pose_estimate,
                                          const float* point_cloud_xyz, int32_t num_points,
                                          int32_t device, double* score,
                                          cmx_pose2d* pose_estimate, cmx_match_stats* stats) {
  return cmx::Guard([&] {
    CMX_REQUIRE(weight_cells != nullptr, "null argument");
    cmx::Rt2DMatch(options, limits, tsd_cells, weight_cells, truncation_distance, max_weight,
                   initial_pose_estimate, point_cloud_xyz, num_points, device, score,
                   pose_estimate, stats, nullptr);
  });
}

extern "C" cmx_status cmx_rt2d_score_candidates(
    const cmx_rt_options* options, const cmx_grid2d_limits* limits, const uint16_t* cells,
    const uint16_t* weight_cells, float truncation_distance, float max_weight,
    const int32_t* discrete_scans_xy, const int32_t* scan_begin, int32_t num_scans,
    cmx_candidate2d* candidates, int32_t num_candidates, int32_t device) {
  using namespace cmx;
  return Guard([&] {
    CMX_REQUIRE(options && limits && cells && scan_begin && num_scans >= 1, "null argument");
    CMX_REQUIRE(num_candidates >= 0 && (num_candidates == 0 || candidates), "bad candidate list");
    CMX_REQUIRE(limits->num_x_cells >= 1 && limits->num_y_cells >= 1 &&
                    static_cast<long long>(limits->num_x_cells) * limits->num_y_cells < (1ll << 30),
                "bad cell limits");
    const bool tsdf = weight_cells != nullptr;
    if (tsdf) CMX_REQUIRE(truncation_distance > 0.f && max_weight > 0.f, "bad TSDF ranges");
    CMX_REQUIRE(scan_begin[0] == 0, "scan_begin[0] must be 0");
    for (int s = 0; s < num_scans; ++s)       // (an empty scan divides 0 by 0: CHECK_GT at :70)
      CMX_REQUIRE(scan_begin[s + 1] > scan_begin[s], "discrete scan %d is empty", s);
    const int total_points = scan_begin[num_scans];
    CMX_REQUIRE(discrete_scans_xy != nullptr, "null argument");
    for (int c = 0; c < num_candidates; ++c)
      CMX_REQUIRE(candidates[c].scan_index >= 0 && candidates[c].scan_index < num_scans,
                  "candidate %d names scan %d of %d", c, candidates[c].scan_index, num_scans);
    if (num_candidates == 0) return;
    WorkspaceLease ws(device);
    const size_t num_cells = static_cast<size_t>(limits->num_x_cells) * limits->num_y_cells;
    uint16_t* d_cells = ws->dev[0].ReserveAs<uint16_t>(num_cells * (tsdf ? 2 : 1));
    int* d_scans = ws->dev[1].ReserveAs<int>(2 * static_cast<size_t>(total_points) + num_scans + 1);
    int4* d_cand = ws->dev[2].ReserveAs<int4>(num_candidates);
    float* d_scores = ws->dev[3].ReserveAs<float>(num_candidates);
    std::vector<int4> h_cand(num_candidates);
    for (int c = 0; c < num_candidates; ++c)
      h_cand[c] = make_int4(candidates[c].scan_index, candidates[c].x_index_offset,
                            candidates[c].y_index_offset, 0);
    std::vector<float> h_scores(num_candidates);
    CMX_HIP(hipMemcpyAsync(d_cells, cells, num_cells * sizeof(uint16_t), hipMemcpyHostToDevice,
                           ws->stream));
    if (tsdf)
      CMX_HIP(hipMemcpyAsync(d_cells + num_cells, weight_cells, num_cells * sizeof(uint16_t),
                             hipMemcpyHostToDevice, ws->stream));
    CMX_HIP(hipMemcpyAsync(d_scans, discrete_scans_xy, 2 * sizeof(int) * total_points,
                           hipMemcpyHostToDevice, ws->stream));
    int* d_begin = d_scans + 2 * static_cast<size_t>(total_points);
    CMX_HIP(hipMemcpyAsync(d_begin, scan_begin, sizeof(int) * (num_scans + 1),
                           hipMemcpyHostToDevice, ws->stream));
    CMX_HIP(hipMemcpyAsync(d_cand, h_cand.data(), sizeof(int4) * num_candidates,
                           hipMemcpyHostToDevice, ws->stream));
    const int blocks = DivUp(num_candidates, 64);
    if (tsdf)
      Rt2DScoreCandidatesKernel<true><<<blocks, 64, 0, ws->stream>>>(
          d_cells, d_cells + num_cells, limits->num_x_cells, limits->num_y_cells,
          truncation_distance, max_weight, d_scans, d_begin, d_cand, num_candidates, d_scores);
    else
      Rt2DScoreCandidatesKernel<false><<<blocks, 64, 0, ws->stream>>>(
          d_cells, nullptr, limits->num_x_cells, limits->num_y_cells, 0.f, 0.f, d_scans, d_begin,
          d_cand, num_candidates, d_scores);
    CMX_HIP(hipGetLastError());
    CMX_HIP(hipMemcpyAsync(h_scores.data(), d_scores, sizeof(float) * num_candidates,
                           hipMemcpyDeviceToHost, ws->stream));
    CMX_HIP(hipStreamSynchronize(ws->stream));
    for (int c = 0; c < num_candidates; ++c) {
      cmx_candidate2d& cand = candidates[c];
      float score = h_scores[c];
      if (!tsdf) CMX_REQUIRE(score > 0.f, "candidate %d scores %g (CHECK_GT(score, 0))", c, score);
      // `candidate.score *= std::exp(-Pow2(...))`: a float times a double, rounded once (:168-174)
      const double t = std::hypot(cand.x, cand.y) * options->translation_delta_cost_weight +
                       std::abs(cand.orientation) * options->rotation_delta_cost_weight;
      score = static_cast<float>(static_cast<double>(score) * std::exp(-(t * t)));
      cand.score = score;
    }
  });
}
