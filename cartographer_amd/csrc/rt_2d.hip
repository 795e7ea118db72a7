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
  // LDS-staged integer bulk pass (Rt2DBulkKernel / Rt2DExactKernel, see below)
  int wp, hp;                // staged grid: cells per row (16 x odd), rows
  int hl, ht;                // left / top halo: LDS (x, y) = grid (x + hl, y + ht)
  int blocks_per_row;        // aligned 4-cell blocks covering a window row at any phase
  int rounds_rot;            // rotations prepared and scored together by a workgroup
  int list_cap;              // per-rotation capacity of the phase-sorted address list
  int task_cap;              // per-rotation task descriptor slots (>= chunks of a rotation)
  int xyz_in_lds;            // the cloud is copied to LDS once per workgroup
  int* qsum;                 // [num_scans][side * side] integer sums of quantised cells
  float* ub;                 // [num_scans][side * side] weighted upper bounds (row-pair kernel)
  // Row-pair bulk pass (Rt2DImageKernel / Rt2DRowPairKernel, see below)
  uint16_t* qimage;          // device: the staged grid as the workgroups copy it (zero halo baked in)
  int pitch;                 // bytes per image row (multiple of 8; rows of a half-wave tile the banks)
  int image_bytes;           // hp * pitch, padded to whole KiB
  int half_rows;             // H: window rows r, r + H, ... belong to one lane
  int rows_per_lane;         // ceil(side / H), 1 .. kMaxRowsPerLane
  int pair_list_cap;         // per-rotation capacity of the phase-sorted u16 entry list
  int image_build;           // 0: qimage is a cached image of a resident grid, already built
  unsigned long long* timeline;   // CMX_TIMELINE=1: 16 stamps per bulk / exact block, else null
  int timeline_exact_base;        // first block slot of the exact kernel
};

// ProbabilityGrid::GetProbability (mapping/2d/probability_grid.cc:78-82) with
// kValueToCorrespondenceCost (mapping/probability_values.cc:33-41,65-74)
// evaluated arithmetically.
__device__ __forceinline__ float CellProbability(unsigned raw) {
  const float kMinP = 0.1f;
  const float kMaxP = 1.f - kMinP;
  const float kMinCC = 1.f - kMaxP;
  const float kMaxCC = 1.f - kMinP;
  const unsigned v = raw & 32767u;
  float cost;
  if (v == 0) {
    cost = kMaxCC;
  } else {
    const float scale = (kMaxCC - kMinCC) / (32768 - 2.f);
    cost = static_cast<float>(v) * scale + (kMinCC - scale);
  }
  return 1.f - cost;
}

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
// unweighted score: (index, score bits) pairs -- the first kFinalistHead next to the
// counters (they travel back with them), the rest in the overflow region.
constexpr int kFinalistCap = 4096;
constexpr int kFinalistHead = 62;   // 2 + 2 * 62 words = 512 bytes per match

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

// ---------------------------------------------------------------------------
// Probability grids that fit in LDS: integer bulk pass + exact finalists
// ---------------------------------------------------------------------------
// The reference's score of a candidate is mean_p P(cell_p + d) summed in f32 in point order
// (:61-75), and P is affine in the stored uint16: P = 0.1 + u * kScale with
// u = 32767 - value (0 for unknown / outside; real arithmetic, the f32 table rounds each
// entry by < 1e-7).  Integer sums of u are exact and order-free, so the bulk of the search
// needs neither the f32 chain nor one gather per (candidate, point):
//   * the grid is staged ONCE per workgroup in LDS as 16-bit fields holding
//     q = u >> kQShift (10 bits), rows padded with a zero halo so that no lookup needs a
//     bounds test;
//   * all (2 nl + 1)^2 candidates of a rotation read, for one point, a (2 nl + 1)^2 window
//     of cells: a lane owns an aligned 4-cell block of one window row, fetches it with ONE
//     ds_read_b64 and adds it to two packed-16-bit registers with two v_pk_add_u16 --
//     four candidates per LDS read, 2 VALU instructions per 4 lookups;
//   * the block is 8-byte aligned in LDS, the window is not: points are sorted by the phase
//     (window start mod 4) and a lane's four sums belong to candidates 4 b + j - phase;
//   * 64 points are added before the 16-bit sums are flushed to 32-bit LDS accumulators
//     (64 * 1023 < 65536).
// This yields, per candidate, Q with sum(u) in [2^kQShift Q, 2^kQShift Q + (2^kQShift - 1) N], i.e.
// a score interval of width 31 kScale = 7.6e-4.  Every candidate whose weighted upper
// bound reaches the best weighted lower bound (with 1e-4 of slack for the rounding of the
// f32 chain) is a finalist; Rt2DExactKernel recomputes those -- a handful -- with the
// reference's sequential f32 sum, and the host applies the libm weight and the
// first-maximum rule to them exactly as before.  Returned score and pose are bit-identical
// to the one-thread-per-candidate kernels above; those remain the path for TSDFs and for
// grids that do not fit in LDS.
constexpr int kQShift = 5;
constexpr int kQChunk = 64;                 // points per packed accumulation
constexpr int kBulkThreads = 1024;
constexpr int kBulkWaves = kBulkThreads / 64;
constexpr int kMaxRoundRot = 4;
constexpr int kBulkMaxPoints = 8192;
constexpr double kBoundSlack = 1e-4;        // f32-chain rounding (N * 2^-24 * sum / N, generous)

// The reference's discretisation of point i of rotation (q0, qs): cell (ix, iy), clamped to
// one cell further outside the grid than any offset can reach back in (same as
// Rt2DPrepKernel above).
__device__ __forceinline__ void Rt2DPointCell(const Rt2DParams& P, const Quat& q0, const Quat& qs,
                                              const F3& p, int* ix, int* iy) {
  // Rt2DPrepKernel's two yaw rotations and translation without the exactly-zero terms
  // (RotateZ, cmx_device.h: bit-identical x / y for finite coordinates).
  float ax, ay, bx, by;
  RotateZ(q0.w, q0.z, p.x, p.y, &ax, &ay);
  RotateZ(qs.w, qs.z, ax, ay, &bx, &by);
  const float x = bx + P.tx;
  const float y = by + P.ty;
  // lround((max - v) / res - 0.5) from an f32 estimate when provably equal (cmx_device.h).
  const int cx = CellIndexFast(P.max_y, y, P.res, P.inv_res);
  const int cy = CellIndexFast(P.max_x, x, P.res, P.inv_res);
  *ix = min(max(cx, -(P.nl + 1)), P.nx + P.nl);
  *iy = min(max(cy, -(P.nl + 1)), P.ny + P.nl);
}

// exp(-(hypot(x, y) w_t + |theta| w_r)^2) in f32 (relative error ~1e-6): only used for the
// bounds below, which carry 1e-5 of relative slack on top; the returned score is weighted on
// the host with libm.
__device__ __forceinline__ float Rt2DWeight(const Rt2DParams& P, int s, int dx, int dy) {
  const float res = static_cast<float>(P.res);
  const float cx = -dy * res, cy = -dx * res;
  const float theta = static_cast<float>((s - P.num_angular) * P.step);
  const float t = sqrtf(cx * cx + cy * cy) * static_cast<float>(P.wt) +
                  fabsf(theta) * static_cast<float>(P.wr);
  return __expf(-(t * t));
}

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

// Points 16 J + K0 .. 16 J + K0 + 3 of a chunk.  `addrs` holds, in every row of 16 lanes, the
// block addresses of points 16 J .. 16 J + 15: `v_add_u32_dpp ... row_newbcast:k` adds lane k
// of the row to the lane's own (row, block) offset in ONE instruction (a v_readlane to an
// SGPR, its hazard nop and the add were three issue slots per point).
template <int K>
__device__ __forceinline__ int RowBroadcastAdd(int addrs, int lane_off) {
  int out;   // (asm: the compiler splits the intrinsic form into v_mov_b32_dpp + v_add3_u32)
  asm("v_add_u32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
      : "=v"(out) : "v"(addrs), "v"(lane_off), "i"(K));
  return out;
}
// (`lane_off` already contains the LDS address of the staged grid: the sum is an absolute LDS
// address, read through an address_space(3) pointer -- `smem + offset` costs an extra v_add of
// the array's base per read.)
typedef unsigned uint2v __attribute__((ext_vector_type(2)));
typedef const uint2v __attribute__((address_space(3)))* LdsUint2Ptr;
template <int K0>
__device__ __forceinline__ void Load4(int addrs, int lane_off, uint2 (&v)[4]) {
  { const uint2v t = *reinterpret_cast<LdsUint2Ptr>(static_cast<uintptr_t>(RowBroadcastAdd<K0>(addrs, lane_off))); v[0] = make_uint2(t[0], t[1]); }
  { const uint2v t = *reinterpret_cast<LdsUint2Ptr>(static_cast<uintptr_t>(RowBroadcastAdd<K0 + 1>(addrs, lane_off))); v[1] = make_uint2(t[0], t[1]); }
  { const uint2v t = *reinterpret_cast<LdsUint2Ptr>(static_cast<uintptr_t>(RowBroadcastAdd<K0 + 2>(addrs, lane_off))); v[2] = make_uint2(t[0], t[1]); }
  { const uint2v t = *reinterpret_cast<LdsUint2Ptr>(static_cast<uintptr_t>(RowBroadcastAdd<K0 + 3>(addrs, lane_off))); v[3] = make_uint2(t[0], t[1]); }
}

// Two points per instruction: a 64-point chunk never carries out of a 16-bit field
// (64 * 1023 < 65536), so the packed sums are plain 32-bit additions and v_add3_u32 adds two
// points' cells at once.  (asm: as C++ integer adds LLVM reassociates the 128 additions of a
// chunk into a tree evaluated after all 64 loads -- 128 live VGPRs and spills in this loop.)
__device__ __forceinline__ void Add4(const uint2 (&v)[4], uint32_t* lo, uint32_t* hi) {
  asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(*lo) : "v"(v[0].x), "v"(v[1].x));
  asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(*hi) : "v"(v[0].y), "v"(v[1].y));
  asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(*lo) : "v"(v[2].x), "v"(v[3].x));
  asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(*hi) : "v"(v[2].y), "v"(v[3].y));
}

// Inclusive prefix sum across the 64 lanes (DPP ladder of WaveSum without the broadcast).
__device__ __forceinline__ int WaveInclusiveScan(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);   // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);   // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);   // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);   // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, true);   // row_bcast:15
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, true);   // row_bcast:31
  return v;
}

// grid (workgroups per match, matches); a workgroup stages its match's grid and then takes
// the rotations blockIdx.x, blockIdx.x + gridDim.x, ... `rounds_rot` at a time.
// Dynamic LDS: grid[hp][wp] u16 | tmp[R][n_pad] | list[R][list_cap] | counts[R][chunks][4] |
// acc[R][side^2] | desc[R][task_cap] | ctl[64] | xyz[3 n] (when it fits).
__global__ void __launch_bounds__(kBulkThreads)
Rt2DBulkKernel(const Rt2DParams* __restrict__ params) {
  extern __shared__ __attribute__((aligned(16))) unsigned char bulk_smem[];
  const Rt2DParams& P = params[blockIdx.y];
  if (static_cast<int>(blockIdx.x) >= P.num_scans) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = P.n, n_pad = P.n_pad, pchunks = n_pad >> 6;
  const int side = 2 * P.nl + 1, cands = side * side;
  const int R = P.rounds_rot, B = P.blocks_per_row;
  const int wp = P.wp, hp = P.hp;
  uint16_t* grid = reinterpret_cast<uint16_t*>(bulk_smem);
  int* tmp = reinterpret_cast<int*>(bulk_smem + static_cast<size_t>(wp) * hp * 2);
  int* list = tmp + R * n_pad;
  int* counts = list + R * P.list_cap;      // [R][pchunks][4]
  int* acc = counts + R * pchunks * 4;      // [R][cands]
  int* desc = acc + R * cands;              // [R][tcap] task descriptors
  int* ctl = desc + R * P.task_cap;         // [8 rr + ph]: first chunk of phase ph (4: #chunks)
  float* xyz_lds = reinterpret_cast<float*>(ctl + 64);     // x[n_pad] y[n_pad] z[n_pad] when P.xyz_in_lds
  const int tcap = P.task_cap;

  unsigned long long* const tl = P.timeline;
  const int tl_block = blockIdx.y * gridDim.x + blockIdx.x;
  Stamp(tl, tl_block, 0);
  // ---- stage the grid: zero (halo included), then quantise the cells ------------------
  {
    uint4* g4 = reinterpret_cast<uint4*>(grid);
    const int vecs = (wp * hp) >> 3;
    for (int i = tid; i < vecs; i += kBulkThreads) g4[i] = make_uint4(0u, 0u, 0u, 0u);
  }
  __syncthreads();
  {
    // Four independent loads in flight per thread before the first is converted (one load
    // per loop iteration made this the longest phase of a single match: a chain of L2 round
    // trips).  Rows of a multiple of 8 cells are read 16 bytes (8 cells) at a time.
    const auto quantise = [](unsigned v) -> unsigned {
      v &= 32767u;
      return v ? (32767u - v) >> kQShift : 0u;
    };
    if ((P.nx & 7) == 0 && (reinterpret_cast<uintptr_t>(P.cells) & 15) == 0) {
      typedef unsigned uint4v __attribute__((ext_vector_type(4)));
      const auto* cells8 = AsGlobal(reinterpret_cast<const uint4v*>(P.cells));
      const int total8 = (P.nx * P.ny) >> 3, row8 = P.nx >> 3;
      for (int base = tid; base < total8; base += 4 * kBulkThreads) {
        uint4v v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int e = base + k * kBulkThreads;
          v[k] = cells8[min(e, total8 - 1)];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int e = base + k * kBulkThreads;
          if (e >= total8) break;
          const int y = e / row8, x = (e - y * row8) << 3;
          const unsigned w[4] = {v[k][0], v[k][1], v[k][2], v[k][3]};
          unsigned q[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) q[j] = quantise(w[j]) | (quantise(w[j] >> 16) << 16);
          // (hl is a multiple of 4 cells: 8-byte aligned destination)
          uint2* dst = reinterpret_cast<uint2*>(grid + (y + P.ht) * wp + (x + P.hl));
          dst[0] = make_uint2(q[0], q[1]);
          dst[1] = make_uint2(q[2], q[3]);
        }
      }
    } else {
      const auto* cells = AsGlobal(P.cells);
      const int total = P.nx * P.ny;
      for (int base = tid; base < total; base += 8 * kBulkThreads) {
        unsigned v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = cells[min(base + k * kBulkThreads, total - 1)];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int e = base + k * kBulkThreads;
          if (e >= total) break;
          const int y = e / P.nx, x = e - y * P.nx;
          grid[(y + P.ht) * wp + (x + P.hl)] = static_cast<uint16_t>(quantise(v[k]));
        }
      }
    }
  }
  // The cloud is rotated once per rotation of this workgroup: keep it on chip (a global
  // load per rotation and point was a chain of L2 round trips: 1.2 us per rotation).
  // (as three arrays: an xyz triple per lane is a 12-byte-strided ds_read_b96, off its natural
  // alignment three times out of four and replayed at 64 cycles each.)
  if (P.xyz_in_lds) {
    for (int i = tid; i < 3 * n; i += kBulkThreads) {
      const int pt = i / 3, c = i - 3 * pt;
      xyz_lds[c * n_pad + pt] = P.xyz[i];
    }
  }
  Stamp(tl, tl_block, 1);      // grid staged (this wave)
  const Quat q0{P.init_qw, 0.f, 0.f, P.init_qz};
  const int slices = (side * B + 63) >> 6;
  const float kScale = ((1.f - 0.1f) - (1.f - (1.f - 0.1f))) / 32766.f;   // (kMaxCC - kMinCC) / 32766

  for (int s0 = blockIdx.x; s0 < P.num_scans; s0 += gridDim.x * R) {
    // Rotations of this round: s0, s0 + gridDim.x, ... (at most R).
    const int round_rot = min(R, (P.num_scans - s0 + static_cast<int>(gridDim.x) - 1) /
                                     static_cast<int>(gridDim.x));
    __syncthreads();                       // previous round's accumulators have been read
    if (s0 == static_cast<int>(blockIdx.x)) Stamp(tl, tl_block, 2);      // grid staged (all waves)
    for (int i = tid; i < round_rot * cands; i += kBulkThreads) acc[i] = 0;
    for (int i = tid; i < round_rot * P.list_cap; i += kBulkThreads) list[i] = 0;   // null block
    for (int i = tid; i < round_rot * tcap; i += kBulkThreads) desc[i] = -1;
    // ---- discretise: one wavefront per 64 points of one rotation -----------------------
    for (int vw = wave; vw < round_rot * pchunks; vw += kBulkWaves) {
      const int rr = vw / pchunks, pc = vw - rr * pchunks;
      const int s = s0 + rr * gridDim.x;
      const float2 r = P.scan_rot[s];
      const Quat qs{r.x, 0.f, 0.f, r.y};
      const int i = pc * 64 + lane;
      int packed = -1;
      if (i < n) {
        int ix, iy;
        const F3 p = P.xyz_in_lds
                         ? F3{xyz_lds[i], xyz_lds[n_pad + i], xyz_lds[2 * n_pad + i]}
                         : F3{P.xyz[3 * i], P.xyz[3 * i + 1], P.xyz[3 * i + 2]};
        Rt2DPointCell(P, q0, qs, p, &ix, &iy);
        const int wx = ix - P.nl + P.hl, wy = iy - P.nl + P.ht;   // window start, LDS coordinates
        packed = (((wy * wp + (wx & ~3)) * 2) << 2) | (wx & 3);
      }
      tmp[rr * n_pad + i] = packed;
#pragma unroll
      for (int ph = 0; ph < 4; ++ph) {
        const int c = __popcll(__ballot(packed >= 0 && (packed & 3) == ph));
        if (lane == 0) counts[(rr * pchunks + pc) * 4 + ph] = c;
      }
    }
    __syncthreads();
    if (s0 == static_cast<int>(blockIdx.x)) Stamp(tl, tl_block, 3);      // points discretised
    // ---- per (rotation, phase): exclusive offsets, phases padded to whole chunks; one
    // wavefront per rotation, lanes = 64-point chunks, DPP prefix sums ----------------------
    if (wave < round_rot) {
      const int rr = wave;
      int start_chunk = 0;                  // first chunk of the current phase
#pragma unroll
      for (int ph = 0; ph < 4; ++ph) {
        int carry = 0;                      // points of this phase in earlier 64-chunk blocks
        for (int p0 = 0; p0 < pchunks; p0 += 64) {
          const int pc = p0 + lane;
          const int c = pc < pchunks ? counts[(rr * pchunks + pc) * 4 + ph] : 0;
          const int incl = WaveInclusiveScan(c);
          if (pc < pchunks)
            counts[(rr * pchunks + pc) * 4 + ph] = start_chunk * kQChunk + carry + incl - c;
          carry += __builtin_amdgcn_readlane(incl, 63);
        }
        if (lane == 0) ctl[8 * rr + ph] = start_chunk;
        const int chunks = (carry + kQChunk - 1) / kQChunk;
        // Task descriptors of this phase's chunks: rr | phase << 4 | chunk << 8.
        for (int c = lane; c < chunks; c += 64)
          desc[rr * tcap + start_chunk + c] = rr | (ph << 4) | ((start_chunk + c) << 8);
        start_chunk += chunks;
      }
      if (lane == 0) ctl[8 * rr + 4] = start_chunk;
    }
    __syncthreads();
    for (int vw = wave; vw < round_rot * pchunks; vw += kBulkWaves) {
      const int rr = vw / pchunks, pc = vw - rr * pchunks;
      const int i = pc * 64 + lane;
      const int packed = tmp[rr * n_pad + i];
      const int ph = packed & 3;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const unsigned long long m = __ballot(packed >= 0 && ph == q);
        if (packed >= 0 && ph == q) {
          const int pos = counts[(rr * pchunks + pc) * 4 + q] +
                          __popcll(m & ((1ull << lane) - 1ull));
          list[rr * P.list_cap + pos] = packed >> 2;
        }
      }
    }
    __syncthreads();
    if (s0 == static_cast<int>(blockIdx.x)) Stamp(tl, tl_block, 4);      // lists sorted by phase
    // ---- tasks: (rotation, 64-point chunk, lane slice), dealt round-robin to the waves ----
    int first_task[kMaxRoundRot + 1];        // dense numbering over the round's rotations
    first_task[0] = 0;
#pragma unroll
    for (int rr = 0; rr < kMaxRoundRot; ++rr)
      first_task[rr + 1] = first_task[rr] + (rr < round_rot ? ctl[8 * rr + 4] * slices : 0);
    for (int t = wave; t < first_task[kMaxRoundRot]; t += kBulkWaves) {
      int rr = 0;
#pragma unroll
      for (int q = 1; q < kMaxRoundRot; ++q) rr += t >= first_task[q] ? 1 : 0;
      const int local = t - first_task[rr];
      const int chunk = local / slices, slice = local - chunk * slices;
      const int phase = (desc[rr * tcap + chunk] >> 4) & 15;
      const int item = slice * 64 + lane;
      const bool valid = item < side * B;
      const int row = valid ? item / B : 0, blk = valid ? item - row * B : 0;
      const int lane_off =
          (row * wp + blk * 4) * 2 +
          static_cast<int>(reinterpret_cast<uintptr_t>(
              (const __attribute__((address_space(3))) unsigned char*)bulk_smem));
      const int* my_list = list + rr * P.list_cap + chunk * kQChunk + (lane & 15);
      const int a0 = my_list[0], a1 = my_list[16], a2 = my_list[32], a3 = my_list[48];
      uint32_t lo = 0, hi = 0;      // packed 16-bit sums: cells (0 | 1 << 16), (2 | 3 << 16)
      // Three banks of 4 LDS reads: groups g+1 and g+2 are in flight while group g is added
      // (12 outstanding: lgkmcnt counts to 15; 16 waves per CU keep the LDS pipe busy).  The
      // scheduling barriers keep the compiler from hoisting all 64 reads.
      uint2 v0[4], v1[4], v2[4];
      Load4<0>(a0, lane_off, v0);
      Load4<4>(a0, lane_off, v1);
      __builtin_amdgcn_sched_barrier(0);
#define CMX_RT2D_STEP(ADDRS, K0, LOAD_BANK, ADD_BANK)              \
      Load4<K0>(ADDRS, lane_off, LOAD_BANK);             \
      Add4(ADD_BANK, &lo, &hi);                                     \
      __builtin_amdgcn_sched_barrier(0);
      CMX_RT2D_STEP(a0, 8, v2, v0)
      CMX_RT2D_STEP(a0, 12, v0, v1)
      CMX_RT2D_STEP(a1, 0, v1, v2)
      CMX_RT2D_STEP(a1, 4, v2, v0)
      CMX_RT2D_STEP(a1, 8, v0, v1)
      CMX_RT2D_STEP(a1, 12, v1, v2)
      CMX_RT2D_STEP(a2, 0, v2, v0)
      CMX_RT2D_STEP(a2, 4, v0, v1)
      CMX_RT2D_STEP(a2, 8, v1, v2)
      CMX_RT2D_STEP(a2, 12, v2, v0)
      CMX_RT2D_STEP(a3, 0, v0, v1)
      CMX_RT2D_STEP(a3, 4, v1, v2)
      CMX_RT2D_STEP(a3, 8, v2, v0)
      CMX_RT2D_STEP(a3, 12, v0, v1)
#undef CMX_RT2D_STEP
      Add4(v2, &lo, &hi);
      Add4(v0, &lo, &hi);
      if (valid) {
        int* out = acc + rr * cands;
        const int d0 = blk * 4 - phase;          // candidate x index of the block's first cell
        const int sums[4] = {static_cast<int>(lo & 0xffffu), static_cast<int>(lo >> 16),
                             static_cast<int>(hi & 0xffffu), static_cast<int>(hi >> 16)};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int dxi = d0 + j;
          if (dxi >= 0 && dxi < side && sums[j]) atomicAdd(&out[dxi * side + row], sums[j]);
        }
      }
    }
    if (s0 == static_cast<int>(blockIdx.x)) Stamp(tl, tl_block, 5);      // wave 0 out of tasks
    __syncthreads();
    if (s0 == static_cast<int>(blockIdx.x)) Stamp(tl, tl_block, 6);      // all tasks done
    // ---- per candidate: integer sum out, weighted lower bound into the match's maximum --
    float lb_max = 0.f;
    for (int e = tid; e < round_rot * cands; e += kBulkThreads) {
      const int rr = e / cands, c = e - rr * cands;
      const int s = s0 + rr * gridDim.x;
      const int q = acc[e];
      P.qsum[static_cast<size_t>(s) * cands + c] = q;
      const int dxi = c / side, dyi = c - dxi * side;
      // (the integer is < 2^25 x 2^5: exact in f64; one f32 rounding at the end, downwards
      // by the 1e-5 factor)
      const double lo_score =
          0.1 + static_cast<double>(kScale) * (static_cast<double>(q) * (1 << kQShift)) / n;
      const float lb = static_cast<float>(lo_score - kBoundSlack) *
                       Rt2DWeight(P, s, dxi - P.nl, dyi - P.nl) * (1.f - 1e-5f);
      lb_max = fmaxf(lb_max, lb);
    }
    unsigned bits = __float_as_uint(fmaxf(lb_max, 0.f));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) bits = max(bits, __shfl_xor(bits, off, 64));
    if (lane == 0 && bits) atomicMax(&P.misc[0], bits);
    if (s0 == static_cast<int>(blockIdx.x)) Stamp(tl, tl_block, 7);      // first round finished
  }
  Stamp(tl, tl_block, 8);
}

// ---------------------------------------------------------------------------
// Row-pair bulk pass (round 3): the same integer sums, organised around the LDS
// ---------------------------------------------------------------------------
// What bounded Rt2DBulkKernel (profiles/r02e_c1_pmc_sq_*): 58 % of the VALU issue slots and 37 %
// of the LDS cycles -- half of the vector instructions were NOT the window update (every
// workgroup re-quantised the whole grid: 19 k instructions; two passes over the rotated points;
// f64 bound arithmetic), every ds_read_b64 cost an address instruction, 12 of 64 lanes idled,
// phases were padded to 64-point chunks, and 27 % of the LDS cycles were bank conflicts.
// This kernel keeps the idea (aligned 4-cell blocks of 16-bit q, packed adds, phases) and
// changes the mapping:
//   * the staged grid is built ONCE per grid as a ready-to-copy image in HBM (quantised,
//     zero halo, skewed pitch: Rt2DImageKernel; cached with a resident cmx_grid2d) and enters
//     LDS by LDS-DMA while the points are being discretised;
//   * a HALF-wavefront (32 lanes) is one stream of points of one phase: lane = (row r < H,
//     block b < B), H * B <= 32, and a lane owns the window rows r, r + H, ... (two for the
//     13 x 13 window of C1: H = 8, B = 4): ONE address instruction (v_add_u32_dpp
//     row_newbcast) serves rows_per_lane reads through the instruction's immediate offset;
//   * the row pitch is chosen on the host so that the H x B 8-byte blocks a half-wavefront
//     reads in one cycle fall into distinct banks for every base address: the reads are
//     conflict-free by construction;
//   * the two halves of a wavefront run two phases side by side, phases are paired by size,
//     lists are padded to 16 entries only, tasks are at most 64 iterations and dealt
//     dynamically;
//   * points are pre-rotated by the initial yaw once per workgroup and discretised ONCE per
//     rotation (the packed entries wait in registers for the list offsets).
// For C1: 1.5 vector instructions and 1 ds_read_b64 wave-instruction per point and
// half-wavefront, against 2 + 1 per point and wavefront before, with all 32 lanes of a half busy.
constexpr int kMaxRowsPerLane = 4;
constexpr int kExactSplit = 4;              // workgroups sharing a match's rotations with finalists
constexpr int kPairTaskIters = 64;          // iterations (entries per stream) of one task
constexpr int kPairChunksPerWave = 16;      // (rotation, 64-point chunk) pairs a wave discretises per round

// grid (ceil(image_bytes / 16 / 256), matches): the quantised image, 8 cells per thread.
__global__ void __launch_bounds__(256)
Rt2DImageKernel(const Rt2DParams* __restrict__ params) {
  const Rt2DParams& P = params[blockIdx.y];
  const int v = blockIdx.x * 256 + threadIdx.x;
  if (!P.image_build || v >= (P.image_bytes >> 4)) return;
  const int byte = v << 4;
  const auto* cells = AsGlobal(P.cells);
  unsigned q[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {          // (the pitch is a multiple of 8 bytes, not of 16)
    const int at = byte + 2 * c;
    const int Y = at / P.pitch, X = (at - Y * P.pitch) >> 1;
    const int gx = X - P.hl, gy = Y - P.ht;
    unsigned val = 0;
    if (static_cast<unsigned>(gx) < static_cast<unsigned>(P.nx) &&
        static_cast<unsigned>(gy) < static_cast<unsigned>(P.ny)) {
      const unsigned raw = cells[gy * P.nx + gx] & 32767u;
      val = raw ? (32767u - raw) >> kQShift : 0u;
    }
    q[c] = val;
  }
  uint4 out = make_uint4(q[0] | (q[1] << 16), q[2] | (q[3] << 16), q[4] | (q[5] << 16),
                         q[6] | (q[7] << 16));
  reinterpret_cast<uint4*>(P.qimage)[v] = out;
}

template <int K>
__device__ __forceinline__ int RowBcastAdd(int addrs, int lane_off) {
  int out;
  asm("v_add_u32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
      : "=v"(out) : "v"(addrs), "v"(lane_off), "i"(K));
  return out;
}
__device__ __forceinline__ uint2 LdsRead64(int addr) {
  const uint2v t = *reinterpret_cast<LdsUint2Ptr>(static_cast<uintptr_t>(addr));
  return make_uint2(t[0], t[1]);
}
__device__ __forceinline__ void Add3(uint32_t* acc, uint32_t a, uint32_t b) {
  asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(*acc) : "v"(a), "v"(b));
}

// One task: `iters` (a multiple of 16) entries of the two streams of this wavefront.  RPL =
// rows per lane; `row_stride` = H * pitch bytes.  acc32[j][c]: sum of cell c of the lane's
// block in its j-th row.
// The loop body is hand-scheduled: loads, waits and adds are all `asm volatile`, because the
// compiler's own s_waitcnt placement drains the LDS queue (lgkmcnt(0)) before every group of
// adds -- eight reads in flight, then none.  Here two banks of 2 x RPL reads alternate and every
// add waits for exactly the older bank (LDS returns in order: at most 2 RPL operations pending
// means the older bank has landed, whatever else the compiler has in flight).
template <int kImm>
__device__ __forceinline__ void LdsRead64Asm(uint2v* out, int addr) {
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(*out) : "v"(addr), "i"(kImm));
}
template <int kPending>
__device__ __forceinline__ void WaitLds() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(kPending) : "memory");
}
__device__ __forceinline__ void Add3Asm(uint32_t* acc, uint32_t a, uint32_t b) {
  asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(*acc) : "v"(a), "v"(b));
}

template <int RPL, int kRowStride, int K>
__device__ __forceinline__ void PairLoad(uint2v (&bank)[2][RPL], int addrs, int lane_off,
                                         int row_stride_rt) {
  const int va0 = RowBcastAdd<K>(addrs, lane_off);
  const int va1 = RowBcastAdd<K + 1>(addrs, lane_off);
  if constexpr (kRowStride > 0) {
    LdsRead64Asm<0>(&bank[0][0], va0);
    if constexpr (RPL > 1) LdsRead64Asm<kRowStride>(&bank[0][1], va0);
    if constexpr (RPL > 2) LdsRead64Asm<2 * kRowStride>(&bank[0][2], va0);
    if constexpr (RPL > 3) LdsRead64Asm<3 * kRowStride>(&bank[0][3], va0);
    LdsRead64Asm<0>(&bank[1][0], va1);
    if constexpr (RPL > 1) LdsRead64Asm<kRowStride>(&bank[1][1], va1);
    if constexpr (RPL > 2) LdsRead64Asm<2 * kRowStride>(&bank[1][2], va1);
    if constexpr (RPL > 3) LdsRead64Asm<3 * kRowStride>(&bank[1][3], va1);
  } else {
#pragma unroll
    for (int j = 0; j < RPL; ++j) LdsRead64Asm<0>(&bank[0][j], va0 + j * row_stride_rt);
#pragma unroll
    for (int j = 0; j < RPL; ++j) LdsRead64Asm<0>(&bank[1][j], va1 + j * row_stride_rt);
  }
}
template <int RPL>
__device__ __forceinline__ void PairAdd(const uint2v (&bank)[2][RPL], uint32_t (&lo)[RPL],
                                        uint32_t (&hi)[RPL]) {
#pragma unroll
  for (int j = 0; j < RPL; ++j) {
    Add3Asm(&lo[j], bank[0][j][0], bank[1][j][0]);
    Add3Asm(&hi[j], bank[0][j][1], bank[1][j][1]);
  }
}

// One task: `iters` (a multiple of 16) entries of the two streams of this wavefront.  RPL =
// rows per lane; kRowStride = H * pitch bytes when that is a compile-time constant (it then
// rides in the read's immediate offset), 0: runtime stride, one v_add per extra row.
// acc32[j][c]: sum of cell c of the lane's block in its j-th row.
template <int RPL, int kRowStride>
__device__ __forceinline__ void RowPairAccumulate(const uint16_t* my_list, int my_len, int iters,
                                                  int lane, int lane_off, int row_stride_rt,
                                                  uint32_t (&acc32)[RPL][4]) {
  uint32_t lo[RPL], hi[RPL];
#pragma unroll
  for (int j = 0; j < RPL; ++j) lo[j] = hi[j] = 0;
  const int groups = iters >> 4;
  constexpr int kBank = 2 * RPL;             // reads of one bank
  int e = lane & 15;
  int addrs = e < my_len ? static_cast<int>(my_list[e]) << 3 : 0;   // 0: the zero corner
  for (int g = 0; g < groups; ++g) {
    // the next group's entries are fetched under this group's reads
    const int e_next = e + 16;
    // (unconditional read of a slot inside the padded list, selected afterwards: a branch
    // around the read makes the compiler wait for it -- and for everything else -- at once)
    const int raw_next = my_list[min(e_next, iters - 1)];
    const int addrs_next = e_next < my_len ? raw_next << 3 : 0;
    uint2v a[2][RPL], b[2][RPL];
    PairLoad<RPL, kRowStride, 0>(a, addrs, lane_off, row_stride_rt);
    PairLoad<RPL, kRowStride, 2>(b, addrs, lane_off, row_stride_rt);
    WaitLds<kBank>(); PairAdd<RPL>(a, lo, hi); PairLoad<RPL, kRowStride, 4>(a, addrs, lane_off, row_stride_rt);
    WaitLds<kBank>(); PairAdd<RPL>(b, lo, hi); PairLoad<RPL, kRowStride, 6>(b, addrs, lane_off, row_stride_rt);
    WaitLds<kBank>(); PairAdd<RPL>(a, lo, hi); PairLoad<RPL, kRowStride, 8>(a, addrs, lane_off, row_stride_rt);
    WaitLds<kBank>(); PairAdd<RPL>(b, lo, hi); PairLoad<RPL, kRowStride, 10>(b, addrs, lane_off, row_stride_rt);
    WaitLds<kBank>(); PairAdd<RPL>(a, lo, hi); PairLoad<RPL, kRowStride, 12>(a, addrs, lane_off, row_stride_rt);
    WaitLds<kBank>(); PairAdd<RPL>(b, lo, hi); PairLoad<RPL, kRowStride, 14>(b, addrs, lane_off, row_stride_rt);
    WaitLds<kBank>(); PairAdd<RPL>(a, lo, hi);
    WaitLds<0>();     PairAdd<RPL>(b, lo, hi);
    if ((g & 3) == 3 || g + 1 == groups) {       // 64 entries: the 16-bit fields are full
#pragma unroll
      for (int j = 0; j < RPL; ++j) {
        acc32[j][0] += lo[j] & 0xffffu; acc32[j][1] += lo[j] >> 16;
        acc32[j][2] += hi[j] & 0xffffu; acc32[j][3] += hi[j] >> 16;
        lo[j] = hi[j] = 0;
      }
    }
    e = e_next;
    addrs = addrs_next;
  }
}

// grid (workgroups per match, matches), 1024 threads; dynamic LDS:
//   image[image_bytes] | acc[R][side^2] | bases[R][pchunks][4] | cnt[R][4] | pstart[R][4] |
//   tasks[task_cap] x 4 | ctl[16] | rots[R] | ax[n_pad] ay[n_pad] | list[R][cap] u16
template <int RPL, int kRowStride>
__global__ void __launch_bounds__(kBulkThreads)
Rt2DRowPairKernel(const Rt2DParams* __restrict__ params) {
  extern __shared__ __attribute__((aligned(16))) unsigned char bulk_smem[];
  const Rt2DParams& P = params[blockIdx.y];
  if (static_cast<int>(blockIdx.x) >= P.num_scans) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = P.n, n_pad = P.n_pad, pchunks = n_pad >> 6;
  const int side = 2 * P.nl + 1, cands = side * side;
  const int R = P.rounds_rot, B = P.blocks_per_row, H = P.half_rows;
  const int cap = P.pair_list_cap;
  int* acc = reinterpret_cast<int*>(bulk_smem + P.image_bytes);
  int* bases = acc + ((R * cands + 3) & ~3);   // [R][pchunks][4]: a chunk's offset inside its phase (16-byte aligned)
  int* cnt = bases + R * pchunks * 4;       // [R][4]
  int* pstart = cnt + R * 4;                // [R][4]: first list slot of the phase
  int* tasks = pstart + R * 4;              // [task_cap][4]: rr | phA << 8 | phB << 16, startA, startB, lenA | lenB << 16
  int* ctl = tasks + P.task_cap * 4;        // [0] number of tasks, [1] next task
  float2* rots = reinterpret_cast<float2*>(ctl + 16);       // [R]: this round's (cos, sin) pairs
  float* ax = reinterpret_cast<float*>(rots + R);
  float* ay = ax + n_pad;
  uint16_t* list = reinterpret_cast<uint16_t*>(ay + n_pad);

  unsigned long long* const tl = P.timeline;
  const int tl_block = blockIdx.y * gridDim.x + blockIdx.x;
  Stamp(tl, tl_block, 0);
  // ---- the image enters LDS as it is, by LDS-DMA (global_load_lds_dwordx4: 1 KiB per
  // wave-instruction, no registers, asynchronous): nothing reads it before the tasks, so the
  // copy runs under the discretisation; vmcnt(0) + barrier order the reads behind it -------
  {
    // (the cloud's loads go out first: their latency runs under the DMA issue)
    const auto* xyz = AsGlobal(P.xyz);
    const int i0 = min(tid, n - 1);
    const float px = xyz[3 * i0], py = xyz[3 * i0 + 1];
    const auto* src = (const __attribute__((address_space(1))) unsigned char*)P.qimage;
    auto* dst = (__attribute__((address_space(3))) unsigned char*)bulk_smem;
    const int kib = P.image_bytes >> 10;              // the image is padded to whole KiB
    for (int k = wave; k < kib; k += kBulkWaves)
      __builtin_amdgcn_global_load_lds(src + (k << 10) + (lane << 4), dst + (k << 10), 16, 0, 0);
    // the cloud pre-rotated by the initial yaw (once per workgroup)
    for (int i = tid; i < n_pad; i += kBulkThreads) {
      float x = 0.f, y = 0.f;
      if (i < n) {
        const float vx = i == tid ? px : xyz[3 * i], vy = i == tid ? py : xyz[3 * i + 1];
        RotateZ(P.init_qw, P.init_qz, vx, vy, &x, &y);
      }
      ax[i] = x;
      ay[i] = y;
    }
  }
  Stamp(tl, tl_block, 1);
  const float kScale = ((1.f - 0.1f) - (1.f - (1.f - 0.1f))) / 32766.f;   // (kMaxCC - kMinCC) / 32766
  const int lds_image = static_cast<int>(reinterpret_cast<uintptr_t>(
      (const __attribute__((address_space(3))) unsigned char*)bulk_smem));
  // Lane geometry inside a half-wavefront.
  const int li = lane & 31;
  const int row = li / B, blk = li - row * B;
  const bool lane_used = row < H;
  // (lanes beyond H * B read the image's first rows like everyone else and drop the result)
  const int lane_off = lds_image + (lane_used ? row * P.pitch + blk * 8 : 0);
  const int row_stride = H * P.pitch;

  for (int s0 = blockIdx.x; s0 < P.num_scans; s0 += gridDim.x * R) {
    const int round_rot = min(R, (P.num_scans - s0 + static_cast<int>(gridDim.x) - 1) /
                                     static_cast<int>(gridDim.x));
    __syncthreads();                       // previous round's accumulators have been read
    if (s0 == static_cast<int>(blockIdx.x)) Stamp(tl, tl_block, 2);
    for (int i = tid; i < round_rot * cands; i += kBulkThreads) acc[i] = 0;
    if (tid < round_rot * 4) cnt[tid] = 0;
    if (tid < 2) ctl[tid] = 0;
    if (tid < round_rot) rots[tid] = P.scan_rot[s0 + tid * gridDim.x];
    __syncthreads();
    // ---- discretise once: entry << 2 | phase stays in a register -------------------------
    int pk[kPairChunksPerWave];
    const int wave_chunks = round_rot * pchunks;
    {
      const float tx = P.tx, ty = P.ty;
      const double max_x = P.max_x, max_y = P.max_y, res = P.res, inv_res = P.inv_res;
      const int nl = P.nl, nx = P.nx, ny = P.ny, hl = P.hl, ht = P.ht, pitch = P.pitch;
      // CellIndexFast (cmx_device.h) with its per-call constants hoisted and its error bound
      // simplified upwards: |a| inv <= |b| (1 + 2^-23), so
      //   (|max| + |a|) inv 2^-23 + |b| 2^-21 + 2^-20  <=  c0 + |b| 2^-20,
      // c0 = |max| inv 2^-23 + 2^-20.  A larger bound only sends more points to the exact f64
      // expression; the result is the same lround either way.
      const float maxxf = static_cast<float>(max_x), maxyf = static_cast<float>(max_y);
      const float invf = static_cast<float>(inv_res);
      const float c0x = fabsf(maxxf) * invf * 0x1p-23f + 0x1p-20f;
      const float c0y = fabsf(maxyf) * invf * 0x1p-23f + 0x1p-20f;
      const auto cell = [&](float maxf, float c0, double max_d, float v) -> int {
        const float b = (maxf - v) * invf;
        const float t = b - 0.5f;
        const float r = rintf(t);
        const float margin = 0.5f - fabsf(t - r);
        if (margin > c0 + fabsf(b) * 0x1p-20f && fabsf(t) < 1e6f) return static_cast<int>(r);
        return CellIndexF64(max_d - static_cast<double>(v), res, inv_res);
      };
      int rr = wave / pchunks, pc = wave - rr * pchunks;         // chunk wave + 16 j, incrementally
#pragma unroll
      for (int j = 0; j < kPairChunksPerWave; ++j) {
        int packed = -1;
        if (wave + j * kBulkWaves < wave_chunks) {
          const float2 rot = rots[rr];
          const int i = pc * 64 + lane;
          if (i < n) {
            float bx, by;
            RotateZ(rot.x, rot.y, ax[i], ay[i], &bx, &by);
            const int cx = cell(maxyf, c0y, max_y, by + ty);
            const int cy = cell(maxxf, c0x, max_x, bx + tx);
            const int ix = min(max(cx, -(nl + 1)), nx + nl);
            const int iy = min(max(cy, -(nl + 1)), ny + nl);
            const int wx = ix - nl + hl, wy = iy - nl + ht;   // window start, image coordinates
            packed = ((wy * pitch + (wx & ~3) * 2) >> 1) | (wx & 3);   // (byte >> 3) << 2 | phase
          }
          // the chunk's place inside each phase list: ONE returning LDS atomic (lane q = phase q)
          const int c0 = __popcll(__ballot(packed >= 0 && (packed & 3) == 0));
          const int c1 = __popcll(__ballot(packed >= 0 && (packed & 3) == 1));
          const int c2 = __popcll(__ballot(packed >= 0 && (packed & 3) == 2));
          const int c3 = __popcll(__ballot(packed >= 0 && (packed & 3) == 3));
          if (lane < 4) {
            const int c = lane == 0 ? c0 : lane == 1 ? c1 : lane == 2 ? c2 : c3;
            bases[(rr * pchunks + pc) * 4 + lane] = atomicAdd(&cnt[rr * 4 + lane], c);
          }
        }
        pk[j] = packed;
        pc += kBulkWaves;
        while (pc >= pchunks) { pc -= pchunks; ++rr; }
      }
    }
    __syncthreads();
    if (s0 == static_cast<int>(blockIdx.x)) Stamp(tl, tl_block, 3);      // points discretised
    // ---- per rotation: phase offsets (lists padded to 16 entries), phases paired by size,
    // tasks of at most kPairTaskIters iterations ---------------------------------------------
    if (wave == 0) {                         // lane = rotation of the round (R <= 64)
      const int rr = lane;
      const bool live = rr < round_rot;
      int key[4], start = 0, first_slot[4];   // count << 2 | phase
#pragma unroll
      for (int ph = 0; ph < 4; ++ph) {
        const int c = live ? cnt[rr * 4 + ph] : 0;
        key[ph] = (c << 2) | ph;
        first_slot[ph] = start;
        if (live) pstart[rr * 4 + ph] = start;
        start += (c + 15) & ~15;
      }
      // the four phases by count, descending (sorting network of five exchanges)
#define CMX_CSWAP(I, J) { const int hi_k = max(key[I], key[J]), lo_k = min(key[I], key[J]); key[I] = hi_k; key[J] = lo_k; }
      CMX_CSWAP(0, 1) CMX_CSWAP(2, 3) CMX_CSWAP(0, 2) CMX_CSWAP(1, 3) CMX_CSWAP(1, 2)
#undef CMX_CSWAP
      const int la0 = key[0] >> 2, la1 = key[2] >> 2;
      const int mine = (la0 + kPairTaskIters - 1) / kPairTaskIters + (la1 + kPairTaskIters - 1) / kPairTaskIters;
      const int incl = WaveInclusiveScan(mine);
      int t = incl - mine;
      if (lane == 63) ctl[0] = incl;
#pragma unroll
      for (int pair = 0; pair < 2; ++pair) {
        const int pa = key[2 * pair] & 3, pb = key[2 * pair + 1] & 3;
        const int la = key[2 * pair] >> 2, lb = key[2 * pair + 1] >> 2;
        // (static selects instead of first_slot[pa]: no dynamically indexed private array)
        const int sa = pa == 0 ? first_slot[0] : pa == 1 ? first_slot[1] : pa == 2 ? first_slot[2] : first_slot[3];
        const int sb = pb == 0 ? first_slot[0] : pb == 1 ? first_slot[1] : pb == 2 ? first_slot[2] : first_slot[3];
        for (int off = 0; off < la; off += kPairTaskIters, ++t) {
          tasks[4 * t] = rr | (pa << 8) | (pb << 16);
          tasks[4 * t + 1] = sa + off;
          tasks[4 * t + 2] = sb + off;
          tasks[4 * t + 3] = min(kPairTaskIters, la - off) | (max(0, min(kPairTaskIters, lb - off)) << 16);
        }
      }
    }
    __syncthreads();
    // ---- scatter the entries from the registers into the phase lists ---------------------
    {
      int rr = wave / pchunks, pc = wave - rr * pchunks;
#pragma unroll
      for (int j = 0; j < kPairChunksPerWave; ++j) {
        if (wave + j * kBulkWaves < wave_chunks) {
          const int packed = pk[j];
          const int ph = packed & 3;
          const int4 b4 = *reinterpret_cast<const int4*>(&bases[(rr * pchunks + pc) * 4]);
          const int4 p4 = *reinterpret_cast<const int4*>(&pstart[rr * 4]);
          const unsigned long long m0 = __ballot(packed >= 0 && ph == 0);
          const unsigned long long m1 = __ballot(packed >= 0 && ph == 1);
          const unsigned long long m2 = __ballot(packed >= 0 && ph == 2);
          const unsigned long long m3 = __ballot(packed >= 0 && ph == 3);
          if (packed >= 0) {
            // the lane's own phase: its mask, its list start, its chunk base -- then ONE rank
            const unsigned long long mine = ph == 0 ? m0 : ph == 1 ? m1 : ph == 2 ? m2 : m3;
            const int first = ph == 0 ? p4.x + b4.x : ph == 1 ? p4.y + b4.y : ph == 2 ? p4.z + b4.z : p4.w + b4.w;
            const int rank = __builtin_amdgcn_mbcnt_hi(static_cast<unsigned>(mine >> 32),
                                                       __builtin_amdgcn_mbcnt_lo(static_cast<unsigned>(mine), 0));
            list[rr * cap + first + rank] = static_cast<uint16_t>(packed >> 2);
          }
        }
        pc += kBulkWaves;
        while (pc >= pchunks) { pc -= pchunks; ++rr; }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's share of the image has landed
    __syncthreads();
    if (s0 == static_cast<int>(blockIdx.x)) Stamp(tl, tl_block, 4);      // lists sorted by phase
    // ---- tasks, dealt dynamically: the halves of a wavefront run two phases --------------
    const int num_tasks = ctl[0];          // <= task_cap by construction (host)
    for (;;) {
      int t = 0;
      if (lane == 0) t = atomicAdd(&ctl[1], 1);
      t = __builtin_amdgcn_readfirstlane(t);
      if (t >= num_tasks) break;
      const int d0 = tasks[4 * t], d3 = tasks[4 * t + 3];
      const int rr = d0 & 255;
      const bool second = lane >= 32;
      const int phase = second ? (d0 >> 16) & 255 : (d0 >> 8) & 255;
      const int start = second ? tasks[4 * t + 2] : tasks[4 * t + 1];
      const int my_len = second ? d3 >> 16 : d3 & 0xffff;
      const int iters = ((d3 & 0xffff) + 15) & ~15;          // the first stream is the longer
      uint32_t acc32[RPL][4];
#pragma unroll
      for (int j = 0; j < RPL; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc32[j][c] = 0;
      RowPairAccumulate<RPL, kRowStride>(list + rr * cap + start, my_len, iters, lane, lane_off,
                                         row_stride, acc32);
      if (lane_used) {
        int* out = acc + rr * cands;
        const int d0x = blk * 4 - phase;           // candidate x index of the block's first cell
#pragma unroll
        for (int j = 0; j < RPL; ++j) {
          const int wrow = row + j * H;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int dxi = d0x + c;
            if (wrow < side && dxi >= 0 && dxi < side && acc32[j][c])
              atomicAdd(&out[dxi * side + wrow], static_cast<int>(acc32[j][c]));
          }
        }
      }
    }
    if (s0 == static_cast<int>(blockIdx.x)) Stamp(tl, tl_block, 5);      // this wave out of tasks
    __syncthreads();
    if (s0 == static_cast<int>(blockIdx.x)) Stamp(tl, tl_block, 6);      // all tasks done
    // ---- per candidate: integer sum out; weighted lower bound into the match's maximum,
    // weighted upper bound out for the finalist selection.  Bounds only SELECT finalists (their
    // scores are recomputed exactly), so f32 with slack is enough: the base value carries one
    // rounding of 2^-23 relative (1.2e-7 absolute) against kBoundSlack = 1e-4 --------------------
    float lb_max = 0.f;
    const float per_q = kScale * static_cast<float>(1 << kQShift) / static_cast<float>(n);
    const float width = kScale * static_cast<float>((1 << kQShift) - 1);
    for (int e = tid; e < round_rot * cands; e += kBulkThreads) {
      const int rr = e / cands, c = e - rr * cands;
      const int s = s0 + rr * gridDim.x;
      const int q = acc[e];
      P.qsum[static_cast<size_t>(s) * cands + c] = q;
      const int dxi = c / side, dyi = c - dxi * side;
      const float base = 0.1f + per_q * static_cast<float>(q);
      const float w = Rt2DWeight(P, s, dxi - P.nl, dyi - P.nl);
      const float lb = (base - static_cast<float>(kBoundSlack)) * w * (1.f - 1e-5f);
      P.ub[static_cast<size_t>(s) * cands + c] =
          (base + width + static_cast<float>(kBoundSlack)) * w * (1.f + 1e-5f);
      lb_max = fmaxf(lb_max, lb);
    }
    unsigned bits = __float_as_uint(fmaxf(lb_max, 0.f));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) bits = max(bits, __shfl_xor(bits, off, 64));
    if (lane == 0 && bits) atomicMax(&P.misc[0], bits);
    if (s0 == static_cast<int>(blockIdx.x)) Stamp(tl, tl_block, 7);
  }
  Stamp(tl, tl_block, 8);
}

// grid (num_scans, matches), 256 threads: the finalists of one rotation with the reference's
// sequential f32 sum (:61-75).  The sum is a chain of N dependent additions, but the N lookups
// behind it are independent: all threads fetch the probabilities of a finalist's points into
// LDS (a few loads per thread, all in flight at once), then ONE lane per finalist runs the
// chain out of LDS.  (One lane doing its own 891 gathers took 130 us of dependent round
// trips for a single finalist.)  Up to `group` finalists share a pass.
// Dynamic LDS: cells[n_pad] u32 | prob[group][n_pad + 1] f32 | fin[side^2].
__global__ void __launch_bounds__(256)
Rt2DExactKernel(const Rt2DParams* __restrict__ params, int group) {
  extern __shared__ __attribute__((aligned(16))) unsigned char exact_smem[];
  const Rt2DParams& P = params[blockIdx.y];
  const int s = blockIdx.x;
  if (s >= P.num_scans) return;
  const int tid = threadIdx.x;
  const int side = 2 * P.nl + 1, cands = side * side, n = P.n, n_pad = P.n_pad;
  unsigned long long* const tl = P.timeline;
  const int tl_block = P.timeline_exact_base + blockIdx.y * gridDim.x + blockIdx.x;
  Stamp(tl, tl_block, 0);
  uint32_t* cellbuf = reinterpret_cast<uint32_t*>(exact_smem);
  float* prob = reinterpret_cast<float*>(cellbuf + n_pad);
  int* fin = reinterpret_cast<int*>(prob + group * (n_pad + 1));
  __shared__ int nfin;
  if (tid == 0) nfin = 0;
  __syncthreads();
  const float kScale = ((1.f - 0.1f) - (1.f - (1.f - 0.1f))) / 32766.f;
  const float best_lb = __uint_as_float(P.misc[0]);
  const int* __restrict__ qsum = P.qsum + static_cast<size_t>(s) * cands;
  // Candidates of this rotation whose weighted upper bound reaches the best lower bound.
  for (int c = tid; c < cands; c += blockDim.x) {
    const int dxi = c / side, dyi = c - dxi * side;
    const double hi_score = 0.1 + static_cast<double>(kScale) *
                                      (static_cast<double>(qsum[c]) * (1 << kQShift) +
                                       ((1 << kQShift) - 1) * static_cast<double>(n)) / n;
    const float ub = static_cast<float>(hi_score + kBoundSlack) *
                     Rt2DWeight(P, s, dxi - P.nl, dyi - P.nl) * (1.f + 1e-5f);
    if (ub >= best_lb) fin[atomicAdd(&nfin, 1)] = c;
  }
  __syncthreads();
  const int count = nfin;
  if (count == 0) return;                 // most rotations
  Stamp(tl, tl_block, 1);
  {
    const float2 r = P.scan_rot[s];
    const Quat q0{P.init_qw, 0.f, 0.f, P.init_qz};
    const Quat qs{r.x, 0.f, 0.f, r.y};
    for (int i = tid; i < n; i += blockDim.x) {
      int ix, iy;
      Rt2DPointCell(P, q0, qs, F3{P.xyz[3 * i], P.xyz[3 * i + 1], P.xyz[3 * i + 2]}, &ix, &iy);
      cellbuf[i] = (static_cast<uint32_t>(ix) & 0xffffu) | (static_cast<uint32_t>(iy) << 16);
    }
  }
  __syncthreads();
  Stamp(tl, tl_block, 2);
  const auto* cells = AsGlobal(P.cells);
  const int row = n_pad + 1;              // odd row pitch: the chain lanes hit distinct banks
  for (int f0 = 0; f0 < count; f0 += group) {
    const int g = min(group, count - f0);
    for (int f = 0; f < g; ++f) {
      const int c = fin[f0 + f];
      const int dxi = c / side, dyi = c - dxi * side;
      const int dx = dxi - P.nl, dy = dyi - P.nl;
      for (int base = tid; base < n; base += 4 * blockDim.x) {
        unsigned raw[4];
        bool inside[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const uint32_t pc = cellbuf[min(base + k * static_cast<int>(blockDim.x), n - 1)];
          const int x = static_cast<short>(pc & 0xffffu) + dx;
          const int y = static_cast<short>(pc >> 16) + dy;
          inside[k] = static_cast<unsigned>(x) < static_cast<unsigned>(P.nx) &&
                      static_cast<unsigned>(y) < static_cast<unsigned>(P.ny);
          raw[k] = cells[inside[k] ? P.nx * y + x : 0];     // unconditional load, masked below
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int i = base + k * blockDim.x;
          if (i < n) prob[f * row + i] = inside[k] ? CellProbability(raw[k]) : 0.1f;   // kMinProbability
        }
      }
    }
    __syncthreads();
    if (tid < g) {
      const float* mine = prob + tid * row;
      float sum = 0.f;
      int i = 0;
      for (; i + 8 <= n; i += 8) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = mine[i + k];
#pragma unroll
        for (int k = 0; k < 8; ++k) sum += v[k];            // in point order
      }
      for (; i < n; ++i) sum += mine[i];
      const float score = sum / static_cast<float>(n);
      const int c = fin[f0 + tid];
      const int dxi = c / side, dyi = c - dxi * side;
      const int cg = (s * side + dxi) * side + dyi;           // x outer, y inner (:99-113)
      const unsigned slot = atomicAdd(&P.misc[1], 1u);
      if (slot < static_cast<unsigned>(kFinalistCap)) {
        unsigned* pair = slot < static_cast<unsigned>(kFinalistHead)
                             ? P.misc + 2 + 2 * slot
                             : P.overflow + 2 * (slot - kFinalistHead);
        pair[0] = static_cast<unsigned>(cg);
        pair[1] = __float_as_uint(score);
      }
    }
    __syncthreads();
  }
  Stamp(tl, tl_block, 3);
}

// grid (matches, kExactSplit), 1024 threads: the same finalists, a few workgroups per MATCH (round 3).  The
// per-rotation grid above launches num_scans x matches blocks of which 96 % only find out that
// their rotation has no finalist (3456 blocks, 15 us for 128 matches of C1: more than a quarter
// of the bulk kernel).  Here a match's 27 x 169 bounds are scanned by one workgroup (coalesced),
// the few finalists are grouped by rotation, and per rotation with finalists: discretise once,
// all threads gather the probabilities of up to `group` finalists into LDS, one lane per
// finalist runs the reference's f32 chain out of LDS (32 values in flight ahead of the adds).
// Dynamic LDS: cells[n_pad] u32 | prob[group][n_pad + 1] f32 | rot_count[num_scans] | fin[kFinalistCap].
__global__ void __launch_bounds__(1024)
Rt2DExactMatchKernel(const Rt2DParams* __restrict__ params, int group) {
  extern __shared__ __attribute__((aligned(16))) unsigned char exact_smem[];
  const Rt2DParams& P = params[blockIdx.x];
  const int tid = threadIdx.x;
  const int side = 2 * P.nl + 1, cands = side * side, n = P.n, n_pad = P.n_pad;
  const int total = P.num_scans * cands;
  unsigned long long* const tl = P.timeline;
  const int tl_block = P.timeline_exact_base + blockIdx.x * gridDim.y + blockIdx.y;
  Stamp(tl, tl_block, 0);
  uint32_t* cellbuf = reinterpret_cast<uint32_t*>(exact_smem);
  float* prob = reinterpret_cast<float*>(cellbuf + n_pad);
  int* rot_count = reinterpret_cast<int*>(prob + group * (n_pad + 1));
  int* fin = rot_count + ((P.num_scans + 3) & ~3);
  __shared__ int nfin;
  __shared__ int sel[18];
  if (tid == 0) nfin = 0;
  for (int s = tid; s < P.num_scans; s += blockDim.x) rot_count[s] = 0;
  __syncthreads();
  const float best_lb = __uint_as_float(P.misc[0]);
  // Candidates whose weighted upper bound (stored by the bulk kernel) reaches the best lower
  // bound.  Every block of a match selects the same list (in its own order).
  for (int e = tid; e < total; e += blockDim.x) {
    if (P.ub[e] >= best_lb) {
      const int slot = atomicAdd(&nfin, 1);
      if (slot < kFinalistCap) fin[slot] = e;
      atomicAdd(&rot_count[e / cands], 1);
    }
  }
  __syncthreads();
  const int count = nfin;
  if (count > kFinalistCap) {              // flat landscape: the host repeats the match on the
    if (tid == 0 && blockIdx.y == 0) P.misc[1] = count;       // per-candidate kernels
    return;
  }
  Stamp(tl, tl_block, 1);
  const auto* cells = AsGlobal(P.cells);
  const auto* xyz = AsGlobal(P.xyz);
  const int row = n_pad + 1;              // odd row pitch: the chain lanes hit distinct banks
  const Quat q0{P.init_qw, 0.f, 0.f, P.init_qz};
  int rank = 0;                           // rotations with finalists are dealt to the match's blocks
  for (int s = 0; s < P.num_scans; ++s) {
    if (rot_count[s] == 0) continue;      // (uniform: LDS value, no writer since the barrier)
    if (rank++ % static_cast<int>(gridDim.y) != static_cast<int>(blockIdx.y)) continue;
    __syncthreads();                      // the previous rotation's cells and sums are done with
    {
      const float2 r = P.scan_rot[s];
      const Quat qs{r.x, 0.f, 0.f, r.y};
      for (int i = tid; i < n; i += blockDim.x) {
        int ix, iy;
        Rt2DPointCell(P, q0, qs, F3{xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]}, &ix, &iy);
        cellbuf[i] = (static_cast<uint32_t>(ix) & 0xffffu) | (static_cast<uint32_t>(iy) << 16);
      }
    }
    __syncthreads();
    // this rotation's finalists, `group` (<= 16) at a time: thread 0 picks them from the list
    int next = 0;
    for (;;) {
      if (tid == 0) {
        int g = 0;
        for (; next < count && g < group; ++next) {
          const int e = fin[next];
          if (e / cands == s) sel[g++] = e - s * cands;
        }
        sel[16] = g;
        sel[17] = next;
      }
      __syncthreads();
      const int g = sel[16];
      next = sel[17];
      if (g == 0) break;
      for (int f = 0; f < g; ++f) {
        const int c = sel[f];
        const int dxi = c / side, dyi = c - dxi * side;
        const int dx = dxi - P.nl, dy = dyi - P.nl;
        for (int i = tid; i < n; i += blockDim.x) {
          const uint32_t pc = cellbuf[i];
          const int x = static_cast<short>(pc & 0xffffu) + dx;
          const int y = static_cast<short>(pc >> 16) + dy;
          const bool inside = static_cast<unsigned>(x) < static_cast<unsigned>(P.nx) &&
                              static_cast<unsigned>(y) < static_cast<unsigned>(P.ny);
          const unsigned raw = cells[inside ? P.nx * y + x : 0];
          prob[f * row + i] = inside ? CellProbability(raw) : 0.1f;   // kMinProbability
        }
      }
      __syncthreads();
      if (tid < g) {
        const float* vals = prob + tid * row;
        float sum = 0.f;
        int i = 0;
        for (; i + 32 <= n; i += 32) {
          float v[32];
#pragma unroll
          for (int k = 0; k < 32; ++k) v[k] = vals[i + k];
#pragma unroll
          for (int k = 0; k < 32; ++k) sum += v[k];            // in point order
        }
        for (; i < n; ++i) sum += vals[i];
        const float score = sum / static_cast<float>(n);
        const int c = sel[tid];
        const int dxi = c / side, dyi = c - dxi * side;
        const int cg = (s * side + dxi) * side + dyi;           // x outer, y inner (:99-113)
        const unsigned slot = atomicAdd(&P.misc[1], 1u);
        if (slot < static_cast<unsigned>(kFinalistCap)) {
          unsigned* pair = slot < static_cast<unsigned>(kFinalistHead)
                               ? P.misc + 2 + 2 * slot
                               : P.overflow + 2 * (slot - kFinalistHead);
          pair[0] = static_cast<unsigned>(cg);
          pair[1] = __float_as_uint(score);
        }
      }
      __syncthreads();
      if (g < group) break;
    }
  }
  Stamp(tl, tl_block, 3);
}

size_t Align16(size_t v) { return (v + 15) & ~static_cast<size_t>(15); }

}  // namespace

Rt2DImageCache::~Rt2DImageCache() {
  if (image) (void)hipFree(image);
}

// A batch of independent matches (one per trajectory / robot) in one set of launches.  Per
// item `cells` is a host buffer, or -- when `device_cells` is given -- ignored in favour of
// a grid that already lives in HBM (cmx_grid2d): nothing but the scan is uploaded then.
namespace {
// CMX_RT2D_BULK=0 keeps every match on the one-thread-per-candidate kernels (parity tests
// run both paths).
bool BulkEnabled() {
  const char* e = getenv("CMX_RT2D_BULK");
  return !(e && e[0] == '0');
}

// Returns false when the integer bulk pass produced more finalists than the list holds (a
// flat score landscape); the caller then repeats the batch on the per-candidate kernels.
bool Rt2DMatchBatchImpl(const cmx_rt_options* options, const Rt2DItem* items, int num,
                        int32_t device, cmx_match_stats* stats, bool force_legacy) {
  CMX_REQUIRE(options && items && num >= 1, "null argument");
  // CMX_HOST_TRACE=1: wall-clock of the host phases (tools only).
  static const bool host_trace = [] { const char* e = getenv("CMX_HOST_TRACE"); return e && e[0] == '1'; }();
  auto t_last = std::chrono::steady_clock::now();
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
  struct Plan {
    int n, nx, ny, nl, na, num_scans, n_pad, pad;
    long long side, num_candidates, stride, rows;
    double res, step;
    float q0w, q0z;
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
    pl.n = it.n; pl.nx = it.limits->num_x_cells; pl.ny = it.limits->num_y_cells;
    pl.res = it.limits->resolution;
  }
  lap("args");
  // SearchParameters of every item (a range scan over its cloud, acos): on the host pool.
  ParallelFor(num, 8, [&](int m) {
    const Rt2DItem& it = items[m];
    Plan& pl = plan[m];
    const int n = pl.n;
    const double res = pl.res;
    // SearchParameters on the cloud pre-rotated by the initial yaw (:123-130).
    const float ha0 = 0.5f * static_cast<float>(it.initial->theta);
    const float q0w = std::cos(ha0), q0z = std::sin(ha0) * 1.f;
    // Longest xy range of the cloud pre-rotated by the initial yaw (:123-130, :27-36).  The
    // rotation is the device's RotateZ (cmx_device.h: bit-identical to Eigen's product by
    // (w, 0, 0, z) for finite inputs); sqrt is monotone and correctly rounded, so the maximum
    // of the norms is the norm of the largest squared norm.  Eight independent maxima in
    // structure-of-arrays form: the loop vectorises (IEEE adds and multiplies only, no
    // contraction: the same bits in every lane as in the scalar expression).
    constexpr int kLanes = 8;
    float max_sq[kLanes];
    for (int k = 0; k < kLanes; ++k) max_sq[k] = 0.f;
    const auto squared_range = [q0w, q0z](float px, float py) {
      float uvx = -(q0z * py), uvy = q0z * px;
      uvx += uvx; uvy += uvy;
      const float cxx = -(q0z * uvy), cyy = q0z * uvx;
      const float rx = (px + q0w * uvx) + cxx, ry = (py + q0w * uvy) + cyy;
      return rx * rx + ry * ry;
    };
    int i = 0;
    if (it.far_points) {            // only these points can hold the f32 maximum (cmx_cloud)
      for (int k = 0; k < it.num_far_points; ++k) {
        const int idx = it.far_points[k];
        max_sq[0] = std::max(max_sq[0], squared_range(it.xyz[3 * idx], it.xyz[3 * idx + 1]));
      }
      i = n;
    }
    for (; i + kLanes <= n; i += kLanes) {
      float px[kLanes], py[kLanes];
      for (int k = 0; k < kLanes; ++k) { px[k] = it.xyz[3 * (i + k)]; py[k] = it.xyz[3 * (i + k) + 1]; }
      for (int k = 0; k < kLanes; ++k) max_sq[k] = std::max(max_sq[k], squared_range(px[k], py[k]));
    }
    for (; i < n; ++i)
      max_sq[0] = std::max(max_sq[0], squared_range(it.xyz[3 * i], it.xyz[3 * i + 1]));
    float max_all = 0.f;
    for (int k = 0; k < kLanes; ++k) max_all = std::max(max_all, max_sq[k]);
    const float max_scan_range = std::max(static_cast<float>(3.f * res), std::sqrt(max_all));
    const double kSafetyMargin = 1. - 1e-3;
    const float range_sq = max_scan_range * (max_scan_range * 1.f);
    pl.step = kSafetyMargin * std::acos(1. - (res * (res * 1.)) / (2. * range_sq));
    pl.na = std::ceil(options->angular_search_window / pl.step);
    pl.q0w = q0w; pl.q0z = q0z;
  });
  lap("range");
  for (int m = 0; m < num; ++m) {
    const Rt2DItem& it = items[m];
    Plan& pl = plan[m];
    const int n = pl.n;
    const double res = pl.res;
    pl.num_scans = 2 * pl.na + 1;
    pl.nl = std::ceil(options->linear_search_window / res);
    CMX_REQUIRE(pl.num_scans >= 1 && pl.num_scans < (1 << 16) && pl.nl >= 0 && pl.nl < (1 << 12),
                "unsupported search window");
    pl.side = 2ll * pl.nl + 1;
    pl.num_candidates = pl.side * pl.side * pl.num_scans;
    CMX_REQUIRE(pl.num_candidates < (1ll << 30), "search window too large");
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

  // ---- LDS-staged integer bulk pass: eligibility and geometry ------------------------
  struct Bulk { int wp, hp, hl, ht, bpr, rounds, list_cap, task_cap; bool xyz_lds; size_t lds; size_t off_qsum; };
  std::vector<Bulk> bulk(num);
  bool use_bulk = !tsdf && !force_legacy && BulkEnabled();
  size_t bulk_lds = 0, qsum_total = 0;
  for (int m = 0; m < num && use_bulk; ++m) {
    const Plan& pl = plan[m];
    Bulk& b = bulk[m];
    const int side = static_cast<int>(pl.side);
    b.bpr = (side + 3 + 3) / 4;
    b.hl = (2 * pl.nl + 4 + 3) & ~3;
    b.ht = 2 * pl.nl + 1;
    b.hp = pl.ny + 4 * pl.nl + 2;
    int wp = (pl.nx + b.hl + 4 * b.bpr + 15) & ~15;
    if ((wp >> 4) % 2 == 0) wp += 16;               // 16 x odd: conflict-free row pitch
    b.wp = wp;
    b.list_cap = pl.n_pad + 4 * kQChunk;
    b.task_cap = pl.n_pad / 64 + 4;
    const size_t grid_bytes = static_cast<size_t>(b.wp) * b.hp * 2;
    const size_t per_rot = 4 * (static_cast<size_t>(pl.n_pad) + b.list_cap +
                                4 * static_cast<size_t>(pl.n_pad / 64) + side * side + b.task_cap);
    const size_t budget = 160 * 1024 - 512;
    const size_t fixed = grid_bytes + 256;
    if (pl.n > kBulkMaxPoints || pl.nx > 16384 || pl.ny > 16384 ||
        fixed + per_rot > budget) {
      use_bulk = false;
      break;
    }
    // The cloud itself goes to LDS when at least two rotations per round still fit.
    const size_t xyz_bytes = 12 * static_cast<size_t>(pl.n_pad);
    b.xyz_lds = fixed + xyz_bytes + 2 * per_rot <= budget;
    const size_t avail = budget - fixed - (b.xyz_lds ? xyz_bytes : 0);
    b.rounds = static_cast<int>(std::min<size_t>(kMaxRoundRot, avail / per_rot));
    b.lds = fixed + (b.xyz_lds ? xyz_bytes : 0) + b.rounds * per_rot;
    bulk_lds = std::max(bulk_lds, b.lds);
    b.off_qsum = qsum_total;
    qsum_total += static_cast<size_t>(pl.num_scans) * side * side;
  }
  // ---- row-pair pass (round 3): geometry; falls back to the chunked bulk kernel above for
  // windows that need more than kMaxRowsPerLane rows per lane, mixed geometries in one batch,
  // or when the image and one rotation's lists do not fit in LDS together ----------------
  struct Pair { int B, H, rpl, hl, ht, hp, pitch, image_bytes, rounds, cap, task_cap; size_t lds, off_image; };
  std::vector<Pair> pairg(num);
  // CMX_RT2D_ROWPAIR=0 keeps the chunked bulk kernel (parity tests run all three paths).
  const char* pair_env = getenv("CMX_RT2D_ROWPAIR");
  bool use_pair = use_bulk && !(pair_env && pair_env[0] == '0');
  size_t pair_lds = 0, image_total = 0;
  for (int m = 0; m < num && use_pair; ++m) {
    const Plan& pl = plan[m];
    Pair& g = pairg[m];
    const int side = static_cast<int>(pl.side);
    g.B = (side + 3 + 3) / 4;
    if (g.B > 32) { use_pair = false; break; }
    g.H = 32 / g.B;
    g.rpl = (side + g.H - 1) / g.H;
    g.hl = (2 * pl.nl + 4 + 3) & ~3;               // == 4 * B: the null entry's blocks are halo
    g.ht = 2 * pl.nl + 1;
    g.hp = pl.ny + g.ht + g.rpl * g.H;
    const int min_pitch = (2 * (pl.nx + g.hl + 4 * g.B) + 7) & ~7;
    g.pitch = 0;
    if (m > 0 && plan[m - 1].nx == pl.nx && plan[m - 1].nl == pl.nl) {
      g.pitch = pairg[m - 1].pitch;           // same geometry as the previous item (the usual batch)
    } else {
      for (int cand = min_pitch; cand < min_pitch + 512; cand += 8) {
        // conflict-free: the H x B 8-byte blocks of a half-wavefront touch 2 H B distinct banks
        unsigned long long used = 0;
        bool ok = true;
        for (int r = 0; r < g.H && ok; ++r)
          for (int b = 0; b < g.B && ok; ++b)
            for (int w = 0; w < 2; ++w) {
              const int bank = ((r * cand + b * 8) / 4 + w) & 63;
              if (used >> bank & 1) ok = false;
              used |= 1ull << bank;
            }
        if (ok) { g.pitch = cand; break; }
      }
    }
    g.image_bytes = (g.hp * g.pitch + 16 + 1023) & ~1023;        // whole KiB: LDS-DMA granule
    g.cap = pl.n_pad + 4 * 16;
    const int pchunks = pl.n_pad / 64;
    const size_t budget = 160 * 1024 - 512;
    const size_t fixed = static_cast<size_t>(g.image_bytes) + 64 + 8 * static_cast<size_t>(pl.n_pad) + 64;
    const auto per_round = [&](int R) {
      const size_t tasks = static_cast<size_t>(R) * 2 * (pl.n_pad / kPairTaskIters + 1);
      return 4 * (((static_cast<size_t>(R) * side * side + 3) & ~size_t{3}) +
                  static_cast<size_t>(R) * pchunks * 4 + 8 * static_cast<size_t>(R) + 4 * tasks +
                  2 * static_cast<size_t>(R)) +
             2 * static_cast<size_t>(R) * g.cap;
    };
    int R = std::min({pl.num_scans, 64, kPairChunksPerWave * kBulkWaves / pchunks});
    while (R >= 1 && fixed + per_round(R) > budget) --R;
    if (g.pitch == 0 || g.rpl > kMaxRowsPerLane || R < 1 || pl.n > kBulkMaxPoints ||
        g.image_bytes > (1 << 19) ||
        (m > 0 && (g.rpl != pairg[0].rpl || g.H * g.pitch != pairg[0].H * pairg[0].pitch))) {
      use_pair = false;
      break;
    }
    g.rounds = R;
    g.task_cap = R * 2 * (pl.n_pad / kPairTaskIters + 1);
    g.lds = fixed + per_round(R);
    pair_lds = std::max(pair_lds, g.lds);
    g.off_image = image_total;
    image_total += Align16(static_cast<size_t>(g.image_bytes));
  }
  // Exact kernel: finalists per pass so that cells + probabilities + list stay within 60 KB.
  int exact_group = 8;
  size_t exact_lds = 0;
  if (use_bulk) {
    size_t max_npad = 0, max_cands = 0;
    for (int m = 0; m < num; ++m) {
      max_npad = std::max<size_t>(max_npad, plan[m].n_pad);
      max_cands = std::max<size_t>(max_cands, plan[m].side * plan[m].side);
    }
    const size_t fixed = 4 * max_npad + 4 * max_cands + 16;
    const size_t budget = 60 * 1024;
    if (fixed + 4 * (max_npad + 1) > budget) {
      use_bulk = false;
    } else {
      exact_group = static_cast<int>(std::min<size_t>(8, (budget - fixed) / (4 * (max_npad + 1))));
      exact_lds = fixed + 4 * (max_npad + 1) * exact_group;
    }
  }
  // The per-match result words ride in the upload (zeroed) so that no kernel has to clear
  // them before the bulk kernel's atomicMax.
  const size_t off_misc = in_bytes;
  in_bytes += Align16(sizeof(unsigned) * 128 * static_cast<size_t>(num));

  lap("plan");
  WorkspaceLease ws(device);
  lap("lease");
  char* h_in = ws->pinned[0].ReserveAs<char>(in_bytes);
  char* d_in = ws->dev[0].ReserveAs<char>(in_bytes);
  // (scratch of the per-candidate kernels; the bulk path needs none of it)
  int* d_offsets = ws->dev[1].ReserveAs<int>(use_bulk ? 1 : offsets_total);
  char* d_padded = ws->dev[2].ReserveAs<char>(use_bulk ? 16 : padded_bytes);
  float* d_unweighted = ws->dev[3].ReserveAs<float>(use_bulk ? 1 : scores_total);
  float* d_weighted = ws->dev[4].ReserveAs<float>(use_bulk ? 1 : scores_total);
  static_assert(2 + 2 * kFinalistHead <= 128, "a match's head must fit its 128-word slot");
  unsigned* d_misc = reinterpret_cast<unsigned*>(d_in + off_misc);
  std::memset(h_in + off_misc, 0, sizeof(unsigned) * 128 * static_cast<size_t>(num));
  int* d_qsum = use_bulk ? ws->dev[7].ReserveAs<int>(qsum_total) : nullptr;
  float* d_ub = use_pair ? ws->dev[10].ReserveAs<float>(qsum_total) : nullptr;
  unsigned* d_overflow = ws->dev[6].ReserveAs<unsigned>(static_cast<size_t>(num) * 2 *
                                                        (kFinalistCap - kFinalistHead));
  unsigned* h_misc = ws->pinned[1].ReserveAs<unsigned>(static_cast<size_t>(num) * 128);
  char* d_images = use_pair ? ws->dev[9].ReserveAs<char>(image_total) : nullptr;
  // Images: a resident grid keeps its own (built once per grid version and window); everything
  // else is built into scratch by this call.  A cache being (re)built stays locked until the
  // stream has been waited for, so a concurrent match on the same grid sees a finished image.
  std::vector<uint16_t*> image_of(num, nullptr);
  std::vector<int> build_image(num, 0);
  std::vector<std::unique_lock<std::mutex>> cache_locks;
  if (use_pair) {
    for (int m = 0; m < num; ++m) {
      const Pair& g = pairg[m];
      Rt2DImageCache* c = items[m].image_cache;
      bool seen_before = false;                      // the same grid earlier in this batch
      for (int k = 0; k < m && !seen_before; ++k)
        if (c && items[k].image_cache == c) { seen_before = true; image_of[m] = image_of[k]; }
      if (seen_before) continue;
      if (!c || !items[m].device_cells) {
        image_of[m] = reinterpret_cast<uint16_t*>(d_images + g.off_image);
        build_image[m] = 1;
        continue;
      }
      std::unique_lock<std::mutex> lock(c->mutex);
      const bool valid = c->image && c->version == items[m].grid_version && c->nl == plan[m].nl &&
                         c->nx == plan[m].nx && c->ny == plan[m].ny && c->pitch == g.pitch &&
                         c->hp == g.hp && c->image_bytes == g.image_bytes;
      if (!valid) {
        if (c->capacity < static_cast<size_t>(g.image_bytes)) {
          if (c->image) (void)hipFree(c->image);
          c->image = nullptr;
          c->capacity = 0;
          CMX_HIP(hipMalloc(reinterpret_cast<void**>(&c->image), g.image_bytes));
          c->capacity = g.image_bytes;
        }
        c->version = items[m].grid_version; c->nl = plan[m].nl; c->nx = plan[m].nx;
        c->ny = plan[m].ny; c->pitch = g.pitch; c->hp = g.hp; c->image_bytes = g.image_bytes;
        build_image[m] = 1;
        cache_locks.push_back(std::move(lock));
      }
      image_of[m] = c->image;
    }
  }

  // CMX_TIMELINE=1: stamps of the bulk blocks, then of the exact blocks.
  unsigned long long* d_timeline = nullptr;
  int per_match_wgs = static_cast<int>(max_scans);
  if (use_bulk) {
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
    // Workgroups per match: one rotation each for a few matches (latency), fewer for big
    // batches (the grid is staged once per workgroup): about two rounds of the chip.
    if (static_cast<long long>(per_match_wgs) * num > 2ll * cus)
      per_match_wgs = std::max(1, std::min<int>(per_match_wgs, (2 * cus + num - 1) / num));
    // (row-pair kernel: the image is copied once per workgroup and its tasks are dealt
    // dynamically: ONE round of the chip)
    if (use_pair && static_cast<long long>(per_match_wgs) * num > cus)
      per_match_wgs = std::max(1, std::min<int>(per_match_wgs, (cus + num - 1) / num));
    if (const char* e = getenv("CMX_RT2D_WGS")) per_match_wgs = std::max(1, atoi(e));   // experiments
  }
  const int timeline_bulk_blocks = per_match_wgs * num;
  const int timeline_blocks = timeline_bulk_blocks + static_cast<int>(max_scans) * num;
  if (use_bulk && TimelineEnabled()) {
    const size_t bytes = static_cast<size_t>(timeline_blocks) * kTimelineStamps * 8;
    d_timeline = static_cast<unsigned long long*>(ws->dev[8].Reserve(bytes));
    CMX_HIP(hipMemsetAsync(d_timeline, 0, bytes, ws->stream));
  }
  Rt2DParams* h_params = reinterpret_cast<Rt2DParams*>(h_in);
  ParallelFor(num, 8, [&](int m) {
    const Rt2DItem& it = items[m];
    const Plan& pl = plan[m];
    const size_t cell_count = static_cast<size_t>(pl.nx) * pl.ny;
    if (!it.device_xyz) std::memcpy(h_in + pl.off_xyz, it.xyz, 3 * sizeof(float) * pl.n);
    float2* h_rot = reinterpret_cast<float2*>(h_in + pl.off_rot);
    double delta_theta = -pl.na * pl.step;
    for (int s = 0; s < pl.num_scans; ++s, delta_theta += pl.step) {
      const float ha = 0.5f * static_cast<float>(delta_theta);
      h_rot[s] = make_float2(std::cos(ha), std::sin(ha) * 1.f);
    }
    if (!it.device_cells)
      std::memcpy(h_in + pl.off_cells, it.cells, sizeof(uint16_t) * cell_count);
    if (tsdf) std::memcpy(h_in + pl.off_weights, it.weight_cells, sizeof(uint16_t) * cell_count);

    Rt2DParams P{};
    P.cells = it.device_cells ? it.device_cells
                              : reinterpret_cast<const uint16_t*>(d_in + pl.off_cells);
    P.weights = tsdf ? reinterpret_cast<const uint16_t*>(d_in + pl.off_weights) : nullptr;
    P.nx = pl.nx; P.ny = pl.ny;
    P.res = pl.res; P.max_x = it.limits->max_x; P.max_y = it.limits->max_y;
    P.inv_res = 1.0 / pl.res;
    P.tx = static_cast<float>(it.initial->x);
    P.ty = static_cast<float>(it.initial->y);
    P.init_qw = pl.q0w; P.init_qz = pl.q0z;
    P.nl = pl.nl; P.num_scans = pl.num_scans; P.num_angular = pl.na;
    P.step = pl.step;
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
    P.timeline = d_timeline;
    P.timeline_exact_base = timeline_bulk_blocks;
    if (use_bulk) {
      const Bulk& b = bulk[m];
      P.wp = b.wp; P.hp = b.hp; P.hl = b.hl; P.ht = b.ht;
      P.blocks_per_row = b.bpr; P.rounds_rot = b.rounds; P.list_cap = b.list_cap;
      P.task_cap = b.task_cap; P.xyz_in_lds = b.xyz_lds ? 1 : 0;
      P.qsum = d_qsum + b.off_qsum;
    }
    if (use_pair) {
      P.ub = d_ub + bulk[m].off_qsum;
      const Pair& g = pairg[m];
      P.hl = g.hl; P.ht = g.ht; P.hp = g.hp; P.blocks_per_row = g.B;
      P.half_rows = g.H; P.rows_per_lane = g.rpl; P.pitch = g.pitch;
      P.image_bytes = g.image_bytes; P.rounds_rot = g.rounds; P.pair_list_cap = g.cap;
      P.task_cap = g.task_cap;
      P.qimage = image_of[m];
      P.image_build = build_image[m];
    }
    h_params[m] = P;
  });
  lap("fill");
  SmallCopyAsync(d_in, h_in, in_bytes, /*to_device=*/true, ws->stream);
  const Rt2DParams* d_params = reinterpret_cast<const Rt2DParams*>(d_in);

  CMX_HIP(hipEventRecord(ws->ev_begin, ws->stream));
  const dim3 prep_grid(max_prep, 1, num), score_grid(max_tiles, max_scans, num),
      collect_grid(max_collect, 1, num);
  if (use_bulk) {
    static const bool lds_opt_in = [] {
      return hipFuncSetAttribute(reinterpret_cast<const void*>(Rt2DBulkKernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize,
                                 160 * 1024) == hipSuccess;
    }();
    CMX_REQUIRE(lds_opt_in, "cannot opt in to 160 KB of dynamic LDS");
    if (use_pair) {
      unsigned max_vecs = 0;
      for (int m = 0; m < num; ++m) max_vecs = std::max<unsigned>(max_vecs, pairg[m].image_bytes >> 4);
      bool any_build = false;
      for (int m = 0; m < num; ++m) any_build = any_build || build_image[m];
      if (any_build)
        Rt2DImageKernel<<<dim3(DivUp(max_vecs, 256), num), 256, 0, ws->stream>>>(d_params);
      CMX_HIP(hipEventRecord(ws->ev_k0, ws->stream));
      const dim3 grid(per_match_wgs, num);
      const int rpl = pairg[0].rpl, stride = pairg[0].H * pairg[0].pitch;
      const auto launch = [&](auto kernel) {
        static thread_local const void* opted = nullptr;
        const void* fn = reinterpret_cast<const void*>(kernel);
        if (opted != fn) {
          CMX_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
          opted = fn;
        }
        kernel<<<grid, kBulkThreads, pair_lds, ws->stream>>>(d_params);
      };
      if (rpl == 1) launch(Rt2DRowPairKernel<1, 0>);
      else if (rpl == 2 && stride == 8 * 224) launch(Rt2DRowPairKernel<2, 8 * 224>);
      else if (rpl == 2 && stride == 8 * 480) launch(Rt2DRowPairKernel<2, 8 * 480>);
      else if (rpl == 2 && stride == 8 * 736) launch(Rt2DRowPairKernel<2, 8 * 736>);
      else if (rpl == 2) launch(Rt2DRowPairKernel<2, 0>);
      else if (rpl == 3) launch(Rt2DRowPairKernel<3, 0>);
      else launch(Rt2DRowPairKernel<4, 0>);
    } else {
      CMX_HIP(hipEventRecord(ws->ev_k0, ws->stream));
      Rt2DBulkKernel<<<dim3(per_match_wgs, num), kBulkThreads, bulk_lds, ws->stream>>>(d_params);
    }
    CMX_HIP(hipEventRecord(ws->ev_k1, ws->stream));
    {
      // one workgroup per match (CMX_RT2D_EXACT_PER_ROTATION=1: the round-2 grid, for A/B runs)
      const char* per_rot = getenv("CMX_RT2D_EXACT_PER_ROTATION");
      if (!use_pair || (per_rot && per_rot[0] == '1')) {
        Rt2DExactKernel<<<dim3(max_scans, num), 256, exact_lds, ws->stream>>>(d_params, exact_group);
      } else {
        const size_t lds = exact_lds + 4 * ((max_scans + 3) & ~3u) + 4 * static_cast<size_t>(kFinalistCap);
        // (the kernel has a few static __shared__ words: ask for what it needs, not for all 160 KB)
        static thread_local size_t exact_opted = 0;
        if (lds > exact_opted) {
          CMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(Rt2DExactMatchKernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize,
                                      static_cast<int>(lds)));
          exact_opted = lds;
        }
        Rt2DExactMatchKernel<<<dim3(num, kExactSplit), 1024, lds, ws->stream>>>(d_params, exact_group);
      }
    }
  } else if (tsdf) {
    Rt2DPrepKernel<true><<<prep_grid, 256, 0, ws->stream>>>(d_params);
    CMX_HIP(hipEventRecord(ws->ev_k0, ws->stream));
    Rt2DScoreKernel<true><<<score_grid, 64, 0, ws->stream>>>(d_params);
  } else {
    Rt2DPrepKernel<false><<<prep_grid, 256, 0, ws->stream>>>(d_params);
    CMX_HIP(hipEventRecord(ws->ev_k0, ws->stream));
    // One match (81 waves) is bound by the latency of its sequential sums: the scalar
    // variant keeps 32 gathers in flight per wave (C1: 19 us vs 34 us).  From ~16
    // concurrent matches on, the gather path is the limit and four candidates per
    // gather win (128 matches: 169 us vs 293 us, 3.5e9 candidates/s).
    if (num >= 16)
      Rt2DScoreX4Kernel<<<dim3(max_tiles4, max_scans, num), 64, 0, ws->stream>>>(d_params);
    else
      Rt2DScoreKernel<false><<<score_grid, 64, 0, ws->stream>>>(d_params);
  }
  if (!use_bulk) {
    CMX_HIP(hipEventRecord(ws->ev_k1, ws->stream));
    Rt2DCollectKernel<<<collect_grid, 256, 0, ws->stream>>>(d_params);
  }
  CMX_HIP(hipGetLastError());
  CMX_HIP(hipEventRecord(ws->ev_end, ws->stream));
  SmallCopyAsync(h_misc, d_misc, sizeof(unsigned) * 128 * num, /*to_device=*/false, ws->stream);
  lap("enqueue");
  CMX_HIP(hipStreamSynchronize(ws->stream));
  cache_locks.clear();                     // rebuilt images are complete
  lap("wait");

  if (d_timeline) {
    ReportTimeline("Rt2DBulkKernel", d_timeline, timeline_bulk_blocks, ws->stream);
    ReportTimeline("Rt2DExactKernel", d_timeline + static_cast<size_t>(timeline_bulk_blocks) *
                                                       kTimelineStamps,
                   timeline_blocks - timeline_bulk_blocks, ws->stream);
  }
  if (use_bulk) {
    for (int m = 0; m < num; ++m)
      if (h_misc[static_cast<size_t>(m) * 128 + 1] > static_cast<unsigned>(kFinalistCap)) return false;
  }
  // Exact weighting + first-maximum on the finalists (:142-143,170-174).
  cmx_match_stats total{};
  std::vector<std::pair<int, float>> finalists;
  std::vector<unsigned> extra;
  std::vector<float> all;
  for (int m = 0; m < num; ++m) {
    const Rt2DItem& it = items[m];
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
    const int side_i = static_cast<int>(pl.side), nl = pl.nl, na = pl.na;
    const double res = pl.res, step = pl.step;
    float best_score = -1.f;
    int best = -1;
    for (const auto& f : finalists) {
      const int c = f.first;
      const int s = c / (side_i * side_i);
      const int rem = c - s * side_i * side_i;
      const int dx = rem / side_i - nl, dy = rem % side_i - nl;
      const double cx = -dy * res, cy = -dx * res;
      const double theta = (s - na) * step;
      const double t = std::hypot(cx, cy) * options->translation_delta_cost_weight +
                       std::abs(theta) * options->rotation_delta_cost_weight;
      float sc = f.second;
      sc *= std::exp(-(t * (t * 1.)));
      if (sc > best_score) { best_score = sc; best = c; }   // finalists ascend: first max wins
    }
    // CHECK_GT(score, 0) in the probability branch (:73); a TSDF may score 0 everywhere
    // (CHECK_GE at :56), in which case the first candidate wins like std::max_element.
    const int s = best / (side_i * side_i);
    const int rem = best - s * side_i * side_i;
    const int dx = rem / side_i - nl, dy = rem % side_i - nl;
    it.pose->x = it.initial->x + (-dy * res);
    it.pose->y = it.initial->y + (-dx * res);
    it.pose->theta = it.initial->theta + (s - na) * step;
    *it.score = best_score;
    total.candidates_scored += pl.num_candidates;
    total.coarse_candidates += pl.num_candidates;
    total.num_scans += pl.num_scans;
  }
  lap("finish");
  if (stats) {
    float ms = 0.f;
    CMX_HIP(hipEventElapsedTime(&ms, ws->ev_begin, ws->ev_end));
    total.device_ms = ms;
    CMX_HIP(hipEventElapsedTime(&ms, ws->ev_k0, ws->ev_k1));
    total.dominant_kernel_ms = ms;
    *stats = total;
  }
  if (host_trace) fprintf(stderr, "[cmx host] rt2d batch(%d):%s us (in %zu B)\n", num,
                          host_report.c_str(), in_bytes);
  return true;
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
  // A large batch goes out as two half-batches (CMX_RT2D_SPLIT parts) from as many host threads
  // (the caller and pool workers), each with its own workspace and stream: the host's preparation of one half runs
  // under the kernels of the other (of 172 us for 128 C1 matches the host held 75 before the
  // first launch).  Not when the caller ordered the work on a stream of its own (cmx_set_stream
  // is per thread), and CMX_RT2D_SPLIT=1 keeps one batch.
  static const int max_parts = [] {
    const char* e = getenv("CMX_RT2D_SPLIT");
    return e && e[0] ? std::max(1, std::min(8, atoi(e))) : 2;
  }();
  const int parts = std::min(max_parts, num / 32);             // (at least 32 matches per part)
  if (parts <= 1 || OverrideStream(device) != nullptr) {
    if (!Rt2DMatchBatchImpl(options, items, num, device, stats, false))
      Rt2DMatchBatchImpl(options, items, num, device, stats, true);
    return;
  }
  std::vector<cmx_match_stats> part(parts);
  std::vector<cmx_status> status(parts, CMX_OK);
  std::vector<std::string> error(parts);
  ParallelFor(parts, 0, [&](int h) {
    const int begin = static_cast<int>(static_cast<long long>(num) * h / parts),
              end = static_cast<int>(static_cast<long long>(num) * (h + 1) / parts);
    status[h] = Guard([&] {
      if (!Rt2DMatchBatchImpl(options, items + begin, end - begin, device, &part[h], false))
        Rt2DMatchBatchImpl(options, items + begin, end - begin, device, &part[h], true);
    });
    if (status[h] != CMX_OK) error[h] = LastError();      // (the message is per thread)
  });
  for (int h = 0; h < parts; ++h) {
    if (status[h] == CMX_OK) continue;
    SetLastError("%s", error[h].c_str());
    throw HipError{status[h]};
  }
  if (stats) {
    *stats = part[0];
    for (int h = 1; h < parts; ++h) {
      stats->candidates_scored += part[h].candidates_scored;
      stats->coarse_candidates += part[h].coarse_candidates;
      stats->nodes_expanded += part[h].nodes_expanded;
      stats->num_scans += part[h].num_scans;
      stats->device_ms = std::max(stats->device_ms, part[h].device_ms);     // the parts overlap
      stats->dominant_kernel_ms += part[h].dominant_kernel_ms;
    }
  }
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
