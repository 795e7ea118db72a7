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
#include <cmath>
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
  float tx, ty, init_qw, init_qz;
  int nl, num_scans, num_angular;
  double step, wt, wr;
  float max_tsd, max_weight;  // TSDF only
  const float2* scan_rot;
  int* offsets;              // [num_scans][n_pad] byte offsets into `padded` (see PrepKernel)
  int n_pad;                 // n rounded up to a multiple of 64
  void* padded;              // float[rows][stride] or float2[rows][stride]
  int pad, stride, rows;
  unsigned* misc;            // [0] max weighted score bits, [1] finalist count
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

template <bool kTsdf>
__global__ void __launch_bounds__(256)
Rt2DPrepKernel(Rt2DParams P, const float* __restrict__ xyz, int n) {
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
Rt2DScoreKernel(Rt2DParams P, int n, float* __restrict__ unweighted,
                float* __restrict__ weighted) {
  using Cell = typename Acc<kTsdf>::Cell;
  const int s = blockIdx.y;
  const int side = 2 * P.nl + 1;
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

// Candidates whose device-weighted score is within 1e-5 of the maximum, with
// their exact unweighted score: (index, score bits) pairs after the 2-word header.
__global__ void Rt2DCollectKernel(const float* __restrict__ weighted,
                                  const float* __restrict__ unweighted, int num_candidates,
                                  unsigned* __restrict__ misc, int capacity) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= num_candidates) return;
  const float threshold = __uint_as_float(misc[0]) * (1.f - 1e-5f);
  if (weighted[c] >= threshold) {
    const unsigned slot = atomicAdd(&misc[1], 1u);
    if (slot < static_cast<unsigned>(capacity)) {
      misc[2 + 2 * slot] = static_cast<unsigned>(c);
      misc[3 + 2 * slot] = __float_as_uint(unweighted[c]);
    }
  }
}

constexpr int kFinalistCap = 4096;
constexpr int kFinalistHead = 62;   // pairs returned with the first (512-byte) read

size_t Align16(size_t v) { return (v + 15) & ~static_cast<size_t>(15); }

}  // namespace

// `cells` is a host buffer, or -- when `device_cells` is given -- ignored in favour of a
// grid that already lives in HBM (cmx_grid2d): nothing but the scan is uploaded then.
void Rt2DMatch(const cmx_rt_options* options, const cmx_grid2d_limits* limits,
               const uint16_t* cells, const uint16_t* weight_cells, float max_tsd,
               float max_weight, const cmx_pose2d* initial_pose_estimate,
               const float* point_cloud_xyz, int32_t num_points, int32_t device, double* score,
               cmx_pose2d* pose_estimate, cmx_match_stats* stats,
               const uint16_t* device_cells) {
  CMX_REQUIRE(options && limits && (cells || device_cells) && initial_pose_estimate &&
                  point_cloud_xyz,
              "null argument");
  CMX_REQUIRE(pose_estimate != nullptr && score != nullptr,
              "pose_estimate must not be null");            // CHECK at :121
  CMX_REQUIRE(num_points >= 1 && num_points <= (1 << 24), "bad point count");
  CMX_REQUIRE(limits->resolution > 0. && limits->num_x_cells >= 1 && limits->num_y_cells >= 1,
              "bad map limits");
  const bool tsdf = weight_cells != nullptr;
  if (tsdf) CMX_REQUIRE(max_tsd > 0.f && max_weight > 0.f, "bad TSDF ranges");
  const int n = num_points, nx = limits->num_x_cells, ny = limits->num_y_cells;
  const double res = limits->resolution;

  // SearchParameters on the cloud pre-rotated by the initial yaw (:123-130).
  const float ha0 = 0.5f * static_cast<float>(initial_pose_estimate->theta);
  const float q0w = std::cos(ha0), q0z = std::sin(ha0) * 1.f;
  float max_scan_range = 3.f * res;
  for (int i = 0; i < n; ++i) {
    // Same rotation as the device applies (Eigen operation order), f32.
    const float px = point_cloud_xyz[3 * i], py = point_cloud_xyz[3 * i + 1],
                pz = point_cloud_xyz[3 * i + 2];
    const float qx = 0.f, qy = 0.f;
    float uvx = qy * pz - q0z * py, uvy = q0z * px - qx * pz, uvz = qx * py - qy * px;
    uvx += uvx; uvy += uvy; uvz += uvz;
    const float cxx = qy * uvz - q0z * uvy, cyy = q0z * uvx - qx * uvz;
    const float rx = ((px + q0w * uvx) + cxx) + 0.f, ry = ((py + q0w * uvy) + cyy) + 0.f;
    const float range = std::sqrt(rx * rx + ry * ry);
    max_scan_range = std::max(range, max_scan_range);
  }
  const double kSafetyMargin = 1. - 1e-3;
  const float range_sq = max_scan_range * (max_scan_range * 1.f);
  const double step = kSafetyMargin * std::acos(1. - (res * (res * 1.)) / (2. * range_sq));
  const int na = std::ceil(options->angular_search_window / step);
  const int num_scans = 2 * na + 1;
  const int nl = std::ceil(options->linear_search_window / res);
  CMX_REQUIRE(num_scans >= 1 && num_scans < (1 << 16) && nl >= 0 && nl < (1 << 12),
              "unsupported search window");
  const long long side = 2ll * nl + 1;
  const long long num_candidates = side * side * num_scans;
  CMX_REQUIRE(num_candidates < (1ll << 30), "search window too large");
  const int pad = 2 * nl + 1;
  const long long stride = nx + 2ll * pad, rows = ny + 2ll * pad;
  CMX_REQUIRE(stride * rows < (1ll << 27), "grid plus search window too large");
  CMX_REQUIRE(static_cast<long long>(num_scans) * n < (1ll << 30), "too many rotated points");

  WorkspaceLease ws(device);
  // One staging buffer: [xyz | rotations | cells | weight cells] -> one H2D copy.
  const size_t cell_count = static_cast<size_t>(nx) * ny;
  const size_t off_rot = Align16(3 * sizeof(float) * n);
  const size_t off_cells = off_rot + Align16(sizeof(float2) * num_scans);
  const size_t off_weights =
      off_cells + (device_cells ? 0 : Align16(sizeof(uint16_t) * cell_count));
  const size_t in_bytes = off_weights + (tsdf ? Align16(sizeof(uint16_t) * cell_count) : 0);
  char* h_in = ws->pinned[0].ReserveAs<char>(in_bytes);
  char* d_in = ws->dev[0].ReserveAs<char>(in_bytes);
  std::memcpy(h_in, point_cloud_xyz, 3 * sizeof(float) * n);
  float2* h_rot = reinterpret_cast<float2*>(h_in + off_rot);
  double delta_theta = -na * step;
  for (int s = 0; s < num_scans; ++s, delta_theta += step) {
    const float ha = 0.5f * static_cast<float>(delta_theta);
    h_rot[s] = make_float2(std::cos(ha), std::sin(ha) * 1.f);
  }
  if (!device_cells) std::memcpy(h_in + off_cells, cells, sizeof(uint16_t) * cell_count);
  if (tsdf) std::memcpy(h_in + off_weights, weight_cells, sizeof(uint16_t) * cell_count);

  const int n_pad = (n + 63) / 64 * 64;
  int* d_offsets = ws->dev[1].ReserveAs<int>(static_cast<size_t>(num_scans) * n_pad);
  void* d_padded = ws->dev[2].ReserveAs<char>(static_cast<size_t>(stride * rows + 1) *
                                               (tsdf ? sizeof(float2) : sizeof(float)));
  float* d_unweighted = ws->dev[3].ReserveAs<float>(num_candidates);
  float* d_weighted = ws->dev[4].ReserveAs<float>(num_candidates);
  unsigned* d_misc = ws->dev[5].ReserveAs<unsigned>(2 + 2 * kFinalistCap);
  unsigned* h_misc = ws->pinned[1].ReserveAs<unsigned>(2 + 2 * kFinalistCap);

  CMX_HIP(hipMemcpyAsync(d_in, h_in, in_bytes, hipMemcpyHostToDevice, ws->stream));

  Rt2DParams P;
  P.cells = device_cells ? device_cells : reinterpret_cast<const uint16_t*>(d_in + off_cells);
  P.weights = tsdf ? reinterpret_cast<const uint16_t*>(d_in + off_weights) : nullptr;
  P.nx = nx; P.ny = ny;
  P.res = res; P.max_x = limits->max_x; P.max_y = limits->max_y;
  P.tx = static_cast<float>(initial_pose_estimate->x);
  P.ty = static_cast<float>(initial_pose_estimate->y);
  P.init_qw = q0w; P.init_qz = q0z;
  P.nl = nl; P.num_scans = num_scans; P.num_angular = na;
  P.step = step;
  P.wt = options->translation_delta_cost_weight;
  P.wr = options->rotation_delta_cost_weight;
  P.max_tsd = max_tsd; P.max_weight = max_weight;
  P.scan_rot = reinterpret_cast<const float2*>(d_in + off_rot);
  P.offsets = d_offsets;
  P.n_pad = n_pad;
  P.padded = d_padded;
  P.pad = pad; P.stride = static_cast<int>(stride); P.rows = static_cast<int>(rows);
  P.misc = d_misc;
  const float* d_xyz = reinterpret_cast<const float*>(d_in);

  CMX_HIP(hipEventRecord(ws->ev_begin, ws->stream));
  const int prep_blocks = num_scans + static_cast<int>(DivUp(stride * rows, 1024));
  const dim3 score_grid(static_cast<unsigned>(DivUp(side * side, 64)), num_scans);
  if (tsdf) {
    Rt2DPrepKernel<true><<<prep_blocks, 256, 0, ws->stream>>>(P, d_xyz, n);
    CMX_HIP(hipEventRecord(ws->ev_k0, ws->stream));
    Rt2DScoreKernel<true><<<score_grid, 64, 0, ws->stream>>>(P, n, d_unweighted, d_weighted);
  } else {
    Rt2DPrepKernel<false><<<prep_blocks, 256, 0, ws->stream>>>(P, d_xyz, n);
    CMX_HIP(hipEventRecord(ws->ev_k0, ws->stream));
    Rt2DScoreKernel<false><<<score_grid, 64, 0, ws->stream>>>(P, n, d_unweighted, d_weighted);
  }
  CMX_HIP(hipEventRecord(ws->ev_k1, ws->stream));
  Rt2DCollectKernel<<<DivUp(num_candidates, 256), 256, 0, ws->stream>>>(
      d_weighted, d_unweighted, static_cast<int>(num_candidates), d_misc, kFinalistCap);
  CMX_HIP(hipGetLastError());
  CMX_HIP(hipEventRecord(ws->ev_end, ws->stream));
  CMX_HIP(hipMemcpyAsync(h_misc, d_misc, sizeof(unsigned) * (2 + 2 * kFinalistHead),
                         hipMemcpyDeviceToHost, ws->stream));
  CMX_HIP(hipStreamSynchronize(ws->stream));

  // Exact weighting + first-maximum on the finalists (:142-143,170-174).
  std::vector<std::pair<int, float>> finalists;
  const long long count = h_misc[1];
  if (count <= kFinalistCap) {
    if (count > kFinalistHead) {
      CMX_HIP(hipMemcpyAsync(h_misc, d_misc, sizeof(unsigned) * (2 + 2 * count),
                             hipMemcpyDeviceToHost, ws->stream));
      CMX_HIP(hipStreamSynchronize(ws->stream));
    }
    finalists.resize(count);
    for (long long i = 0; i < count; ++i) {
      float v;
      std::memcpy(&v, &h_misc[3 + 2 * i], sizeof(float));
      finalists[i] = {static_cast<int>(h_misc[2 + 2 * i]), v};
    }
    std::sort(finalists.begin(), finalists.end());
  } else {  // flat score landscape: take everything
    std::vector<float> all(num_candidates);
    CMX_HIP(hipMemcpyAsync(all.data(), d_unweighted, sizeof(float) * num_candidates,
                           hipMemcpyDeviceToHost, ws->stream));
    CMX_HIP(hipStreamSynchronize(ws->stream));
    finalists.resize(num_candidates);
    for (long long c = 0; c < num_candidates; ++c) finalists[c] = {static_cast<int>(c), all[c]};
  }
  CMX_REQUIRE(!finalists.empty(), "internal error: no candidate collected");
  const int side_i = static_cast<int>(side);
  float best_score = -1.f;
  int best = -1;
  for (const auto& f : finalists) {
    const int c = f.first;
    const int s = c / (side_i * side_i);
    const int rem = c - s * side_i * side_i;
    const int dx = rem / side_i - nl, dy = rem % side_i - nl;
    const double cx = -dy * res, cy = -dx * res;
    const double theta = (s - na) * step;
    const double t = std::hypot(cx, cy) * P.wt + std::abs(theta) * P.wr;
    float sc = f.second;
    sc *= std::exp(-(t * (t * 1.)));
    if (sc > best_score) { best_score = sc; best = c; }   // finalists ascend: first max wins
  }
  // CHECK_GT(score, 0) in the probability branch (:73); a TSDF may score 0 everywhere
  // (CHECK_GE at :56), in which case the first candidate wins like std::max_element.
  {
    const int s = best / (side_i * side_i);
    const int rem = best - s * side_i * side_i;
    const int dx = rem / side_i - nl, dy = rem % side_i - nl;
    pose_estimate->x = initial_pose_estimate->x + (-dy * res);
    pose_estimate->y = initial_pose_estimate->y + (-dx * res);
    pose_estimate->theta = initial_pose_estimate->theta + (s - na) * step;
    *score = best_score;
  }
  if (stats) {
    cmx_match_stats st{};
    st.candidates_scored = num_candidates;
    st.coarse_candidates = num_candidates;
    st.num_scans = num_scans;
    float ms = 0.f;
    CMX_HIP(hipEventElapsedTime(&ms, ws->ev_begin, ws->ev_end));
    st.device_ms = ms;
    CMX_HIP(hipEventElapsedTime(&ms, ws->ev_k0, ws->ev_k1));
    st.dominant_kernel_ms = ms;
    *stats = st;
  }
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
