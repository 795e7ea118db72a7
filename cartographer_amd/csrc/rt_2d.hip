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

#include "scan_matching_2d.h"

namespace cmx {
namespace {

struct Rt2DParams {
  const uint16_t* cells;   // device grid
  int nx, ny;
  double res, max_x, max_y;
  float tx, ty, init_qw, init_qz;
  int nl, num_scans, num_angular;
  double step, wt, wr;
  const float2* scan_rot;
  int2* discrete;          // [num_scans][n]
};

__global__ void __launch_bounds__(256)
Rt2DPrepKernel(Rt2DParams P, const float* __restrict__ xyz, int n) {
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
    const int ix = LRoundF64((P.max_y - static_cast<double>(y)) / P.res - 0.5);
    const int iy = LRoundF64((P.max_x - static_cast<double>(x)) / P.res - 0.5);
    P.discrete[static_cast<size_t>(s) * n + i] = make_int2(ix, iy);
  }
}

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

__global__ void __launch_bounds__(64)
Rt2DScoreKernel(Rt2DParams P, int n, int num_candidates, float* __restrict__ unweighted,
                float* __restrict__ weighted, unsigned* __restrict__ max_bits) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  float w = 0.f;
  if (c < num_candidates) {
    const int side = 2 * P.nl + 1;
    const int s = c / (side * side);
    const int rem = c - s * side * side;
    const int dx = rem / side - P.nl, dy = rem % side - P.nl;   // x outer, y inner (:99-113)
    const int2* scan = P.discrete + static_cast<size_t>(s) * n;
    float acc = 0.f;
#pragma unroll 8
    for (int i = 0; i < n; ++i) {
      const int2 p = scan[i];
      const int x = p.x + dx, y = p.y + dy;
      const bool inside = static_cast<unsigned>(x) < static_cast<unsigned>(P.nx) &&
                          static_cast<unsigned>(y) < static_cast<unsigned>(P.ny);
      // Unconditional load from a clamped offset so the loop's loads pipeline.
      const unsigned raw = P.cells[inside ? P.nx * y + x : 0];
      const float prob = inside ? CellProbability(raw) : 0.1f;  // kMinProbability outside
      acc += prob;
    }
    acc /= static_cast<float>(n);
    unweighted[c] = acc;
    const double cx = -dy * P.res, cy = -dx * P.res;
    const double theta = (s - P.num_angular) * P.step;
    const double t = hypot(cx, cy) * P.wt + fabs(theta) * P.wr;
    w = static_cast<float>(static_cast<double>(acc) * exp(-(t * t)));
    weighted[c] = w;
  }
  unsigned bits = __float_as_uint(w);   // scores are > 0
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) bits = max(bits, __shfl_xor(bits, off, 64));
  if (threadIdx.x == 0) atomicMax(max_bits, bits);
}

__global__ void Rt2DCollectKernel(const float* __restrict__ weighted, int num_candidates,
                                  const unsigned* __restrict__ max_bits, int* __restrict__ count,
                                  int* __restrict__ finalists, int capacity) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= num_candidates) return;
  const float threshold = __uint_as_float(*max_bits) * (1.f - 1e-5f);
  if (weighted[c] >= threshold) {
    const int slot = atomicAdd(count, 1);
    if (slot < capacity) finalists[slot] = c;
  }
}

}  // namespace
}  // namespace cmx

extern "C" cmx_status cmx_rt2d_match(const cmx_rt_options* options,
                                     const cmx_grid2d_limits* limits, const uint16_t* cells,
                                     const cmx_pose2d* initial_pose_estimate,
                                     const float* point_cloud_xyz, int32_t num_points,
                                     int32_t device, double* score, cmx_pose2d* pose_estimate,
                                     cmx_match_stats* stats) {
  using namespace cmx;
  return Guard([&] {
    CMX_REQUIRE(options && limits && cells && initial_pose_estimate && point_cloud_xyz,
                "null argument");
    CMX_REQUIRE(pose_estimate != nullptr && score != nullptr,
                "pose_estimate must not be null");            // CHECK at :121
    CMX_REQUIRE(num_points >= 1 && num_points <= (1 << 24), "bad point count");
    CMX_REQUIRE(limits->resolution > 0. && limits->num_x_cells >= 1 && limits->num_y_cells >= 1,
                "bad map limits");
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
    const double step =
        kSafetyMargin * std::acos(1. - (res * (res * 1.)) / (2. * range_sq));
    const int na = std::ceil(options->angular_search_window / step);
    const int num_scans = 2 * na + 1;
    const int nl = std::ceil(options->linear_search_window / res);
    CMX_REQUIRE(num_scans >= 1 && num_scans < (1 << 20) && nl >= 0 && nl < (1 << 14),
                "unsupported search window");
    const long long side = 2ll * nl + 1;
    const long long num_candidates = side * side * num_scans;
    CMX_REQUIRE(num_candidates < (1ll << 30), "search window too large");

    WorkspaceLease ws(device);
    float* d_xyz = ws->dev[0].ReserveAs<float>(3 * static_cast<size_t>(n));
    uint16_t* d_cells = ws->dev[1].ReserveAs<uint16_t>(static_cast<size_t>(nx) * ny);
    float2* d_rot = ws->dev[2].ReserveAs<float2>(num_scans);
    int2* d_discrete = ws->dev[3].ReserveAs<int2>(static_cast<size_t>(num_scans) * n);
    float* d_unweighted = ws->dev[4].ReserveAs<float>(num_candidates);
    float* d_weighted = ws->dev[5].ReserveAs<float>(num_candidates);
    const int kFinalistCap = 4096;
    int* d_misc = ws->dev[6].ReserveAs<int>(2 + kFinalistCap);
    float2* h_rot = ws->pinned[0].ReserveAs<float2>(num_scans);
    int* h_misc = ws->pinned[1].ReserveAs<int>(2 + kFinalistCap);

    double delta_theta = -na * step;
    for (int s = 0; s < num_scans; ++s, delta_theta += step) {
      const float ha = 0.5f * static_cast<float>(delta_theta);
      h_rot[s] = make_float2(std::cos(ha), std::sin(ha) * 1.f);
    }
    CMX_HIP(hipMemcpyAsync(d_xyz, point_cloud_xyz, 3 * sizeof(float) * n, hipMemcpyHostToDevice,
                           ws->stream));
    CMX_HIP(hipMemcpyAsync(d_cells, cells, sizeof(uint16_t) * nx * ny, hipMemcpyHostToDevice,
                           ws->stream));
    CMX_HIP(hipMemcpyAsync(d_rot, h_rot, sizeof(float2) * num_scans, hipMemcpyHostToDevice,
                           ws->stream));
    CMX_HIP(hipMemsetAsync(d_misc, 0, 2 * sizeof(int), ws->stream));

    Rt2DParams P;
    P.cells = d_cells; P.nx = nx; P.ny = ny;
    P.res = res; P.max_x = limits->max_x; P.max_y = limits->max_y;
    P.tx = static_cast<float>(initial_pose_estimate->x);
    P.ty = static_cast<float>(initial_pose_estimate->y);
    P.init_qw = q0w; P.init_qz = q0z;
    P.nl = nl; P.num_scans = num_scans; P.num_angular = na;
    P.step = step;
    P.wt = options->translation_delta_cost_weight;
    P.wr = options->rotation_delta_cost_weight;
    P.scan_rot = d_rot;
    P.discrete = d_discrete;

    CMX_HIP(hipEventRecord(ws->ev_begin, ws->stream));
    Rt2DPrepKernel<<<num_scans, 256, 0, ws->stream>>>(P, d_xyz, n);
    unsigned* d_max = reinterpret_cast<unsigned*>(d_misc);
    int* d_count = d_misc + 1;
    int* d_finalists = d_misc + 2;
    CMX_HIP(hipEventRecord(ws->ev_k0, ws->stream));
    Rt2DScoreKernel<<<DivUp(num_candidates, 64), 64, 0, ws->stream>>>(
        P, n, static_cast<int>(num_candidates), d_unweighted, d_weighted, d_max);
    CMX_HIP(hipEventRecord(ws->ev_k1, ws->stream));
    Rt2DCollectKernel<<<DivUp(num_candidates, 256), 256, 0, ws->stream>>>(
        d_weighted, static_cast<int>(num_candidates), d_max, d_count, d_finalists, kFinalistCap);
    CMX_HIP(hipGetLastError());
    CMX_HIP(hipEventRecord(ws->ev_end, ws->stream));
    CMX_HIP(hipMemcpyAsync(h_misc, d_misc, sizeof(int) * (2 + kFinalistCap),
                           hipMemcpyDeviceToHost, ws->stream));
    CMX_HIP(hipStreamSynchronize(ws->stream));

    // Exact weighting + first-maximum on the finalists (:142-143,170-174).
    std::vector<int> finalists;
    std::vector<float> acc;
    const int count = h_misc[1];
    if (count <= kFinalistCap) {
      finalists.assign(h_misc + 2, h_misc + 2 + count);
      std::sort(finalists.begin(), finalists.end());
      acc.resize(count);
      for (int i = 0; i < count; ++i)
        CMX_HIP(hipMemcpy(&acc[i], d_unweighted + finalists[i], sizeof(float),
                          hipMemcpyDeviceToHost));
    } else {  // flat score landscape: take everything
      finalists.resize(num_candidates);
      for (long long c = 0; c < num_candidates; ++c) finalists[c] = static_cast<int>(c);
      acc.resize(num_candidates);
      CMX_HIP(hipMemcpy(acc.data(), d_unweighted, sizeof(float) * num_candidates,
                        hipMemcpyDeviceToHost));
    }
    CMX_REQUIRE(!finalists.empty(), "internal error: no candidate collected");
    float best_score = -1.f;
    int best = -1;
    for (size_t i = 0; i < finalists.size(); ++i) {
      const int c = finalists[i];
      const int s = c / static_cast<int>(side * side);
      const int rem = c - s * static_cast<int>(side * side);
      const int dx = rem / static_cast<int>(side) - nl, dy = rem % static_cast<int>(side) - nl;
      const double cx = -dy * res, cy = -dx * res;
      const double theta = (s - na) * step;
      const double t = std::hypot(cx, cy) * P.wt + std::abs(theta) * P.wr;
      float sc = acc[i];
      sc *= std::exp(-(t * (t * 1.)));
      if (sc > best_score) { best_score = sc; best = c; }   // finalists ascend: first max wins
    }
    {
      const int s = best / static_cast<int>(side * side);
      const int rem = best - s * static_cast<int>(side * side);
      const int dx = rem / static_cast<int>(side) - nl, dy = rem % static_cast<int>(side) - nl;
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
  });
}
