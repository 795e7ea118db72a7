// FastCorrelativeScanMatcher3D on gfx950.
//
// Reference behaviour being replaced (SM3 = cartographer/mapping/internal/3d/scan_matching):
//   SM3/precomputation_grid_3d.cc:49-81              ConvertToPrecomputationGrid / PrecomputeGrid
//   SM3/fast_correlative_scan_matcher_3d.cc:57-77    PrecomputationGridStack3D
//   SM3/fast_correlative_scan_matcher_3d.cc:127-198  Match / MatchFullSubmap / MatchWithSearchParameters
//   SM3/fast_correlative_scan_matcher_3d.cc:200-295  DiscretizeScan / GenerateDiscreteScans
//   SM3/fast_correlative_scan_matcher_3d.cc:297-440  candidates, ScoreCandidates, BranchAndBound
//   SM3/rotational_scan_matcher.cc:121-189           histogram yaw pre-filter (host: A x bins work)
//   SM3/low_resolution_matcher.cc:23-35              leaf verification
//
// Host / device split.  Everything that involves libm (acos, sin, cos, atan2)
// or Eigen-ordered quaternion algebra on a handful of values runs on the host
// with the reference's operation order; the device does the N-point work:
// discretising every surviving yaw, scoring every lowest-resolution candidate,
// and the branch and bound (one wave per node, eight children scored per
// point), including the low-resolution verification of leaves.
#include <atomic>
#include <string>
#include <thread>
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "scan_matching_3d.h"

namespace cmx {
namespace {

constexpr int kSubLists3 = 64;
constexpr int kSeeds3 = 64;

struct Node3D {
  int level;               // depth of this node (0 = leaf)
  int scan;
  int ox, oy, oz;          // Candidate3D::offset
  float score;
  float coarse_score;      // score of the lowest-resolution ancestor
  int coarse_index;        // its generation index
  unsigned long long path; // sibling ranks along the descent, 3 bits per level
  float low_resolution_score;
  int pad;
};

struct Counters3 {
  int frontier[kMaxDepth + 2][kSubLists3];
  int dive[2][kSubLists3];
  int leaves[kSubLists3];
  int overflow;
  int pad0;
  unsigned best_bits;      // float bits of the best verified leaf (>= min_score floor)
  int pad;
  unsigned long long scored[16];
  unsigned long long expanded[16];
};

struct List3 {
  Node3D* nodes;
  int* counts;
  int sub_capacity;
};

struct Fast3DProblem {
  Brick level[kMaxDepth];
  int depth, full_resolution_depth;
  Brick low;
  float low_resolution, resolution;
  int wxy, wz;
  int num_scans, n, n_low;
  const int4* cells;        // [num_scans][n] full-resolution cell indices
  const float* low_xyz;     // low-resolution cloud
  const float4* scan_q;     // [num_scans] rotation of GetPoseFromCandidate (x,y,z,w)
  float pose_tx, pose_ty, pose_tz;
  float min_score;
  double min_low_resolution_score;
  int ncx, ncy, ncz;        // lowest-resolution candidates per scan and axis
  float* coarse_score;      // [num_scans * ncx*ncy*ncz]
};

// ---------------------------------------------------------------------------
// Precomputation stack (gather form of PrecomputeGrid's scatter-max)
// ---------------------------------------------------------------------------
__global__ void PrecomputeLevel3DKernel(Brick prev, Brick out, int shift, int half) {
  const long long total = static_cast<long long>(out.nx) * out.ny * out.nz;
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int ix = static_cast<int>(i % out.nx);
  const int iy = static_cast<int>((i / out.nx) % out.ny);
  const int iz = static_cast<int>(i / (static_cast<long long>(out.nx) * out.ny));
  const int tx = ix + out.lo_x, ty = iy + out.lo_y, tz = iz + out.lo_z;
  unsigned best = 0;
  const int sub = half ? 2 : 1;
  // out(t) = max over octants o and (for half resolution) sub-cells e of
  // prev(sub*t + e + shift*o)   <=>   t = (c - shift*o) >> (half ? 1 : 0).
  for (int oz = 0; oz < 2; ++oz)
    for (int oy = 0; oy < 2; ++oy)
      for (int ox = 0; ox < 2; ++ox)
        for (int ez = 0; ez < sub; ++ez)
          for (int ey = 0; ey < sub; ++ey)
            for (int ex = 0; ex < sub; ++ex)
              best = max(best, BrickValueU8(prev, sub * tx + ex + shift * ox,
                                            sub * ty + ey + shift * oy,
                                            sub * tz + ez + shift * oz));
  static_cast<uint8_t*>(const_cast<void*>(out.cells))[i] = static_cast<uint8_t>(best);
}

// ---------------------------------------------------------------------------
// Scan discretisation (DiscretizeScan, :200-244: transform + GetCellIndex)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
Discretize3DKernel(const float* __restrict__ xyz, int n, const float4* __restrict__ pose_q,
                   float tx, float ty, float tz, float resolution, int4* __restrict__ cells) {
  const int s = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 q4 = pose_q[s];
  const Quat q{q4.w, q4.x, q4.y, q4.z};
  const F3 p{xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
  const F3 r = Rotate(q, p);
  const F3 t{r.x + tx, r.y + ty, r.z + tz};
  const int3 c = CellIndex3(t, resolution);
  cells[static_cast<size_t>(s) * n + i] = make_int4(c.x, c.y, c.z, 0);
}

// ---------------------------------------------------------------------------
// Scoring
// ---------------------------------------------------------------------------
__device__ __forceinline__ float ToProbability(int sum, int n) {
  // PrecomputationGrid3D::ToProbability(sum / float(N))  (:347-350)
  const float kMinP = 0.1f;
  const float kMaxP = 1.f - kMinP;
  return kMinP + (static_cast<float>(sum) / static_cast<float>(n)) * ((kMaxP - kMinP) / 255.f);
}

// Cell index of point `c` at `depth` (DiscretizeScan's low-resolution
// indices, :223-241) — e = max(0, depth - full_resolution_depth + 1).
__device__ __forceinline__ int3 DepthIndex(const int4& c, int e, int sx, int sy, int sz) {
  if (e == 0) return make_int3(c.x, c.y, c.z);
  return make_int3(((c.x + sx) >> e) - (sx >> e), ((c.y + sy) >> e) - (sy >> e),
                   ((c.z + sz) >> e) - (sz >> e));
}

// Integer sum of one candidate, one wave (ScoreCandidates, :332-355).
__device__ __forceinline__ int ScoreCandidate3D(const Fast3DProblem& P, int depth, int scan,
                                                int ox, int oy, int oz, int lane) {
  const int e = max(0, depth - P.full_resolution_depth + 1);
  const Brick& L = P.level[depth];
  const int4* __restrict__ cells = P.cells + static_cast<size_t>(scan) * P.n;
  const int fx = ox >> e, fy = oy >> e, fz = oz >> e;
  int sum = 0;
#pragma unroll 4
  for (int i = lane; i < P.n; i += kWave) {
    const int3 d = DepthIndex(cells[i], e, -P.wxy, -P.wxy, -P.wz);
    sum += BrickValueU8(L, d.x + fx, d.y + fy, d.z + fz);
  }
  return WaveSum(sum);
}

__global__ void __launch_bounds__(256)
ScoreCoarse3DKernel(Fast3DProblem P) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int per_scan = P.ncx * P.ncy * P.ncz;
  const int total = per_scan * P.num_scans;
  const int step = 1 << (P.depth - 1);
  for (int c = blockIdx.x * 4 + wave; c < total; c += gridDim.x * 4) {
    const int s = c / per_scan;
    int r = c - s * per_scan;
    // z outer, y, x inner (:313-326)
    const int iz = r / (P.ncy * P.ncx);
    r -= iz * P.ncy * P.ncx;
    const int iy = r / P.ncx, ix = r - iy * P.ncx;
    const int sum = ScoreCandidate3D(P, P.depth - 1, s, -P.wxy + ix * step, -P.wxy + iy * step,
                                     -P.wz + iz * step, lane);
    if (lane == 0) P.coarse_score[c] = ToProbability(sum, P.n);
  }
}

// Few lowest-resolution candidates (deep stacks: one per yaw): a whole block per
// candidate, so that its sum is not a 43-iteration chain of one wavefront.
__global__ void __launch_bounds__(256)
ScoreCoarse3DBlockKernel(Fast3DProblem P) {
  __shared__ int partial[4];
  const int per_scan = P.ncx * P.ncy * P.ncz;
  const int total = per_scan * P.num_scans;
  const int step = 1 << (P.depth - 1);
  const int depth = P.depth - 1;
  const int e = max(0, depth - P.full_resolution_depth + 1);
  const Brick L = P.level[depth];
  for (int c = blockIdx.x; c < total; c += gridDim.x) {
    const int s = c / per_scan;
    int r = c - s * per_scan;
    const int iz = r / (P.ncy * P.ncx);
    r -= iz * P.ncy * P.ncx;
    const int iy = r / P.ncx, ix = r - iy * P.ncx;
    const int fx = (-P.wxy + ix * step) >> e, fy = (-P.wxy + iy * step) >> e,
              fz = (-P.wz + iz * step) >> e;
    const int4* __restrict__ cells = P.cells + static_cast<size_t>(s) * P.n;
    int sum = 0;
#pragma unroll 4
    for (int i = threadIdx.x; i < P.n; i += 256) {
      const int3 d = DepthIndex(cells[i], e, -P.wxy, -P.wxy, -P.wz);
      sum += BrickValueU8(L, d.x + fx, d.y + fy, d.z + fz);
    }
    sum = WaveSum(sum);
    if ((threadIdx.x & 63) == 0) partial[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0)
      P.coarse_score[c] = ToProbability(partial[0] + partial[1] + partial[2] + partial[3], P.n);
    __syncthreads();
  }
}

__device__ __forceinline__ Node3D CoarseNode3D(const Fast3DProblem& P, int c) {
  const int per_scan = P.ncx * P.ncy * P.ncz;
  const int step = 1 << (P.depth - 1);
  const int s = c / per_scan;
  int r = c - s * per_scan;
  const int iz = r / (P.ncy * P.ncx);
  r -= iz * P.ncy * P.ncx;
  const int iy = r / P.ncx, ix = r - iy * P.ncx;
  Node3D nd;
  nd.level = P.depth - 1;
  nd.scan = s;
  nd.ox = -P.wxy + ix * step; nd.oy = -P.wxy + iy * step; nd.oz = -P.wz + iz * step;
  nd.score = P.coarse_score[c];
  nd.coarse_score = nd.score;
  nd.coarse_index = c;
  nd.path = 0;
  nd.low_resolution_score = 0.f;
  nd.pad = 0;
  return nd;
}

__device__ __forceinline__ bool Push3(const List3& list, int sub, int slot, const Node3D& nd) {
  if (slot >= list.sub_capacity) return false;
  list.nodes[static_cast<size_t>(sub) * list.sub_capacity + slot] = nd;
  return true;
}
__device__ __forceinline__ int ListMax3(const List3& list) {
  return WaveMax(min(list.counts[threadIdx.x & 63], list.sub_capacity));
}

// Seeds of the dive: the ~64 best lowest-resolution candidates (histogram
// threshold on the scores).
__global__ void __launch_bounds__(1024)
SeedSelect3DKernel(Fast3DProblem P, List3 seeds, Counters3* __restrict__ counters) {
  __shared__ int hist[1024];
  __shared__ int threshold_bin;
  hist[threadIdx.x] = 0;
  __syncthreads();
  const int total = P.ncx * P.ncy * P.ncz * P.num_scans;
  auto bin_of = [](float score) {   // scores lie in [0.1, 0.9]
    return min(1023, max(0, static_cast<int>((score - 0.1f) * (1023.f / 0.8f))));
  };
  for (int c = threadIdx.x; c < total; c += blockDim.x) atomicAdd(&hist[bin_of(P.coarse_score[c])], 1);
  __syncthreads();
  // threshold_bin = the largest b >= 1 with sum_{j >= b} hist[j] >= kSeeds3, else 0 -- found
  // by one wave (16 bins per lane, suffix sums across lanes) instead of a 1023-step serial
  // walk by one thread (51 us when there are fewer than kSeeds3 candidates).
  if (threadIdx.x < 64) {
    const int l = threadIdx.x;
    int mine = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) mine += hist[16 * l + k];
    int suffix = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int o = __shfl_down(suffix, off, 64);
      if (l + off < 64) suffix += o;
    }
    const unsigned long long reach = __ballot(suffix >= kSeeds3);
    if (reach == 0) {
      if (l == 0) threshold_bin = 0;
    } else {
      const int owner = 63 - __clzll(reach);          // highest lane whose suffix reaches it
      const int above = suffix - mine;                 // bins of the lanes above
      if (l == owner) {
        int acc = above, b = 16 * l + 15;
        for (; b > 16 * l; --b) {
          acc += hist[b];
          if (acc >= kSeeds3) break;
        }
        threshold_bin = b;     // b == 16 l: reached with the lane's lowest bin (0 only for l == 0)
      }
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < total; c += blockDim.x) {
    const float sc = P.coarse_score[c];
    if (bin_of(sc) >= threshold_bin && sc > P.min_score) {
      const int slot = atomicAdd(&seeds.counts[0], 1);
      if (slot < kSeeds3) Push3(seeds, 0, slot, CoarseNode3D(P, c));
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) seeds.counts[0] = min(seeds.counts[0], kSeeds3);   // only kSeeds3 stored
}

// Lowest-resolution nodes that can still matter (reference: :405-408).
__global__ void __launch_bounds__(256)
Filter3DKernel(Fast3DProblem P, int strict, int chunk, int num_chunks, List3 out,
               Counters3* __restrict__ counters) {
  const int total = P.ncx * P.ncy * P.ncz * P.num_scans;
  const float best = __uint_as_float(counters->best_bits);
  const int sub = blockIdx.x & (kSubLists3 - 1);
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < total; c += gridDim.x * blockDim.x) {
    if (c % num_chunks != chunk) continue;
    const float sc = P.coarse_score[c];
    if (strict ? (sc > best) : (sc >= best)) {
      if (!Push3(out, sub, atomicAdd(&out.counts[sub], 1), CoarseNode3D(P, c)))
        counters->overflow = 1;
    }
  }
}

// CreateLowResolutionMatcher's lambda (SM3/low_resolution_matcher.cc:23-35) for the pose
// of one leaf, by a whole block: the per-point probabilities are computed in parallel
// into LDS, then summed sequentially in point order (as the reference does) by every
// thread from LDS broadcasts.
constexpr int kLowChunk = 2048;

__device__ __forceinline__ float LowResolutionScore(const Fast3DProblem& P, const Quat& q, float tx,
                                                    float ty, float tz, float* prob /*[kLowChunk]*/) {
  float acc = 0.f;
  for (int base = 0; base < P.n_low; base += kLowChunk) {
    const int cnt = min(kLowChunk, P.n_low - base);
    __syncthreads();                                   // previous chunk consumed
    for (int i = threadIdx.x; i < cnt; i += blockDim.x) {
      const float* xyz = P.low_xyz + 3 * static_cast<size_t>(base + i);
      const F3 r = Rotate(q, F3{xyz[0], xyz[1], xyz[2]});
      const F3 t{r.x + tx, r.y + ty, r.z + tz};
      const int3 c = CellIndex3(t, P.low_resolution);
      prob[i] = ValueToProbabilityDev(BrickValueU16(P.low, c.x, c.y, c.z));
    }
    __syncthreads();
#pragma unroll 8
    for (int i = 0; i < cnt; ++i) acc += prob[i];
  }
  return acc / static_cast<float>(P.n_low);
}

// One 256-thread block per node: scores the <=8 children (z outer, y, x inner with the
// `break`s of :416-431), ranks them as the reference's stable descending sort does, then
//   child depth > 0, full: children that can still matter go to `out`;
//   child depth > 0, dive: only the best child continues;
//   child depth == 0: leaves are verified in rank order with the low-resolution
//     matcher; the first one that passes is recorded (:389-402).
// A search expands a few thousand nodes in total, so what matters is the latency of
// one expansion: four waves share the points, the eight child cells of a point are
// addressed from per-axis clamped offsets (two positions per axis), and the loads of
// several points are in flight together.
struct ExpandShared {
  int partial[4][8];
  float score[8];
  float low_prob[kLowChunk];
  Node3D next;       // dive: the child the descent continues with
  int has_next;
};

// Expansion of one node by the whole block (see above).  dive = 0: children that can still
// matter are appended to `out`; dive = 1: the best child is left in sh->next.
__device__ __forceinline__ void ExpandNode3D(const Fast3DProblem& P, const Node3D& nd, float best,
                                             int dive, int strict, const List3& out,
                                             const List3& leaves,
                                             Counters3* __restrict__ counters, int sub_id,
                                             ExpandShared* sh) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) sh->has_next = 0;
  {
    const int child_depth = nd.level - 1;
    const int half = 1 << child_depth;
    const int e = max(0, child_depth - P.full_resolution_depth + 1);
    const Brick L = P.level[child_depth];
    const uint8_t* __restrict__ cells8 = static_cast<const uint8_t*>(L.cells);
    const int4* __restrict__ cells = P.cells + static_cast<size_t>(nd.scan) * P.n;
    const bool vx = nd.ox + half <= P.wxy, vy = nd.oy + half <= P.wxy, vz = nd.oz + half <= P.wz;
    // Shifted offsets of the 2 positions per axis, relative to the brick origin.
    const int fx[2] = {(nd.ox >> e) - L.lo_x, ((nd.ox + half) >> e) - L.lo_x};
    const int fy[2] = {(nd.oy >> e) - L.lo_y, ((nd.oy + half) >> e) - L.lo_y};
    const int fz[2] = {(nd.oz >> e) - L.lo_z, ((nd.oz + half) >> e) - L.lo_z};
    const int row = L.nx, slab = L.nx * L.ny;
    int sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 2
    for (int q = threadIdx.x; q < P.n; q += 256) {
      const int3 d = DepthIndex(cells[q], e, -P.wxy, -P.wxy, -P.wz);
      // Per axis and position: in-range flag and (clamped) address term.
      int ax[2], ay[2], az[2];
      bool okx[2], oky[2], okz[2];
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int ix = d.x + fx[b], iy = d.y + fy[b], iz = d.z + fz[b];
        okx[b] = static_cast<unsigned>(ix) < static_cast<unsigned>(L.nx);
        oky[b] = static_cast<unsigned>(iy) < static_cast<unsigned>(L.ny);
        okz[b] = static_cast<unsigned>(iz) < static_cast<unsigned>(L.nz);
        ax[b] = okx[b] ? ix : 0;
        ay[b] = oky[b] ? iy * row : 0;
        az[b] = okz[b] ? iz * slab : 0;
      }
      unsigned v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k)    // unconditional loads from in-range addresses
        v[k] = cells8[az[(k >> 2) & 1] + ay[(k >> 1) & 1] + ax[k & 1]];
#pragma unroll
      for (int k = 0; k < 8; ++k)
        sum[k] += (okx[k & 1] && oky[(k >> 1) & 1] && okz[(k >> 2) & 1]) ? v[k] : 0u;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int total = WaveSum(sum[k]);
      if (lane == 0) sh->partial[wave][k] = total;
    }
    __syncthreads();
    if (threadIdx.x < 8) {
      const int k = threadIdx.x;
      const bool valid = (!(k & 1) || vx) && (!(k & 2) || vy) && (!(k & 4) || vz);
      const int total =
          sh->partial[0][k] + sh->partial[1][k] + sh->partial[2][k] + sh->partial[3][k];
      sh->score[k] = valid ? ToProbability(total, P.n) : -1.f;
    }
    __syncthreads();
    float score[8];
    int nvalid = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      score[k] = sh->score[k];
      nvalid += score[k] >= 0.f;
    }
    int rank[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      int r = 0;
#pragma unroll
      for (int o = 0; o < 8; ++o)
        if (o != k && score[o] >= 0.f && (score[o] > score[k] || (score[o] == score[k] && o < k)))
          ++r;
      rank[k] = r;
    }
    if (threadIdx.x == 0) {
      atomicAdd(&counters->scored[sub_id & 15], static_cast<unsigned long long>(nvalid));
      atomicAdd(&counters->expanded[sub_id & 15], 1ull);
    }
    auto make_child = [&](int k) {
      Node3D child = nd;
      child.level = child_depth;
      child.ox = nd.ox + ((k & 1) ? half : 0);
      child.oy = nd.oy + ((k & 2) ? half : 0);
      child.oz = nd.oz + ((k & 4) ? half : 0);
      child.score = score[k];
      child.path = nd.path | (static_cast<unsigned long long>(rank[k]) << (3 * child_depth));
      return child;
    };
    if (child_depth == 0) {
      // Leaves in descending order; the first that passes the low-resolution
      // matcher is the result of this sibling group.  (Everything below is
      // block-uniform, so the barriers inside LowResolutionScore are safe.)
      for (int r = 0; r < nvalid; ++r) {
        int k = 0;
#pragma unroll
        for (int o = 0; o < 8; ++o)
          if (score[o] >= 0.f && rank[o] == r) k = o;
        const float sc = score[k];
        __syncthreads();
        if (threadIdx.x == 0)
          sh->score[0] = __uint_as_float(__hip_atomic_load(&counters->best_bits, __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_AGENT));
        __syncthreads();
        const float now = sh->score[0];
        if (!(sc > P.min_score) || (strict ? !(sc > now) : (sc < now))) break;
        const Node3D leaf = make_child(k);
        const float4 q4 = P.scan_q[nd.scan];
        const float low = LowResolutionScore(
            P, Quat{q4.w, q4.x, q4.y, q4.z},
            (P.pose_tx + 0.f) + P.resolution * static_cast<float>(leaf.ox),
            (P.pose_ty + 0.f) + P.resolution * static_cast<float>(leaf.oy),
            (P.pose_tz + 0.f) + P.resolution * static_cast<float>(leaf.oz), sh->low_prob);
        if (static_cast<double>(low) >= P.min_low_resolution_score) {
          if (threadIdx.x == 0) {
            Node3D rec = leaf;
            rec.low_resolution_score = low;
            if (!Push3(leaves, sub_id, atomicAdd(&leaves.counts[sub_id], 1), rec))
              counters->overflow = 1;
            atomicMax(&counters->best_bits, __float_as_uint(sc));
          }
          break;
        }
      }
    } else if (threadIdx.x == 0) {
      int keep_mask = 0, m = 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (score[k] < 0.f) continue;
        if (dive) {
          if (rank[k] != 0) continue;
        } else if (strict ? !(score[k] > best) : (score[k] < best)) {
          continue;
        }
        keep_mask |= 1 << k;
        ++m;
      }
      if (dive) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (keep_mask >> k & 1) { sh->next = make_child(k); sh->has_next = 1; }
      } else if (m) {
        int slot = atomicAdd(&out.counts[sub_id], m);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          if (!(keep_mask >> k & 1)) continue;
          if (!Push3(out, sub_id, slot, make_child(k))) counters->overflow = 1;
          ++slot;
        }
      }
    }
    __syncthreads();   // scratch reused by the next node; sh->next / has_next visible
  }
}

__global__ void __launch_bounds__(256)
Expand3DKernel(Fast3DProblem P, List3 in, int strict, List3 out, List3 leaves,
               Counters3* __restrict__ counters) {
  __shared__ ExpandShared sh;
  const int max_count = ListMax3(in);
  for (int i = blockIdx.x; i < max_count * kSubLists3; i += gridDim.x) {
    const int in_sub = i & (kSubLists3 - 1), j = i / kSubLists3;
    if (j >= min(in.counts[in_sub], in.sub_capacity)) continue;   // block-uniform
    // Children go to a sub-list derived from the node's slot, not from the block: the
    // survivors of a search cluster in a few subtrees, and appending them to their
    // parent's sub-list would leave the next level with one long list that a handful
    // of blocks walk serially (measured: 0.9 us per node, chip idle).
    const int sub_id = (in_sub * 17 + j) & (kSubLists3 - 1);
    const Node3D nd = in.nodes[static_cast<size_t>(in_sub) * in.sub_capacity + j];
    const float best = __uint_as_float(
        __hip_atomic_load(&counters->best_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    if (strict ? !(nd.score > best) : (nd.score < best)) continue;
    ExpandNode3D(P, nd, best, 0, strict, out, leaves, counters, sub_id, &sh);
  }
}

// Greedy descents (always the best child) from the seeds, one block per seed, all levels in
// one launch: the verified leaf scores bound the search that follows.
__global__ void __launch_bounds__(256)
Dive3DKernel(Fast3DProblem P, List3 seeds, List3 leaves, Counters3* __restrict__ counters) {
  __shared__ ExpandShared sh;
  if (static_cast<int>(blockIdx.x) >= min(seeds.counts[0], seeds.sub_capacity)) return;
  Node3D nd = seeds.nodes[blockIdx.x];
  const int sub_id = blockIdx.x & (kSubLists3 - 1);
  while (nd.level >= 1) {
    ExpandNode3D(P, nd, 0.f, 1, 0, seeds, leaves, counters, sub_id, &sh);
    if (!sh.has_next) break;
    nd = sh.next;
    __syncthreads();   // everyone has read sh.next before the next expansion resets it
  }
}

struct Best3 {
  float score;
  int scan, ox, oy, oz;
  float low_resolution_score;
  int found, ties;
};

// Among the recorded leaves with the best score, the one the reference's
// depth-first search meets first (see fast_2d.hip SelectBestKernel).
__global__ void __launch_bounds__(1024)
SelectBest3DKernel(List3 leaves, const Counters3* __restrict__ counters, Best3* __restrict__ out) {
  __shared__ unsigned best_coarse;
  __shared__ unsigned long long best_key[2];
  __shared__ int ties;
  const int total = ListMax3(leaves) * kSubLists3;
  const unsigned best_bits = counters->best_bits;
  if (threadIdx.x == 0) {
    best_coarse = 0; ties = 0; best_key[0] = ~0ull; best_key[1] = ~0ull;
    Best3 b{};
    *out = b;
  }
  __syncthreads();
  auto leaf_at = [&](int i, Node3D* nd) {
    const int sub = i & (kSubLists3 - 1), j = i / kSubLists3;
    if (j >= min(leaves.counts[sub], leaves.sub_capacity)) return false;
    *nd = leaves.nodes[static_cast<size_t>(sub) * leaves.sub_capacity + j];
    return true;
  };
  Node3D nd;
  for (int i = threadIdx.x; i < total; i += blockDim.x)
    if (leaf_at(i, &nd) && __float_as_uint(nd.score) == best_bits) {
      atomicMax(&best_coarse, __float_as_uint(nd.coarse_score));
      atomicAdd(&ties, 1);
    }
  __syncthreads();
  // key = (coarse_index, path), minimised lexicographically in two steps
  for (int i = threadIdx.x; i < total; i += blockDim.x)
    if (leaf_at(i, &nd) && __float_as_uint(nd.score) == best_bits &&
        __float_as_uint(nd.coarse_score) == best_coarse)
      atomicMin(&best_key[0], static_cast<unsigned long long>(static_cast<unsigned>(nd.coarse_index)));
  __syncthreads();
  for (int i = threadIdx.x; i < total; i += blockDim.x)
    if (leaf_at(i, &nd) && __float_as_uint(nd.score) == best_bits &&
        __float_as_uint(nd.coarse_score) == best_coarse &&
        static_cast<unsigned long long>(static_cast<unsigned>(nd.coarse_index)) == best_key[0])
      atomicMin(&best_key[1], nd.path);
  __syncthreads();
  for (int i = threadIdx.x; i < total; i += blockDim.x)
    if (leaf_at(i, &nd) && __float_as_uint(nd.score) == best_bits &&
        __float_as_uint(nd.coarse_score) == best_coarse &&
        static_cast<unsigned long long>(static_cast<unsigned>(nd.coarse_index)) == best_key[0] &&
        nd.path == best_key[1]) {
      Best3 b;
      b.score = nd.score; b.scan = nd.scan; b.ox = nd.ox; b.oy = nd.oy; b.oz = nd.oz;
      b.low_resolution_score = nd.low_resolution_score;
      b.found = 1; b.ties = 1;
      *out = b;
    }
  __threadfence();
  __syncthreads();
  // ties = 1 + tied records that are a different leaf (dive + search duplicate the best).
  for (int i = threadIdx.x; i < total; i += blockDim.x)
    if (leaf_at(i, &nd) && __float_as_uint(nd.score) == best_bits) {
      const int scan = __hip_atomic_load(&out->scan, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int ox = __hip_atomic_load(&out->ox, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int oy = __hip_atomic_load(&out->oy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int oz = __hip_atomic_load(&out->oz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (nd.scan != scan || nd.ox != ox || nd.oy != oy || nd.oz != oz) atomicAdd(&out->ties, 1);
    }
}

// depth == 1: lowest-resolution candidates are the leaves; they are verified
// in descending score order (BranchAndBound at candidate_depth 0, :382-402).
// Rare configuration; handled by treating every candidate as a leaf group of
// one through the generic expansion of a virtual parent is not possible, so a
// dedicated wave-per-candidate pass records every passing candidate.
__global__ void __launch_bounds__(256)
VerifyCoarseLeaves3DKernel(Fast3DProblem P, List3 leaves, Counters3* __restrict__ counters) {
  __shared__ float low_prob[kLowChunk];
  const int total = P.ncx * P.ncy * P.ncz * P.num_scans;
  const int sub_id = blockIdx.x & (kSubLists3 - 1);
  for (int c = blockIdx.x; c < total; c += gridDim.x) {     // one block per candidate
    const Node3D nd = CoarseNode3D(P, c);
    if (!(nd.score > P.min_score)) continue;                // block-uniform
    const float4 q4 = P.scan_q[nd.scan];
    const float low = LowResolutionScore(
        P, Quat{q4.w, q4.x, q4.y, q4.z},
        (P.pose_tx + 0.f) + P.resolution * static_cast<float>(nd.ox),
        (P.pose_ty + 0.f) + P.resolution * static_cast<float>(nd.oy),
        (P.pose_tz + 0.f) + P.resolution * static_cast<float>(nd.oz), low_prob);
    if (static_cast<double>(low) >= P.min_low_resolution_score && threadIdx.x == 0) {
      Node3D rec = nd;
      rec.level = 0;
      rec.low_resolution_score = low;
      if (!Push3(leaves, sub_id, atomicAdd(&leaves.counts[sub_id], 1), rec)) counters->overflow = 1;
      atomicMax(&counters->best_bits, __float_as_uint(nd.score));
    }
  }
}

}  // namespace

// ---------------------------------------------------------------------------
// Matcher object
// ---------------------------------------------------------------------------
struct Fast3DMatcher {
  cmx_fast3d_options options;
  int device;
  float resolution, low_resolution;
  int width_in_voxels;
  std::vector<std::unique_ptr<DeviceBrick>> levels;
  DeviceBrick low;
  std::vector<float> histogram;
};

namespace {

// RotationalScanMatcher::RotateHistogram / MatchHistograms / Match
// (SM3/rotational_scan_matcher.cc:121-189), host side.  Reductions are
// sequential f32 (Eigen's packet reduction order is unpinned).
std::vector<float> RotateHistogram(const std::vector<float>& histogram, float angle) {
  const int size = static_cast<int>(histogram.size());
  if (size == 0) return histogram;
  const float rotate_by_buckets =
      static_cast<float>(static_cast<double>(-angle * static_cast<float>(size)) / M_PI);
  int full_buckets = static_cast<int>(std::lround(rotate_by_buckets - 0.5f));
  const float fraction = rotate_by_buckets - full_buckets;
  while (full_buckets < 0) full_buckets += size;
  std::vector<float> out(size);
  for (int i = 0; i != size; ++i) {
    const float h0 = histogram[(i + full_buckets) % size];
    const float h1 = histogram[(i + 1 + full_buckets) % size];
    out[i] = fraction * h1 + (1.f - fraction) * h0;
  }
  return out;
}
float Dot(const std::vector<float>& a, const std::vector<float>& b) {
  float s = 0.f;
  for (size_t i = 0; i != a.size(); ++i) s += a[i] * b[i];
  return s;
}
float MatchHistograms(const std::vector<float>& submap, const std::vector<float>& scan) {
  const float scan_norm = std::sqrt(Dot(scan, scan));
  const float submap_norm = std::sqrt(Dot(submap, submap));
  const float normalization = scan_norm * submap_norm;
  if (normalization < 1e-3f) return 1.f;
  return Dot(submap, scan) / normalization;
}

void Match3D(const Fast3DMatcher& m, int wxy, int wz, double angular_search_window,
             const h3::Rigid& node, const h3::Rigid& submap, const cmx_node_data3d& data,
             float min_score, int32_t* found, cmx_result3d* result, cmx_match_stats* stats) {
  CMX_REQUIRE(found && result, "null output");
  CMX_REQUIRE(data.high_resolution_point_cloud && data.num_high_resolution_points >= 1,
              "empty high-resolution point cloud");
  CMX_REQUIRE(data.low_resolution_point_cloud && data.num_low_resolution_points >= 1,
              "empty low-resolution point cloud");
  CMX_REQUIRE(data.histogram_size == static_cast<int>(m.histogram.size()),
              "histogram size %d does not match the submap's %d", data.histogram_size,
              static_cast<int>(m.histogram.size()));
  CMX_REQUIRE(data.histogram_size == 0 || data.rotational_scan_matcher_histogram != nullptr,
              "null histogram");
  CMX_REQUIRE(wxy >= 0 && wz >= 0 && wxy < (1 << 20) && wz < (1 << 20), "bad search window");
  const int n = data.num_high_resolution_points, n_low = data.num_low_resolution_points;
  const float* hi = data.high_resolution_point_cloud;
  const int depth = m.options.branch_and_bound_depth;
  *found = 0;

  // GenerateDiscreteScans (:246-295), host part.
  float max_scan_range = 3.f * m.resolution;
  for (int i = 0; i < n; ++i)
    max_scan_range = std::max(h3::Norm({hi[3 * i], hi[3 * i + 1], hi[3 * i + 2]}), max_scan_range);
  const float kSafetyMargin = 1.f - 1e-2f;
  const float step =
      kSafetyMargin * std::acos(1.f - (m.resolution * (m.resolution * 1.f)) /
                                          (2.f * (max_scan_range * (max_scan_range * 1.f))));
  const int angular_window_size = static_cast<int>(std::lround(angular_search_window / step));
  CMX_REQUIRE(angular_window_size >= 0 && angular_window_size < (1 << 20), "bad angular window");
  std::vector<float> angles;
  for (int rz = -angular_window_size; rz <= angular_window_size; ++rz) angles.push_back(rz * step);
  const h3::Rigid node_to_submap = h3::Mul(h3::InverseRigid(submap), node);
  const double* g = data.gravity_alignment;   // w, x, y, z
  const double n2 = (g[1] * g[1] + g[3] * g[3]) + (g[2] * g[2] + g[0] * g[0]);
  const h3::Q g_inv{static_cast<float>(g[0] / n2), static_cast<float>(-g[1] / n2),
                    static_cast<float>(-g[2] / n2), static_cast<float>(-g[3] / n2)};
  const float initial_angle = h3::GetYaw(h3::Mul(node_to_submap.q, g_inv));
  const std::vector<float> scan_hist(
      data.rotational_scan_matcher_histogram,
      data.rotational_scan_matcher_histogram + data.histogram_size);
  std::vector<h3::Q> pose_q;
  std::vector<float> rotational_score;
  for (size_t i = 0; i != angles.size(); ++i) {
    const float sc =
        MatchHistograms(m.histogram, RotateHistogram(scan_hist, initial_angle + angles[i]));
    if (sc < m.options.min_rotational_score) continue;
    pose_q.push_back(h3::Mul(h3::Mul(h3::Inverse(submap.q),
                                     h3::FromAngleAxisVector({0.f, 0.f, angles[i]})),
                             node.q));
    rotational_score.push_back(sc);
  }
  const int S = static_cast<int>(pose_q.size());
  cmx_match_stats st{};
  st.num_scans = S;
  if (S == 0) {
    if (stats) *stats = st;
    return;
  }
  const h3::V3 pose_t = node_to_submap.t;

  // Lowest-resolution candidates (:297-330).
  const int step_cells = 1 << (depth - 1);
  const long long ncx = (2ll * wxy + step_cells) / step_cells, ncz = (2ll * wz + step_cells) / step_cells;
  const long long per_scan = ncx * ncx * ncz;
  const long long total = per_scan * S;
  CMX_REQUIRE(total < (1ll << 30), "search too large: %lld lowest-resolution candidates", total);

  WorkspaceLease ws(m.device);
  float* d_hi = ws->dev[0].ReserveAs<float>(3 * static_cast<size_t>(n));
  float* d_low = ws->dev[1].ReserveAs<float>(3 * static_cast<size_t>(n_low));
  float4* d_pose_q = ws->dev[2].ReserveAs<float4>(2 * static_cast<size_t>(S));
  float4* d_scan_q = d_pose_q + S;
  int4* d_cells = ws->dev[3].ReserveAs<int4>(static_cast<size_t>(S) * n);
  float* d_coarse = ws->dev[4].ReserveAs<float>(total);
  // CMX_FRONTIER_CAPACITY shrinks the frontier buffers (tests only): overflow -> strict retry.
  const char* cap_env = getenv("CMX_FRONTIER_CAPACITY");
  const int cap_req = cap_env ? atoi(cap_env) : 0;
  const int kFrontierCapacity =
      cap_req >= kSubLists3 ? std::min(cap_req, 1 << 21) / kSubLists3 * kSubLists3 : 1 << 21;
  const int kLeafCapacity = 1 << 18;
  Node3D* d_front[2] = {ws->dev[5].ReserveAs<Node3D>(kFrontierCapacity),
                        ws->dev[6].ReserveAs<Node3D>(kFrontierCapacity)};
  Node3D* d_leaves = ws->dev[7].ReserveAs<Node3D>(kLeafCapacity);
  const int kDiveSub = 256;   // a dive list never holds more than kSeeds3 nodes per sub-list
  Node3D* d_seeds = ws->dev[8].ReserveAs<Node3D>(2 * static_cast<size_t>(kDiveSub) * kSubLists3);
  char* d_misc = static_cast<char*>(ws->dev[9].Reserve(sizeof(Counters3) + sizeof(Best3)));
  Counters3* d_counters = reinterpret_cast<Counters3*>(d_misc);
  Best3* d_best = reinterpret_cast<Best3*>(d_misc + sizeof(Counters3));

  float4* h_q = ws->pinned[0].ReserveAs<float4>(2 * static_cast<size_t>(S));
  std::vector<h3::Q> scan_q(S);
  for (int s = 0; s < S; ++s) {
    h_q[s] = make_float4(pose_q[s].x, pose_q[s].y, pose_q[s].z, pose_q[s].w);
    // GetPoseFromCandidate (:369-375): Translation(res * offset) * pose renormalises
    // the rotation; Identity * q is exact, the normalisation is not.
    const h3::Q iq = h3::Normalized(h3::Mul(h3::Q{1.f, 0.f, 0.f, 0.f}, pose_q[s]));
    scan_q[s] = iq;
    h_q[S + s] = make_float4(iq.x, iq.y, iq.z, iq.w);
  }
  Counters3* h_counters = static_cast<Counters3*>(
      ws->pinned[1].Reserve(sizeof(Counters3) + sizeof(Best3)));
  std::memset(h_counters, 0, sizeof(Counters3));
  {
    const float floor_score = std::max(min_score, 0.f);
    std::memcpy(&h_counters->best_bits, &floor_score, sizeof(float));
  }
  // The high-resolution cloud only ever feeds integer sums (ScoreCandidates), which do
  // not depend on the order of the points.  Upload it sorted along a Morton curve: the
  // 64 points a wavefront gathers together then fall into neighbouring voxels, i.e. into
  // a handful of cache lines instead of 64 (the search is bound by that line traffic).
  float* h_hi = ws->pinned[2].ReserveAs<float>(3 * static_cast<size_t>(n));
  {
    float lo3[3] = {hi[0], hi[1], hi[2]};
    for (int i = 1; i < n; ++i)
      for (int k = 0; k < 3; ++k) lo3[k] = std::min(lo3[k], hi[3 * i + k]);
    const float inv_cell = 1.f / (2.f * m.resolution);
    auto spread = [](uint32_t v) {   // 10 bits -> every third bit
      v &= 0x3ffu;
      v = (v | (v << 16)) & 0x030000ffu;
      v = (v | (v << 8)) & 0x0300f00fu;
      v = (v | (v << 4)) & 0x030c30c3u;
      v = (v | (v << 2)) & 0x09249249u;
      return v;
    };
    std::vector<uint64_t> order(n);
    for (int i = 0; i < n; ++i) {
      uint32_t key = 0;
      for (int k = 0; k < 3; ++k) {
        const float cell = (hi[3 * i + k] - lo3[k]) * inv_cell;
        const uint32_t c = cell >= 1023.f ? 1023u : (cell > 0.f ? static_cast<uint32_t>(cell) : 0u);
        key |= spread(c) << k;
      }
      order[i] = (static_cast<uint64_t>(key) << 32) | static_cast<uint32_t>(i);
    }
    std::sort(order.begin(), order.end());
    for (int i = 0; i < n; ++i) {
      const uint32_t src = static_cast<uint32_t>(order[i]);
      h_hi[3 * i] = hi[3 * src]; h_hi[3 * i + 1] = hi[3 * src + 1]; h_hi[3 * i + 2] = hi[3 * src + 2];
    }
  }
  CMX_HIP(hipMemcpyAsync(d_hi, h_hi, 3 * sizeof(float) * n, hipMemcpyHostToDevice, ws->stream));
  CMX_HIP(hipMemcpyAsync(d_low, data.low_resolution_point_cloud, 3 * sizeof(float) * n_low,
                         hipMemcpyHostToDevice, ws->stream));
  CMX_HIP(hipMemcpyAsync(d_pose_q, h_q, 2 * sizeof(float4) * S, hipMemcpyHostToDevice,
                         ws->stream));
  CMX_HIP(hipMemcpyAsync(d_counters, h_counters, sizeof(Counters3), hipMemcpyHostToDevice,
                         ws->stream));

  Fast3DProblem P{};
  for (int d = 0; d < depth; ++d) P.level[d] = m.levels[d]->desc;
  P.depth = depth;
  P.full_resolution_depth = m.options.full_resolution_depth;
  P.low = m.low.desc;
  P.low_resolution = m.low_resolution;
  P.resolution = m.resolution;
  P.wxy = wxy; P.wz = wz;
  P.num_scans = S; P.n = n; P.n_low = n_low;
  P.cells = d_cells;
  P.low_xyz = d_low;
  P.scan_q = d_scan_q;
  P.pose_tx = pose_t.x; P.pose_ty = pose_t.y; P.pose_tz = pose_t.z;
  P.min_score = min_score;
  P.min_low_resolution_score = m.options.min_low_resolution_score;
  P.ncx = static_cast<int>(ncx); P.ncy = static_cast<int>(ncx); P.ncz = static_cast<int>(ncz);
  P.coarse_score = d_coarse;

  auto front = [&](int stage) {
    return List3{d_front[stage & 1], d_counters->frontier[stage], kFrontierCapacity / kSubLists3};
  };
  const List3 leaf_list{d_leaves, d_counters->leaves, kLeafCapacity / kSubLists3};

  const char* dbg_env = getenv("CMX_SYNC");
  const bool dbg_sync = dbg_env && dbg_env[0] == '1';
  auto dbg = [&](const char* name) {
    if (!dbg_sync) return;
    fprintf(stderr, "[cmx sync] %s ...\n", name);
    CMX_HIP(hipStreamSynchronize(ws->stream));
  };
  dbg("uploads");
  StageTrace trace(ws->stream);
  auto mark = [&](const char* name) { trace.Mark(name); };
  mark("begin");
  CMX_HIP(hipEventRecord(ws->ev_begin, ws->stream));
  Discretize3DKernel<<<dim3(DivUp(n, 256), S), 256, 0, ws->stream>>>(
      d_hi, n, d_pose_q, pose_t.x, pose_t.y, pose_t.z, m.resolution, d_cells);
  dbg("discretize");
  mark("discretize");
  CMX_HIP(hipEventRecord(ws->ev_k0, ws->stream));
  if (total <= 4096)
    ScoreCoarse3DBlockKernel<<<static_cast<unsigned>(total), 256, 0, ws->stream>>>(P);
  else
    ScoreCoarse3DKernel<<<std::min<long long>(8192, DivUp(total, 4)), 256, 0, ws->stream>>>(P);
  CMX_HIP(hipEventRecord(ws->ev_k1, ws->stream));
  dbg("coarse");
  mark("coarse");

  const int blocks = 2048;
  int strict = 0, num_chunks = 1;
  for (;;) {
    if (depth == 1) {
      VerifyCoarseLeaves3DKernel<<<blocks, 256, 0, ws->stream>>>(P, leaf_list, d_counters);
    } else {
      if (!strict) {
        // dive: greedy descents from the best lowest-resolution candidates give
        // a verified leaf score to bound the search with.
        List3 dive[2] = {{d_seeds, d_counters->dive[0], kDiveSub},
                         {d_seeds + kDiveSub * kSubLists3, d_counters->dive[1], kDiveSub}};
        SeedSelect3DKernel<<<1, 1024, 0, ws->stream>>>(P, dive[0], d_counters);
        dbg("seed");
        mark("seed");
        Dive3DKernel<<<kSeeds3, 256, 0, ws->stream>>>(P, dive[0], leaf_list, d_counters);
        dbg("dive");
        mark("dive");
      }
      // The lowest-resolution candidates are searched in `num_chunks` interleaved subsets
      // (1 unless an earlier pass overflowed); later chunks profit from the bound the
      // earlier ones raised.
      for (int chunk = 0; chunk < num_chunks; ++chunk) {
        CMX_HIP(hipMemsetAsync(d_counters->frontier, 0, sizeof(d_counters->frontier),
                               ws->stream));
        Filter3DKernel<<<256, 256, 0, ws->stream>>>(P, strict, chunk, num_chunks, front(0),
                                                    d_counters);
        dbg("filter");
        mark("filter");
        int stage = 0;
        for (int child = depth - 2; child >= 0; --child, ++stage) {
          Expand3DKernel<<<blocks, 256, 0, ws->stream>>>(P, front(stage), strict,
                                                         front(stage + 1), leaf_list, d_counters);
          dbg("expand level");
          mark("expand");
        }
      }
    }
    SelectBest3DKernel<<<1, 1024, 0, ws->stream>>>(leaf_list, d_counters, d_best);
    mark("select");
    CMX_HIP(hipGetLastError());
    CMX_HIP(hipEventRecord(ws->ev_end, ws->stream));
    CMX_HIP(hipMemcpyAsync(h_counters, d_counters, sizeof(Counters3) + sizeof(Best3),
                           hipMemcpyDeviceToHost, ws->stream));
    CMX_HIP(hipStreamSynchronize(ws->stream));
    trace.Report();
    if (!h_counters->overflow) break;
    // Something was dropped.  Retry pruning ties (strict) with the bound lowered by one
    // ulp so the best leaf is found again, over four times as many, smaller chunks.
    CMX_REQUIRE(num_chunks < (1 << 12), "branch-and-bound frontier overflow (search too wide)");
    if (strict) num_chunks *= 4;
    strict = 1;
    const float floor_score = std::max(min_score, 0.f);
    unsigned floor_bits;
    std::memcpy(&floor_bits, &floor_score, sizeof(float));
    Counters3 reset{};
    reset.best_bits = h_counters->best_bits > floor_bits ? h_counters->best_bits - 1 : floor_bits;
    std::memcpy(reset.scored, h_counters->scored, sizeof(reset.scored));
    std::memcpy(reset.expanded, h_counters->expanded, sizeof(reset.expanded));
    *h_counters = reset;
    CMX_HIP(hipMemcpyAsync(d_counters, h_counters, sizeof(Counters3), hipMemcpyHostToDevice,
                           ws->stream));
  }
  const Best3* h_best = reinterpret_cast<const Best3*>(reinterpret_cast<char*>(h_counters) +
                                                       sizeof(Counters3));
  st.coarse_candidates = total;
  st.candidates_scored = total;
  for (int k = 0; k < 16; ++k) {
    st.candidates_scored += h_counters->scored[k];
    st.nodes_expanded += h_counters->expanded[k];
  }
  float ms = 0.f;
  CMX_HIP(hipEventElapsedTime(&ms, ws->ev_begin, ws->ev_end));
  st.device_ms = ms;
  CMX_HIP(hipEventElapsedTime(&ms, ws->ev_k0, ws->ev_k1));
  st.dominant_kernel_ms = ms;
  if (stats) *stats = st;
  Best3 best = *h_best;
  if (best.found && best.ties > 1) {
    // Exact tie resolution (see fast_2d.hip ResolveTies): repeat the reference's
    // std::sort of the lowest-resolution candidates (:352-353) and take the
    // tied leaf its depth-first search meets first.  The dive and the search
    // record the same leaf twice, so first check that distinct leaves tie.
    unsigned best_bits;
    std::memcpy(&best_bits, &best.score, sizeof(float));
    std::vector<Node3D> tied;
    std::vector<Node3D> sub_nodes;
    for (int sub = 0; sub < kSubLists3; ++sub) {
      const int count = std::min(h_counters->leaves[sub], leaf_list.sub_capacity);
      if (count <= 0) continue;
      sub_nodes.resize(count);
      CMX_HIP(hipMemcpy(sub_nodes.data(),
                        leaf_list.nodes + static_cast<size_t>(sub) * leaf_list.sub_capacity,
                        count * sizeof(Node3D), hipMemcpyDeviceToHost));
      for (const Node3D& nd : sub_nodes) {
        unsigned bits;
        std::memcpy(&bits, &nd.score, sizeof(float));
        if (bits == best_bits) tied.push_back(nd);
      }
    }
    bool distinct = false;
    for (const Node3D& nd : tied)
      distinct |= !(nd.scan == tied[0].scan && nd.ox == tied[0].ox && nd.oy == tied[0].oy &&
                    nd.oz == tied[0].oz);
    if (distinct) {
      std::vector<float> scores(total);
      CMX_HIP(hipMemcpy(scores.data(), d_coarse, total * sizeof(float), hipMemcpyDeviceToHost));
      struct ScoreIndex {
        float score; int index;
        bool operator>(const ScoreIndex& o) const { return score > o.score; }
      };
      std::vector<ScoreIndex> sorted(total);
      for (long long c = 0; c < total; ++c) sorted[c] = {scores[c], static_cast<int>(c)};
      std::sort(sorted.begin(), sorted.end(), std::greater<ScoreIndex>());
      std::vector<int> position(total);
      for (long long i = 0; i < total; ++i) position[sorted[i].index] = static_cast<int>(i);
      bool have = false;
      int best_pos = 0;
      unsigned long long best_path = 0;
      for (const Node3D& nd : tied) {
        const int pos = position[nd.coarse_index];
        if (!have || pos < best_pos || (pos == best_pos && nd.path < best_path)) {
          have = true;
          best_pos = pos;
          best_path = nd.path;
          best.scan = nd.scan; best.ox = nd.ox; best.oy = nd.oy; best.oz = nd.oz;
          best.low_resolution_score = nd.low_resolution_score;
        }
      }
    }
  }
  if (best.found && best.score > min_score) {
    *found = 1;
    result->score = best.score;
    h3::Rigid pose;
    // Translation(res * offset) * scan.pose
    pose.t = {(pose_t.x + 0.f) + m.resolution * static_cast<float>(best.ox),
              (pose_t.y + 0.f) + m.resolution * static_cast<float>(best.oy),
              (pose_t.z + 0.f) + m.resolution * static_cast<float>(best.oz)};
    pose.q = scan_q[best.scan];
    result->pose_estimate = h3::ToPose(pose);
    result->rotational_score = rotational_score[best.scan];
    result->low_resolution_score = best.low_resolution_score;
  }
}

}  // namespace
}  // namespace cmx

struct cmx_fast3d {
  cmx::Fast3DMatcher impl;
};

namespace cmx {
// For sharded.hip: the device a 3D matcher's grids live on.
int Fast3DDevice(const cmx_fast3d* matcher) { return matcher->impl.device; }
}  // namespace cmx

extern "C" {

cmx_status cmx_fast3d_create(const cmx_fast3d_options* options, float resolution,
                             int32_t grid_size, const cmx_voxel* voxels, int64_t num_voxels,
                             float low_resolution, const cmx_voxel* low_resolution_voxels,
                             int64_t num_low_resolution_voxels,
                             const float* rotational_scan_matcher_histogram,
                             int32_t histogram_size, int32_t device, cmx_fast3d** out) {
  using namespace cmx;
  return Guard([&] {
    CMX_REQUIRE(options && out, "null argument");
    *out = nullptr;
    // CHECKs of PrecomputationGridStack3D (:60-61).
    CMX_REQUIRE(options->branch_and_bound_depth >= 1 && options->branch_and_bound_depth <= kMaxDepth,
                "branch_and_bound_depth %d outside [1,%d]", options->branch_and_bound_depth,
                kMaxDepth);
    CMX_REQUIRE(options->full_resolution_depth >= 1, "full_resolution_depth must be >= 1");
    CMX_REQUIRE(resolution > 0.f && low_resolution > 0.f, "resolutions must be > 0");
    CMX_REQUIRE(num_voxels == 0 || voxels, "voxels is null");
    CMX_REQUIRE(num_low_resolution_voxels == 0 || low_resolution_voxels, "low voxels null");
    CMX_REQUIRE(histogram_size >= 0 && (histogram_size == 0 || rotational_scan_matcher_histogram),
                "bad histogram");
    CMX_REQUIRE(grid_size >= GridSizeOf(voxels, num_voxels),
                "grid_size %d is smaller than the voxels' extent", grid_size);
    std::unique_ptr<cmx_fast3d> h(new cmx_fast3d);
    Fast3DMatcher& m = h->impl;
    m.options = *options;
    m.device = device;
    m.resolution = resolution;
    m.low_resolution = low_resolution;
    m.width_in_voxels = grid_size;
    m.histogram.assign(rotational_scan_matcher_histogram,
                       rotational_scan_matcher_histogram + histogram_size);
    WorkspaceLease ws(device);
    m.levels.emplace_back(new DeviceBrick);
    BuildBrickFromVoxels(*ws, voxels, num_voxels, 1, m.levels[0].get());
    CMX_REQUIRE(m.levels[0]->bytes < (size_t(1) << 31), "grid too large");   // 32-bit cell offsets
    BuildBrickFromVoxels(*ws, low_resolution_voxels, num_low_resolution_voxels, 2, &m.low);
    // PrecomputationGridStack3D (:57-77).
    int last_width = 1;
    for (int depth = 1; depth != options->branch_and_bound_depth; ++depth) {
      const bool half = depth >= options->full_resolution_depth;
      const int next_width = 1 << depth;
      const int per_voxel = 1 << std::max(0, depth - options->full_resolution_depth);
      const int shift = (next_width - last_width + (per_voxel - 1)) / per_voxel;
      const Brick prev = m.levels.back()->desc;
      Brick b{};
      int lo[3] = {prev.lo_x - shift, prev.lo_y - shift, prev.lo_z - shift};
      int hi[3] = {prev.lo_x + prev.nx - 1, prev.lo_y + prev.ny - 1, prev.lo_z + prev.nz - 1};
      if (half) {
        for (int k = 0; k < 3; ++k) { lo[k] >>= 1; hi[k] >>= 1; }
      }
      b.lo_x = lo[0]; b.lo_y = lo[1]; b.lo_z = lo[2];
      b.nx = hi[0] - lo[0] + 1; b.ny = hi[1] - lo[1] + 1; b.nz = hi[2] - lo[2] + 1;
      std::unique_ptr<DeviceBrick> level(new DeviceBrick);
      level->bytes = static_cast<size_t>(b.nx) * b.ny * b.nz;
      CMX_REQUIRE(level->bytes < (size_t(1) << 31), "precomputation level too large");
      CMX_HIP(hipMalloc(&level->mem, level->bytes));
      b.cells = level->mem;
      level->desc = b;
      PrecomputeLevel3DKernel<<<DivUp(level->bytes, 256), 256, 0, ws->stream>>>(prev, b, shift,
                                                                               half ? 1 : 0);
      CMX_HIP(hipGetLastError());
      m.levels.push_back(std::move(level));
      last_width = next_width;
    }
    CMX_HIP(hipStreamSynchronize(ws->stream));
    *out = h.release();
  });
}

void cmx_fast3d_destroy(cmx_fast3d* matcher) {
  if (!matcher) return;
  (void)hipSetDevice(matcher->impl.device);
  delete matcher;
}

cmx_status cmx_fast3d_match(const cmx_fast3d* matcher, const cmx_pose3d* global_node_pose,
                            const cmx_pose3d* global_submap_pose, const cmx_node_data3d* data,
                            float min_score, int32_t* found, cmx_result3d* result,
                            cmx_match_stats* stats) {
  using namespace cmx;
  return Guard([&] {
    CMX_REQUIRE(matcher && global_node_pose && global_submap_pose && data, "null argument");
    const Fast3DMatcher& m = matcher->impl;
    // Match (:127-146).
    const int wxy = static_cast<int>(std::lround(m.options.linear_xy_search_window / m.resolution));
    const int wz = static_cast<int>(std::lround(m.options.linear_z_search_window / m.resolution));
    Match3D(m, wxy, wz, m.options.angular_search_window, h3::FromPose(*global_node_pose),
            h3::FromPose(*global_submap_pose), *data, min_score, found, result, stats);
  });
}

cmx_status cmx_fast3d_match_full_submap(const cmx_fast3d* matcher,
                                        const double* global_node_rotation_wxyz,
                                        const double* global_submap_rotation_wxyz,
                                        const cmx_node_data3d* data, float min_score,
                                        int32_t* found, cmx_result3d* result,
                                        cmx_match_stats* stats) {
  using namespace cmx;
  return Guard([&] {
    CMX_REQUIRE(matcher && global_node_rotation_wxyz && global_submap_rotation_wxyz && data,
                "null argument");
    CMX_REQUIRE(data->high_resolution_point_cloud && data->num_high_resolution_points >= 1,
                "empty high-resolution point cloud");
    const Fast3DMatcher& m = matcher->impl;
    // MatchFullSubmap (:148-170).
    float max_point_distance = 0.f;
    for (int i = 0; i < data->num_high_resolution_points; ++i) {
      const float* p = data->high_resolution_point_cloud + 3 * i;
      max_point_distance = std::max(max_point_distance, h3::Norm({p[0], p[1], p[2]}));
    }
    const int window = (m.width_in_voxels + 1) / 2 +
                       static_cast<int>(std::lround(max_point_distance / m.resolution + 0.5f));
    h3::Rigid node, submap;
    node.q = {static_cast<float>(global_node_rotation_wxyz[0]),
              static_cast<float>(global_node_rotation_wxyz[1]),
              static_cast<float>(global_node_rotation_wxyz[2]),
              static_cast<float>(global_node_rotation_wxyz[3])};
    submap.q = {static_cast<float>(global_submap_rotation_wxyz[0]),
                static_cast<float>(global_submap_rotation_wxyz[1]),
                static_cast<float>(global_submap_rotation_wxyz[2]),
                static_cast<float>(global_submap_rotation_wxyz[3])};
    Match3D(m, window, window, M_PI, node, submap, *data, min_score, found, result, stats);
  });
}

// The ConstraintBuilder3D fan-out (constraints/constraint_builder_3d.cc:79-147): one node's
// constant data against many submaps' matchers, windowed and full-submap pairs mixed.  The
// reference runs one thread-pool task per pair; here the pairs run concurrently from a few host
// threads, each on its own leased stream and scratch (the matchers are immutable, Match3D is
// re-entrant), so the short dependent kernel chains of independent searches overlap on the
// device.  `node_poses[p]` / `submap_poses[p]`: the global poses of pair p (only their rotations
// are read where match_full_submap[p] != 0).
cmx_status cmx_fast3d_match_batch(const cmx_fast3d* const* matchers, int32_t num_pairs,
                                  const cmx_pose3d* node_poses, const cmx_pose3d* submap_poses,
                                  const int32_t* match_full_submap, const float* min_scores,
                                  const cmx_node_data3d* data, int32_t* found,
                                  cmx_result3d* results, cmx_match_stats* stats) {
  using namespace cmx;
  return Guard([&] {
    CMX_REQUIRE(matchers && node_poses && submap_poses && match_full_submap && min_scores &&
                    data && found && results && num_pairs >= 1,
                "null argument");
    for (int p = 0; p < num_pairs; ++p) CMX_REQUIRE(matchers[p] != nullptr, "null matcher handle");
    std::vector<cmx_match_stats> pair_stats(num_pairs);
    std::vector<cmx_status> status(num_pairs, CMX_OK);
    std::vector<std::string> errors(num_pairs);
    std::atomic<int> next{0};
    const auto worker = [&] {
      for (int p = next.fetch_add(1); p < num_pairs; p = next.fetch_add(1)) {
        found[p] = 0;
        if (match_full_submap[p]) {
          status[p] = cmx_fast3d_match_full_submap(matchers[p], node_poses[p].q,
                                                   submap_poses[p].q, data, min_scores[p],
                                                   &found[p], &results[p], &pair_stats[p]);
        } else {
          status[p] = cmx_fast3d_match(matchers[p], &node_poses[p], &submap_poses[p], data,
                                       min_scores[p], &found[p], &results[p], &pair_stats[p]);
        }
        if (status[p] != CMX_OK) errors[p] = LastError();   // thread-local text of this worker
      }
    };
    const int num_threads = std::min(num_pairs, 8);
    std::vector<std::thread> threads;
    for (int t = 1; t < num_threads; ++t) threads.emplace_back(worker);
    worker();                                              // the calling thread takes its share
    for (std::thread& t : threads) t.join();
    cmx_match_stats total{};
    for (int p = 0; p < num_pairs; ++p) {
      if (status[p] != CMX_OK) {
        SetLastError("pair %d: %s", p, errors[p].c_str());
        throw HipError{status[p]};
      }
      total.candidates_scored += pair_stats[p].candidates_scored;
      total.coarse_candidates += pair_stats[p].coarse_candidates;
      total.nodes_expanded += pair_stats[p].nodes_expanded;
      total.num_scans += pair_stats[p].num_scans;
      total.device_ms += pair_stats[p].device_ms;          // summed: the searches overlap
      total.dominant_kernel_ms += pair_stats[p].dominant_kernel_ms;
    }
    if (stats) *stats = total;
  });
}

// Introspection for the parity tests: dimensions / contents of one
// precomputation level (dense brick, x fastest).
cmx_status cmx_fast3d_level_info(const cmx_fast3d* matcher, int32_t depth, int32_t* lo_xyz,
                                 int32_t* dims_xyz) {
  using namespace cmx;
  return Guard([&] {
    CMX_REQUIRE(matcher && lo_xyz && dims_xyz, "null argument");
    CMX_REQUIRE(depth >= 0 && depth < static_cast<int>(matcher->impl.levels.size()), "bad depth");
    const Brick& b = matcher->impl.levels[depth]->desc;
    lo_xyz[0] = b.lo_x; lo_xyz[1] = b.lo_y; lo_xyz[2] = b.lo_z;
    dims_xyz[0] = b.nx; dims_xyz[1] = b.ny; dims_xyz[2] = b.nz;
  });
}

cmx_status cmx_fast3d_level_cells(const cmx_fast3d* matcher, int32_t depth, uint8_t* out) {
  using namespace cmx;
  return Guard([&] {
    CMX_REQUIRE(matcher && out, "null argument");
    CMX_REQUIRE(depth >= 0 && depth < static_cast<int>(matcher->impl.levels.size()), "bad depth");
    UseDevice(matcher->impl.device);
    const DeviceBrick& b = *matcher->impl.levels[depth];
    CMX_HIP(hipMemcpy(out, b.mem, b.bytes, hipMemcpyDeviceToHost));
  });
}

}  // extern "C"
