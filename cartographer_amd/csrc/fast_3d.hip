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
#include <chrono>
#include <thread>
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "scan_matching_3d.h"

namespace cmx {
namespace {

constexpr int kSubLists3 = 64;
// (Padding the sub-list counters to a cache line each, which took the 2D coarse filter from 323
// to 30 us, measured nothing here -- a node's list reservation is once per block -- and cost
// 40 us per single search in the larger counter copies.)
constexpr int kCountStride3 = 1;
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
  int problem;             // index into the batch's Fast3DProblem array
  int family;              // > 0: this node and the next family - 1 slots of its sub-list are the
                           // children one parent kept (same problem, scan and level, offsets
                           // half a parent step apart); 0: a later member of such a run
  int pad;
};

struct Counters3 {
  int frontier[kMaxDepth + 2][kSubLists3 * kCountStride3];
  int dive[2][kSubLists3];
  int leaves[kSubLists3 * kCountStride3];
  int overflow;
  int pad0;
  int pad1;
  int pad;
  unsigned long long scored[16];
  unsigned long long expanded[16];
};

struct List3 {
  Node3D* nodes;
  int* counts;
  int sub_capacity;
};

// The eight cells the children of a node read for one point, in ONE 8-byte word (what quads
// are to the 2D search): oct(X, Y, Z) byte k = level(x + (k & 1) s, y + (k >> 1 & 1) s,
// z + (k >> 2) s) with (x, y, z) = (X, Y, Z) - s relative to the level's brick, cells outside
// the brick 0; s = the level's child stride 2^min(level, full_resolution_depth - 1).  One
// gather per point and node instead of four (eight in round 1); 8x the bytes of the level
// itself, i.e. ~90 MB per 150^3 submap instead of 12 -- HBM is not the scarce resource.
typedef int I4 __attribute__((ext_vector_type(4)));   // (an int4 the compiler can load from address space 1)

struct OctDesc {
  const uint2* cells;      // [(nz + s)][(ny + s)][(nx + s)]; null: not built
  int qx, qy, qz, s;
};

// Bytes of a level's oct array (the buffer resource's range); 0 = not addressable by one
// (>= 2 GB: a level of more than ~640^3 cells keeps plain loads).
__device__ __forceinline__ unsigned long long OctBytes(const OctDesc& O) {
  const unsigned long long bytes =
      static_cast<unsigned long long>(O.qx) * O.qy * static_cast<unsigned long long>(O.qz) * 8ull;
  return bytes < kMaxBufferBytes ? bytes : 0ull;
}

struct Fast3DProblem {
  Brick level[kMaxDepth];
  OctDesc oct[kMaxDepth];
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
  // Per-problem search state (a batch of searches shares the frontier and leaf lists; nodes
  // carry their problem's index).
  unsigned* best_bits;      // float bits of the best verified leaf (>= min_score floor)
  Node3D* seeds;            // [kSeeds3] dive seeds
  int* seed_count;
  int index;                // this problem's index in the batch
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

__global__ void BuildOct3DKernel(Brick L, int s, uint2* __restrict__ out, int qx, int qy, int qz) {
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= static_cast<size_t>(qx) * qy * qz) return;
  const int X = static_cast<int>(i % qx), Y = static_cast<int>((i / qx) % qy),
            Z = static_cast<int>(i / (static_cast<size_t>(qx) * qy));
  const uint8_t* __restrict__ cells = static_cast<const uint8_t*>(L.cells);
  unsigned lo = 0, hi = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int x = X - s + ((k & 1) ? s : 0), y = Y - s + ((k & 2) ? s : 0),
              z = Z - s + ((k & 4) ? s : 0);
    unsigned v = 0;
    if (static_cast<unsigned>(x) < static_cast<unsigned>(L.nx) &&
        static_cast<unsigned>(y) < static_cast<unsigned>(L.ny) &&
        static_cast<unsigned>(z) < static_cast<unsigned>(L.nz))
      v = cells[(static_cast<size_t>(z) * L.ny + y) * L.nx + x];
    if (k < 4) lo |= v << (8 * k); else hi |= v << (8 * (k - 4));
  }
  out[i] = make_uint2(lo, hi);
}

// ---------------------------------------------------------------------------
// Scan discretisation (DiscretizeScan, :200-244: transform + GetCellIndex)
// ---------------------------------------------------------------------------
// grid (ceil(n / 256), scans of the whole batch): `pose_t[s]` = the translation of the scan's
// problem, .w its grid resolution.
__global__ void __launch_bounds__(256)
Discretize3DKernel(const float* __restrict__ xyz, int n, const float4* __restrict__ pose_q,
                   const float4* __restrict__ pose_t, int4* __restrict__ cells) {
  const int s = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 q4 = pose_q[s];
  const float4 t4 = pose_t[s];
  const Quat q{q4.w, q4.x, q4.y, q4.z};
  const F3 p{xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
  const F3 r = Rotate(q, p);
  const F3 t{r.x + t4.x, r.y + t4.y, r.z + t4.z};
  const int3 c = CellIndex3(t, t4.w);
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

// The same without the branch on e (for e == 0 the shifts are no-ops and the expression is c):
// a branch inside an unrolled gather loop is a basic-block boundary the loads cannot cross.
__device__ __forceinline__ int3 DepthIndexAny(int cx, int cy, int cz, int e, int sx, int sy,
                                              int sz) {
  return make_int3(((cx + sx) >> e) - (sx >> e), ((cy + sy) >> e) - (sy >> e),
                   ((cz + sz) >> e) - (sz >> e));
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

// grid (blocks, problems)
__global__ void __launch_bounds__(256)
ScoreCoarse3DKernel(const Fast3DProblem* __restrict__ problems) {
  const Fast3DProblem& P = problems[blockIdx.y];
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
ScoreCoarse3DBlockKernel(const Fast3DProblem* __restrict__ problems) {
  const Fast3DProblem& P = problems[blockIdx.y];
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
  nd.problem = P.index;
  nd.family = 1;
  nd.pad = 0;
  return nd;
}

__device__ __forceinline__ bool Push3(const List3& list, int sub, int slot, const Node3D& nd) {
  if (slot >= list.sub_capacity) return false;
  list.nodes[static_cast<size_t>(sub) * list.sub_capacity + slot] = nd;
  return true;
}
__device__ __forceinline__ int ListMax3(const List3& list) {
  return WaveMax(min(list.counts[(threadIdx.x & 63) * kCountStride3], list.sub_capacity));
}

// Seeds of the dive: the ~64 best lowest-resolution candidates (histogram
// threshold on the scores).
__global__ void __launch_bounds__(1024)
SeedSelect3DKernel(const Fast3DProblem* __restrict__ problems) {
  const Fast3DProblem& P = problems[blockIdx.x];
  const List3 seeds{P.seeds, P.seed_count, kSeeds3};
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
// grid (blocks, problems)
__global__ void __launch_bounds__(256)
Filter3DKernel(const Fast3DProblem* __restrict__ problems, int strict, int chunk, int num_chunks,
               int affinity, List3 out, Counters3* __restrict__ counters) {
  const Fast3DProblem& P = problems[blockIdx.y];
  const int total = P.ncx * P.ncy * P.ncz * P.num_scans;
  const float best = __uint_as_float(*P.best_bits);
  // With `affinity` the nodes of a problem stay in the sub-lists the workgroups of ONE XCD
  // read (sub % 8 == problem % 8, see Expand3DKernel): lines of its grids that neighbouring
  // nodes share are then found in that XCD's L2 instead of being fetched by eight.
  const int sub = affinity ? (blockIdx.y & 7) | ((blockIdx.x & 7) << 3)
                           : blockIdx.x & (kSubLists3 - 1);
  // One reservation per wavefront: the survivors among 64 consecutive candidates (neighbours
  // in x, then y, z) take consecutive slots, so that nodes which read the same cache lines sit
  // next to each other in the list and are expanded at about the same time.
  const int lane = threadIdx.x & 63;
  for (int c0 = blockIdx.x * blockDim.x + (threadIdx.x & ~63); c0 < total;
       c0 += gridDim.x * blockDim.x) {
    const int c = c0 + lane;
    bool keep = false;
    if (c < total && c % num_chunks == chunk) {
      const float sc = P.coarse_score[c];
      keep = strict ? (sc > best) : (sc >= best);
    }
    const unsigned long long mask = __ballot(keep);
    if (mask == 0) continue;
    int first = 0;
    if (lane == 0) first = atomicAdd(&out.counts[sub * kCountStride3], __popcll(mask));
    first = __builtin_amdgcn_readfirstlane(first);
    if (keep && !Push3(out, sub, first + __popcll(mask & ((1ull << lane) - 1)), CoarseNode3D(P, c)))
      counters->overflow = 1;
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

// Debug switch fast3d_byte_loads (tools / tests): every child cell with its own byte load.
__device__ int g_fast3d_byte_loads = 0;

// Integer sums of the <= 8 children of `nd` over the points first, first + stride, ...
// (ScoreCandidates at the child level, :332-355): the eight child cells of a point are addressed
// from per-axis clamped offsets (two positions per axis), loads of several points in flight.
__device__ __forceinline__ void ChildSums3D(const Fast3DProblem& P, const Node3D& nd, int first,
                                            int stride, int sum[8]) {
  {
    const int child_depth = nd.level - 1;
    const int half = 1 << child_depth;
    const int e = max(0, child_depth - P.full_resolution_depth + 1);
    const Brick L = P.level[child_depth];
    const uint8_t* __restrict__ cells8 = static_cast<const uint8_t*>(L.cells);
    const int4* __restrict__ cells = P.cells + static_cast<size_t>(nd.scan) * P.n;
    // Shifted offsets of the 2 positions per axis, relative to the brick origin.
    const int fx[2] = {(nd.ox >> e) - L.lo_x, ((nd.ox + half) >> e) - L.lo_x};
    const int fy[2] = {(nd.oy >> e) - L.lo_y, ((nd.oy + half) >> e) - L.lo_y};
    const int fz[2] = {(nd.oz >> e) - L.lo_z, ((nd.oz + half) >> e) - L.lo_z};
    const int row = L.nx, slab = L.nx * L.ny;
    // The two x positions of a point lie dx = fx[1] - fx[0] cells apart in one row (1, 2 or 4
    // for the usual full_resolution_depth <= 3).  Up to dx == 4 ONE aligned 8-byte load serves
    // both: half the gather instructions of the search, which are what bounds it.
    const int dx = fx[1] - fx[0];
    const OctDesc O = P.oct[child_depth];
    if (O.cells != nullptr && O.s == dx && !g_fast3d_byte_loads) {
      // One 8-byte gather per point: the eight child cells (see OctDesc).  Bytes are summed
      // as packed 16-bit pairs, widened every 256 points.
      const int ox = fx[0] + O.s, oy = fy[0] + O.s, oz = fz[0] + O.s;
      // (round 4: through a buffer resource when the array is addressable by one -- the plain
      // `O.cells[inside ? index : 0]` was a flat load under an exec mask with a wait of its
      // own, so the four gathers of the unrolled loop went out one after the other)
      const unsigned long long oct_bytes = OctBytes(O);
      const __amdgpu_buffer_rsrc_t oct = UniformBuffer(O.cells, oct_bytes);
      const auto* gcells = AsGlobal(reinterpret_cast<const I4*>(cells));
      const int n = P.n, wxy = P.wxy, wz = P.wz;
      typedef unsigned U2 __attribute__((ext_vector_type(2)));
      for (int q0 = first; q0 < n; q0 += 256 * stride) {
        unsigned e0 = 0, o0 = 0, e1 = 0, o1 = 0;
        const int stop = min(n, q0 + 256 * stride);
        if (oct_bytes != 0) {
          constexpr int kPoints = 4;             // cells first, then their four gathers
          for (int q = q0; q < stop; q += kPoints * stride) {
            I4 cv[kPoints];
#pragma unroll
            for (int u = 0; u < kPoints; ++u) cv[u] = gcells[min(q + u * stride, n - 1)];
            U2 w[kPoints];
#pragma unroll
            for (int u = 0; u < kPoints; ++u) {
              const int3 d = DepthIndexAny(cv[u].x, cv[u].y, cv[u].z, e, -wxy, -wxy, -wz);
              const int X = d.x + ox, Y = d.y + oy, Z = d.z + oz;
              const bool inside = q + u * stride < stop &&
                                  static_cast<unsigned>(X) < static_cast<unsigned>(O.qx) &&
                                  static_cast<unsigned>(Y) < static_cast<unsigned>(O.qy) &&
                                  static_cast<unsigned>(Z) < static_cast<unsigned>(O.qz);
              const unsigned offset = static_cast<unsigned>((Z * O.qy + Y) * O.qx + X) * 8u;
              w[u] = __builtin_bit_cast(U2, __builtin_amdgcn_raw_buffer_load_b64(
                                                oct, inside ? offset : kOutOfBuffer, 0, 0));
            }
#pragma unroll
            for (int u = 0; u < kPoints; ++u) {
              e0 += w[u].x & 0x00ff00ffu; o0 += (w[u].x >> 8) & 0x00ff00ffu;
              e1 += w[u].y & 0x00ff00ffu; o1 += (w[u].y >> 8) & 0x00ff00ffu;
            }
          }
        } else {
#pragma unroll 4
          for (int q = q0; q < stop; q += stride) {
            const int3 d = DepthIndex(cells[q], e, -P.wxy, -P.wxy, -P.wz);
            const int X = d.x + ox, Y = d.y + oy, Z = d.z + oz;
            const bool inside = static_cast<unsigned>(X) < static_cast<unsigned>(O.qx) &&
                                static_cast<unsigned>(Y) < static_cast<unsigned>(O.qy) &&
                                static_cast<unsigned>(Z) < static_cast<unsigned>(O.qz);
            // unconditional load from a valid offset, masked afterwards
            const uint2 w = O.cells[inside ? (static_cast<size_t>(Z) * O.qy + Y) * O.qx + X : 0];
            const unsigned lo = inside ? w.x : 0u, hi = inside ? w.y : 0u;
            e0 += lo & 0x00ff00ffu; o0 += (lo >> 8) & 0x00ff00ffu;
            e1 += hi & 0x00ff00ffu; o1 += (hi >> 8) & 0x00ff00ffu;
          }
        }
        sum[0] += e0 & 0xffffu; sum[2] += e0 >> 16;      // bytes 0, 2 of the low word
        sum[1] += o0 & 0xffffu; sum[3] += o0 >> 16;      // bytes 1, 3
        sum[4] += e1 & 0xffffu; sum[6] += e1 >> 16;
        sum[5] += o1 & 0xffffu; sum[7] += o1 >> 16;
      }
    } else if (dx <= 4 && !g_fast3d_byte_loads) {
#pragma unroll 2
      for (int q = first; q < P.n; q += stride) {
        const int3 d = DepthIndex(cells[q], e, -P.wxy, -P.wxy, -P.wz);
        const int ix0 = d.x + fx[0], ix1 = ix0 + dx;
        const bool okx0 = static_cast<unsigned>(ix0) < static_cast<unsigned>(L.nx);
        const bool okx1 = static_cast<unsigned>(ix1) < static_cast<unsigned>(L.nx);
        const int base = min(max(ix0, 0), L.nx - 1);   // first byte wanted (when any is)
        int ay[2], az[2];
        bool oky[2], okz[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int iy = d.y + fy[b], iz = d.z + fz[b];
          oky[b] = static_cast<unsigned>(iy) < static_cast<unsigned>(L.ny);
          okz[b] = static_cast<unsigned>(iz) < static_cast<unsigned>(L.nz);
          ay[b] = oky[b] ? iy * row : 0;
          az[b] = okz[b] ? iz * slab : 0;
        }
        uint2 w[4];
        unsigned j0[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {     // unconditional loads from in-range addresses
          const unsigned a = static_cast<unsigned>(az[(k >> 1) & 1] + ay[k & 1] + base);
          j0[k] = a & 3u;
          w[k] = *reinterpret_cast<const uint2*>(cells8 + (a & ~3u));
        }
        const unsigned j1 = static_cast<unsigned>(ix1 - base);   // 0 .. 4 when okx1
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const unsigned b0 = j0[k], b1 = j0[k] + j1;             // byte indices, < 8
          const unsigned v0 = ((b0 & 4u) ? w[k].y : w[k].x) >> (8u * (b0 & 3u)) & 0xffu;
          const unsigned v1 = ((b1 & 4u) ? w[k].y : w[k].x) >> (8u * (b1 & 3u)) & 0xffu;
          const bool okyz = oky[k & 1] && okz[(k >> 1) & 1];
          sum[2 * k] += (okx0 && okyz) ? v0 : 0u;
          sum[2 * k + 1] += (okx1 && okyz) ? v1 : 0u;
        }
      }
    } else {
#pragma unroll 2
    for (int q = first; q < P.n; q += stride) {
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
    }
  }
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
  int fam_partial[4][8][8];    // [wave][member][child]
  int fam_total[8][8];
  Node3D fam[8];
  float score[8];
  float low_prob[kLowChunk];
  Node3D next;       // dive: the child the descent continues with
  int has_next;
  // Work counters of this block (flushed once by FlushWork3D: a global atomic pair per node
  // was most of the 2D wave stage's time, profiles/r02_c3_wave_atomics.txt).
  unsigned scored, expanded;
};

__device__ __forceinline__ void InitWork3D(ExpandShared* sh) {
  if (threadIdx.x == 0) { sh->scored = 0; sh->expanded = 0; }
  __syncthreads();
}
__device__ __forceinline__ void FlushWork3D(ExpandShared* sh, Counters3* __restrict__ counters) {
  __syncthreads();
  if (threadIdx.x == 0 && sh->expanded) {
    atomicAdd(&counters->scored[blockIdx.x & 15], static_cast<unsigned long long>(sh->scored));
    atomicAdd(&counters->expanded[blockIdx.x & 15], static_cast<unsigned long long>(sh->expanded));
  }
}

// Expansion of one node by the whole block (see above).  dive = 0: children that can still
// matter are appended to `out`; dive = 1: the best child is left in sh->next.
// The part of an expansion after the child sums: scores, ranks, then the children that can
// still matter (or, at the leaves, the low-resolution verification).  `totals` = the eight
// integer child sums of `nd` (any memory every thread can read).
__device__ __forceinline__ void FinishExpand3D(const Fast3DProblem& P, const Node3D& nd,
                                               const int* totals, float best, int dive, int strict,
                                               const List3& out, const List3& leaves,
                                               Counters3* __restrict__ counters, int sub_id,
                                               ExpandShared* sh) {
  if (threadIdx.x == 0) sh->has_next = 0;
  {
    const int child_depth = nd.level - 1;
    const int half = 1 << child_depth;
    const bool vx = nd.ox + half <= P.wxy, vy = nd.oy + half <= P.wxy, vz = nd.oz + half <= P.wz;
    __syncthreads();
    if (threadIdx.x < 8) {
      const int k = threadIdx.x;
      const bool valid = (!(k & 1) || vx) && (!(k & 2) || vy) && (!(k & 4) || vz);
      sh->score[k] = valid ? ToProbability(totals[k], P.n) : -1.f;
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
      sh->scored += static_cast<unsigned>(nvalid);
      sh->expanded += 1u;
    }
    auto make_child = [&](int k) {
      Node3D child = nd;
      child.family = 0;
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
          sh->score[0] = __uint_as_float(__hip_atomic_load(P.best_bits, __ATOMIC_RELAXED,
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
            if (!Push3(leaves, sub_id, atomicAdd(&leaves.counts[sub_id * kCountStride3], 1), rec))
              counters->overflow = 1;
            atomicMax(P.best_bits, __float_as_uint(sc));
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
        int slot = atomicAdd(&out.counts[sub_id * kCountStride3], m);
        bool leader = true;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          if (!(keep_mask >> k & 1)) continue;
          Node3D child = make_child(k);
          child.family = leader ? m : 0;       // the kept children of one parent: a family
          leader = false;
          if (!Push3(out, sub_id, slot, child)) counters->overflow = 1;
          ++slot;
        }
      }
    }
    __syncthreads();   // scratch reused by the next node; sh->next / has_next visible
  }
}

// Expansion of one node by the whole block (see above).  dive = 0: children that can still
// matter are appended to `out`; dive = 1: the best child is left in sh->next.
__device__ __forceinline__ void ExpandNode3D(const Fast3DProblem& P, const Node3D& nd, float best,
                                             int dive, int strict, const List3& out,
                                             const List3& leaves,
                                             Counters3* __restrict__ counters, int sub_id,
                                             ExpandShared* sh) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  int sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  ChildSums3D(P, nd, threadIdx.x, 256, sum);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int total = WaveSum(sum[k]);
    if (lane == 0) sh->partial[wave][k] = total;
  }
  __syncthreads();
  if (threadIdx.x < 8)
    sh->fam_total[0][threadIdx.x] = sh->partial[0][threadIdx.x] + sh->partial[1][threadIdx.x] +
                                    sh->partial[2][threadIdx.x] + sh->partial[3][threadIdx.x];
  FinishExpand3D(P, nd, sh->fam_total[0], best, dive, strict, out, leaves, counters, sub_id, sh);
}

// The child sums of a FAMILY (round 3): the kept children of one parent are expanded by one
// block.  What bounded the per-node expansion (profiles/r02e_c5_pmc_*): 4.9 GB of 128-byte lines
// per launch for 1.35 GB of oct words -- every (node, point) lookup fetched a line of its own,
// and every node re-read the scan's 16-byte cell records.  Members of a family read oct words
// `half` cells apart: the two x positions share a line, the cell record and its depth index
// are computed once per point for all members, and up to eight gathers per point are in
// flight together.  Returns false (nothing summed) when the level has no oct grid.
// The family loop for a compile-time family size F (round 4).  Until the end of round 3 the
// member loop `if (m >= fam || !(live >> m & 1)) continue;` gave every member a basic block of
// its own -- flat_load_dwordx2 with 64-bit address arithmetic, closed by s_waitcnt vmcnt(0)
// lgkmcnt(0): ONE gather in flight per wavefront where the comment above promises eight.  Now
// the loop body is branch-free: oct words come through a buffer resource (32-bit offsets; dead
// members and cells outside the grid use an out-of-range offset, which reads 0 and fetches
// nothing), so the F gathers of a point -- 2 F with the unroll -- go out back to back.
template <int F>
__device__ __forceinline__ void FamilyLoop3D(__amdgpu_buffer_rsrc_t oct, const OctDesc& O,
                                             const I4 CMX_GLOBAL* __restrict__ cells, int n,
                                             int e, int wxy, int wz, const int (&mx)[8],
                                             const int (&my)[8], const int (&mz)[8],
                                             unsigned live, int first, int stride,
                                             ExpandShared* sh) {
  typedef unsigned U2 __attribute__((ext_vector_type(2)));
  constexpr int kPoints = 2;          // points per iteration: 2 F gathers in flight
  // (n <= 256 * stride, FamilySums3D checks it: a lane adds at most 255 per point, so the
  // packed 16-bit halves cannot overflow and are widened once, after the loop)
  unsigned acc[F][4];
#pragma unroll
  for (int m = 0; m < F; ++m) acc[m][0] = acc[m][1] = acc[m][2] = acc[m][3] = 0;
  for (int q = first; q < n; q += kPoints * stride) {
    I4 cv[kPoints];
#pragma unroll
    for (int u = 0; u < kPoints; ++u) cv[u] = cells[min(q + u * stride, n - 1)];
    U2 w[kPoints][F];
#pragma unroll
    for (int u = 0; u < kPoints; ++u) {
      const bool in_cloud = q + u * stride < n;
      const int3 d = DepthIndexAny(cv[u].x, cv[u].y, cv[u].z, e, -wxy, -wxy, -wz);
#pragma unroll
      for (int m = 0; m < F; ++m) {
        const int X = d.x + mx[m], Y = d.y + my[m], Z = d.z + mz[m];
        const bool inside = in_cloud && (live >> m & 1) &&
                            static_cast<unsigned>(X) < static_cast<unsigned>(O.qx) &&
                            static_cast<unsigned>(Y) < static_cast<unsigned>(O.qy) &&
                            static_cast<unsigned>(Z) < static_cast<unsigned>(O.qz);
        const unsigned offset = static_cast<unsigned>((Z * O.qy + Y) * O.qx + X) * 8u;
        w[u][m] = __builtin_bit_cast(U2, __builtin_amdgcn_raw_buffer_load_b64(
                                             oct, inside ? offset : kOutOfBuffer, 0, 0));
      }
    }
#pragma unroll
    for (int u = 0; u < kPoints; ++u) {
#pragma unroll
      for (int m = 0; m < F; ++m) {
        acc[m][0] += w[u][m].x & 0x00ff00ffu; acc[m][1] += (w[u][m].x >> 8) & 0x00ff00ffu;
        acc[m][2] += w[u][m].y & 0x00ff00ffu; acc[m][3] += (w[u][m].y >> 8) & 0x00ff00ffu;
      }
    }
  }
  // Wave totals of the live members' eight child sums, left in sh->fam_partial[wave][m][k]
  // (the caller adds the four waves after a barrier).
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int m = 0; m < F; ++m) {
    if (!(live >> m & 1)) continue;                   // block-uniform
    const int s8[8] = {static_cast<int>(acc[m][0] & 0xffffu), static_cast<int>(acc[m][1] & 0xffffu),
                       static_cast<int>(acc[m][0] >> 16),     static_cast<int>(acc[m][1] >> 16),
                       static_cast<int>(acc[m][2] & 0xffffu), static_cast<int>(acc[m][3] & 0xffffu),
                       static_cast<int>(acc[m][2] >> 16),     static_cast<int>(acc[m][3] >> 16)};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int total = WaveSum(s8[k]);
      if (lane == 0) sh->fam_partial[wave][m][k] = total;
    }
  }
}

__device__ __forceinline__ bool FamilySums3D(const Fast3DProblem& P, const Node3D* fam_nodes,
                                             int fam, unsigned live, int first, int stride,
                                             ExpandShared* sh) {
  const int child_depth = fam_nodes[0].level - 1;
  const int half = 1 << child_depth;
  const int e = max(0, child_depth - P.full_resolution_depth + 1);
  const Brick L = P.level[child_depth];
  const OctDesc O = P.oct[child_depth];
  if (O.cells == nullptr || g_fast3d_byte_loads) return false;
  if ((((fam_nodes[0].ox + half) >> e) - (fam_nodes[0].ox >> e)) != O.s) return false;
  const unsigned long long oct_bytes = OctBytes(O);
  if (oct_bytes == 0) return false;                 // (the per-node path has plain loads)
  const __amdgpu_buffer_rsrc_t oct = UniformBuffer(O.cells, oct_bytes);
  const auto* cells = AsGlobal(
      reinterpret_cast<const I4*>(P.cells + static_cast<size_t>(fam_nodes[0].scan) * P.n));
  if (P.n > 256 * stride) return false;            // (packed sums: see FamilyLoop3D)
  int mx[8], my[8], mz[8];                          // block-uniform (the family lies in LDS)
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    const Node3D& nd = fam_nodes[min(m, fam - 1)];
    mx[m] = __builtin_amdgcn_readfirstlane((nd.ox >> e) - L.lo_x + O.s);
    my[m] = __builtin_amdgcn_readfirstlane((nd.oy >> e) - L.lo_y + O.s);
    mz[m] = __builtin_amdgcn_readfirstlane((nd.oz >> e) - L.lo_z + O.s);
  }
  const int n = P.n, wxy = P.wxy, wz = P.wz;
  switch (fam) {                                    // block-uniform
    case 2: FamilyLoop3D<2>(oct, O, cells, n, e, wxy, wz, mx, my, mz, live, first, stride, sh); break;
    case 3: FamilyLoop3D<3>(oct, O, cells, n, e, wxy, wz, mx, my, mz, live, first, stride, sh); break;
    case 4: FamilyLoop3D<4>(oct, O, cells, n, e, wxy, wz, mx, my, mz, live, first, stride, sh); break;
    case 5: FamilyLoop3D<5>(oct, O, cells, n, e, wxy, wz, mx, my, mz, live, first, stride, sh); break;
    case 6: FamilyLoop3D<6>(oct, O, cells, n, e, wxy, wz, mx, my, mz, live, first, stride, sh); break;
    case 7: FamilyLoop3D<7>(oct, O, cells, n, e, wxy, wz, mx, my, mz, live, first, stride, sh); break;
    default: FamilyLoop3D<8>(oct, O, cells, n, e, wxy, wz, mx, my, mz, live, first, stride, sh); break;
  }
  return true;
}

// (One wavefront per node for the levels above the leaves -- four times as many nodes in
// flight, 43 points per lane -- was measured and changed nothing: 5.26 vs 5.34 ms for 32 pairs;
// a single pair got slower, 0.45 vs 0.41 ms.  Removed.)
__global__ void __launch_bounds__(256)
Expand3DKernel(const Fast3DProblem* __restrict__ problems, List3 in, int strict, int affinity,
               int use_families, List3 out, List3 leaves, Counters3* __restrict__ counters) {
  __shared__ ExpandShared sh;
  InitWork3D(&sh);
  const int max_count = ListMax3(in);
  for (int i = blockIdx.x; i < max_count * kSubLists3; i += gridDim.x) {
    const int in_sub = i & (kSubLists3 - 1), j = i / kSubLists3;
    if (j >= min(in.counts[in_sub * kCountStride3], in.sub_capacity)) continue;   // block-uniform
    // Children go to a sub-list derived from the node's slot, not from the block: the
    // survivors of a search cluster in a few subtrees, and appending them to their
    // parent's sub-list would leave the next level with one long list that a handful
    // of blocks walk serially (measured: 0.9 us per node, chip idle).
    // (Workgroup b runs on XCD b % 8 and reads the sub-lists with sub % 8 == b % 8: the grid
    // is a multiple of 64 blocks.  With `affinity` the children stay on their parent's XCD.)
    // (Sending the children of 32 consecutive nodes to one sub-list, to keep neighbours
    // together in the next level, measured worse: 7.7 vs 7.2 ms for 32 pairs -- the lists of an
    // XCD then differ in length.)
    const Node3D* slot0 = in.nodes + static_cast<size_t>(in_sub) * in.sub_capacity + j;
    const int family = slot0->family;
    if (family == 0) continue;            // a later member: its leader's block takes it (uniform)
    const int fam = min(family, min(in.counts[in_sub * kCountStride3], in.sub_capacity) - j);
    const Fast3DProblem& P = problems[slot0->problem];
    // The bound moves while this kernel runs: ONE thread reads it and the block shares that
    // value.  (Every thread reading it for itself let some threads skip a node that the
    // others expanded -- barriers of different nodes then met, children were summed from a
    // mixture of two nodes, and about one search in fifty returned a leaf that does not
    // exist.  Present since round 1; found with tools/stress_fast3d.py.)
    __syncthreads();
    if (threadIdx.x == 0)
      sh.score[0] = __uint_as_float(
          __hip_atomic_load(P.best_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    if (threadIdx.x < fam) sh.fam[threadIdx.x] = slot0[threadIdx.x];
    __syncthreads();
    const float best = sh.score[0];
    unsigned live = 0;
    for (int m = 0; m < fam; ++m)
      if (!(strict ? !(sh.fam[m].score > best) : (sh.fam[m].score < best))) live |= 1u << m;
    if (live == 0) continue;                                              // block-uniform
    bool summed = false;
    if (fam > 1 && use_families) {
      summed = FamilySums3D(P, sh.fam, fam, live, threadIdx.x, 256, &sh);
      if (summed) {
        __syncthreads();
        if (threadIdx.x < 64) {
          const int m = threadIdx.x >> 3, k = threadIdx.x & 7;
          sh.fam_total[m][k] = sh.fam_partial[0][m][k] + sh.fam_partial[1][m][k] +
                               sh.fam_partial[2][m][k] + sh.fam_partial[3][m][k];
        }
        __syncthreads();
      }
    }
    for (int m = 0; m < fam; ++m) {
      if (!(live >> m & 1)) continue;
      // (each member's children go to a sub-list of their own, as single nodes' did)
      const int sub_m = affinity ? (in_sub & 7) | ((((in_sub >> 3) * 5 + j + m) & 7) << 3)
                                 : (in_sub * 17 + j + m) & (kSubLists3 - 1);
      const Node3D nd = sh.fam[m];
      if (summed)
        FinishExpand3D(P, nd, sh.fam_total[m], best, 0, strict, out, leaves, counters, sub_m, &sh);
      else
        ExpandNode3D(P, nd, best, 0, strict, out, leaves, counters, sub_m, &sh);
    }
  }
  FlushWork3D(&sh, counters);
}

// Greedy descents (always the best child) from the seeds, one block per seed, all levels in
// one launch: the verified leaf scores bound the search that follows.
// grid (kSeeds3, problems)
__global__ void __launch_bounds__(256)
Dive3DKernel(const Fast3DProblem* __restrict__ problems, List3 leaves,
             Counters3* __restrict__ counters) {
  __shared__ ExpandShared sh;
  const Fast3DProblem& P = problems[blockIdx.y];
  const List3 seeds{P.seeds, P.seed_count, kSeeds3};
  if (static_cast<int>(blockIdx.x) >= min(seeds.counts[0], seeds.sub_capacity)) return;
  InitWork3D(&sh);
  Node3D nd = seeds.nodes[blockIdx.x];
  const int sub_id = (blockIdx.x + 7 * blockIdx.y) & (kSubLists3 - 1);
  while (nd.level >= 1) {
    ExpandNode3D(P, nd, 0.f, 1, 0, seeds, leaves, counters, sub_id, &sh);
    if (!sh.has_next) break;
    nd = sh.next;
    __syncthreads();   // everyone has read sh.next before the next expansion resets it
  }
  FlushWork3D(&sh, counters);
}

struct Best3 {
  float score;
  int scan, ox, oy, oz;
  float low_resolution_score;
  int found, ties;
};

// Among the recorded leaves with the best score, the one the reference's
// depth-first search meets first (see fast_2d.hip SelectBestKernel).
// grid (problems): block p looks at the leaves of problem p only.
// Block p also PUBLISHES problem p's record -- and block 0 the counters and the problems' states,
// complete since the kernels before this one -- into the caller's pinned mirror of `misc` (mapped
// into the device's address space): no copy kernel behind the last launch of the chain.
// misc_host == nullptr: the host fetches `misc` itself.
__global__ void __launch_bounds__(1024)
SelectBest3DKernel(List3 leaves, const Fast3DProblem* __restrict__ problems,
                   Best3* __restrict__ results, const unsigned* __restrict__ misc_dev,
                   unsigned* __restrict__ misc_host, int counters_words, int state_word0,
                   int state_words, int best_word0) {
  __shared__ unsigned best_coarse;
  __shared__ unsigned long long best_key[2];
  __shared__ int ties;
  const int problem = blockIdx.x;
  Best3* out = results + problem;
  const int total = ListMax3(leaves) * kSubLists3;
  const unsigned best_bits = *problems[problem].best_bits;
  if (threadIdx.x == 0) {
    best_coarse = 0; ties = 0; best_key[0] = ~0ull; best_key[1] = ~0ull;
    Best3 b{};
    *out = b;
  }
  __syncthreads();
  auto leaf_at = [&](int i, Node3D* nd) {
    const int sub = i & (kSubLists3 - 1), j = i / kSubLists3;
    if (j >= min(leaves.counts[sub * kCountStride3], leaves.sub_capacity)) return false;
    *nd = leaves.nodes[static_cast<size_t>(sub) * leaves.sub_capacity + j];
    return nd->problem == problem;
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
  if (misc_host == nullptr) return;
  __threadfence();
  __syncthreads();
  const auto publish = [&](int word0, int words) {
    for (int i = threadIdx.x; i < words; i += blockDim.x)
      misc_host[word0 + i] =
          __hip_atomic_load(&misc_dev[word0 + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  constexpr int kBestWords = static_cast<int>(sizeof(Best3) / sizeof(unsigned));
  publish(best_word0 + problem * kBestWords, kBestWords);
  if (problem == 0) {
    publish(0, counters_words);
    publish(state_word0, state_words);
  }
}

// depth == 1: lowest-resolution candidates are the leaves; they are verified
// in descending score order (BranchAndBound at candidate_depth 0, :382-402).
// Rare configuration; handled by treating every candidate as a leaf group of
// one through the generic expansion of a virtual parent is not possible, so a
// dedicated wave-per-candidate pass records every passing candidate.
__global__ void __launch_bounds__(256)
VerifyCoarseLeaves3DKernel(const Fast3DProblem* __restrict__ problems, List3 leaves,
                           Counters3* __restrict__ counters) {
  const Fast3DProblem& P = problems[blockIdx.y];
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
      if (!Push3(leaves, sub_id, atomicAdd(&leaves.counts[sub_id * kCountStride3], 1), rec)) counters->overflow = 1;
      atomicMax(P.best_bits, __float_as_uint(nd.score));
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
  std::vector<std::unique_ptr<DeviceBrick>> octs;   // per child level (see OctDesc)
  std::vector<OctDesc> oct_desc;
  DeviceBrick low;
  DeviceBrick high;                                 // raw uint16 grid (Ceres refinement)
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

// The same two functions for the yaw sweep of a search (tens of angles per pair, hundreds of
// pairs per node): no allocation per angle, no integer division per bucket, the submap's norm
// computed once.  Same operations on the same operands in the same order: same floats.
struct YawSweep {
  const std::vector<float>& submap;
  const std::vector<float>& scan;
  float submap_norm;
  std::vector<float> rotated;
  YawSweep(const std::vector<float>& submap_histogram, const std::vector<float>& scan_histogram)
      : submap(submap_histogram), scan(scan_histogram),
        submap_norm(std::sqrt(Dot(submap_histogram, submap_histogram))),
        rotated(scan_histogram.size()) {}
  float Score(float angle) {
    const int size = static_cast<int>(scan.size());
    if (size != 0) {
      const float rotate_by_buckets =
          static_cast<float>(static_cast<double>(-angle * static_cast<float>(size)) / M_PI);
      int full_buckets = static_cast<int>(std::lround(rotate_by_buckets - 0.5f));
      const float fraction = rotate_by_buckets - full_buckets;
      while (full_buckets < 0) full_buckets += size;
      int i0 = full_buckets % size;
      for (int i = 0; i != size; ++i) {
        const int i1 = i0 + 1 == size ? 0 : i0 + 1;
        rotated[i] = fraction * scan[i1] + (1.f - fraction) * scan[i0];
        i0 = i1;
      }
    }
    const float scan_norm = std::sqrt(Dot(rotated, rotated));
    const float normalization = scan_norm * submap_norm;
    if (normalization < 1e-3f) return 1.f;
    return Dot(submap, rotated) / normalization;
  }
};

// One (node, submap) search of a batch: MatchWithSearchParameters' arguments (:172-198).
struct Search3D {
  const Fast3DMatcher* m;
  int wxy, wz;
  double angular_search_window;
  h3::Rigid node, submap;
  float min_score;
};

void Match3DMany(const Search3D* searches, int num, const cmx_node_data3d& data, int32_t* found,
                 cmx_result3d* results, cmx_match_stats* stats);

// Host side of one search: the yaw pre-filter and the candidate lattice.
struct Prepared3D {
  std::vector<h3::Q> pose_q, scan_q;
  std::vector<float> rotational_score;
  h3::V3 pose_t;
  int S = 0;
  long long ncx = 0, ncz = 0, per_scan = 0, total = 0;
  size_t scan_base = 0, coarse_base = 0;
};

void Match3D(const Fast3DMatcher& m, int wxy, int wz, double angular_search_window,
             const h3::Rigid& node, const h3::Rigid& submap, const cmx_node_data3d& data,
             float min_score, int32_t* found, cmx_result3d* result, cmx_match_stats* stats) {
  const Search3D one{&m, wxy, wz, angular_search_window, node, submap, min_score};
  Match3DMany(&one, 1, data, found, result, stats);
}

// `num` searches of one node's data in ONE chain of launches: every kernel takes the array of
// problems (blockIdx.y, or the index its nodes carry), frontier and leaf lists are shared.  All
// searches must live on the same device.  The rare cases that need a second look at one
// search -- a frontier overflow, distinct leaves tied for the best score -- are repeated one
// search at a time (num == 1 owns the overflow retry and the exact tie resolution).
void Match3DMany(const Search3D* searches, int num, const cmx_node_data3d& data, int32_t* found,
                 cmx_result3d* results, cmx_match_stats* stats) {
  CMX_REQUIRE(searches && num >= 1 && found && results, "null output");
  CMX_REQUIRE(data.high_resolution_point_cloud && data.num_high_resolution_points >= 1,
              "empty high-resolution point cloud");
  CMX_REQUIRE(data.low_resolution_point_cloud && data.num_low_resolution_points >= 1,
              "empty low-resolution point cloud");
  // Debug switch host_trace: wall-clock of the host phases (tools only).
  const bool host_trace = Debug().host_trace != 0;
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
  const int n = data.num_high_resolution_points, n_low = data.num_low_resolution_points;
  const float* hi = data.high_resolution_point_cloud;
  const int device = searches[0].m->device;
  int max_depth = 0, min_depth = kMaxDepth;
  for (int p = 0; p < num; ++p) {
    const Fast3DMatcher& m = *searches[p].m;
    CMX_REQUIRE(m.device == device, "the searches of a batch must share a device");
    CMX_REQUIRE(data.histogram_size == static_cast<int>(m.histogram.size()),
                "histogram size %d does not match the submap's %d", data.histogram_size,
                static_cast<int>(m.histogram.size()));
    CMX_REQUIRE(searches[p].wxy >= 0 && searches[p].wz >= 0 && searches[p].wxy < (1 << 20) &&
                    searches[p].wz < (1 << 20),
                "bad search window");
    max_depth = std::max(max_depth, m.options.branch_and_bound_depth);
    min_depth = std::min(min_depth, m.options.branch_and_bound_depth);
    found[p] = 0;
  }
  CMX_REQUIRE(data.histogram_size == 0 || data.rotational_scan_matcher_histogram != nullptr,
              "null histogram");
  if (num > 1 && min_depth < 2) {     // depth-1 stacks take the leaf-verification path: one by one
    cmx_match_stats total{};
    for (int p = 0; p < num; ++p) {
      cmx_match_stats st{};
      Match3DMany(searches + p, 1, data, found + p, results + p, &st);
      total.candidates_scored += st.candidates_scored; total.coarse_candidates += st.coarse_candidates;
      total.nodes_expanded += st.nodes_expanded; total.num_scans += st.num_scans;
      total.device_ms += st.device_ms; total.dominant_kernel_ms += st.dominant_kernel_ms;
      total.expansion_ms += st.expansion_ms; total.expansion_nodes += st.expansion_nodes;
      total.expansion_lookups += st.expansion_lookups;
      total.expansion_launches += st.expansion_launches;
    }
    if (stats) *stats = total;
    return;
  }

  // GenerateDiscreteScans (:246-295), host part.
  float max_point = 0.f;
  for (int i = 0; i < n; ++i)
    max_point = std::max(h3::Norm({hi[3 * i], hi[3 * i + 1], hi[3 * i + 2]}), max_point);
  const std::vector<float> scan_hist(
      data.rotational_scan_matcher_histogram,
      data.rotational_scan_matcher_histogram + data.histogram_size);
  const double* g = data.gravity_alignment;   // w, x, y, z
  const double n2 = (g[1] * g[1] + g[3] * g[3]) + (g[2] * g[2] + g[0] * g[0]);
  const h3::Q g_inv{static_cast<float>(g[0] / n2), static_cast<float>(-g[1] / n2),
                    static_cast<float>(-g[2] / n2), static_cast<float>(-g[3] / n2)};
  std::vector<Prepared3D> prep(num);
  size_t scans_total = 0, coarse_total = 0;
  long long max_total = 0;
  cmx_match_stats st{};
  for (int p = 0; p < num; ++p) {
    const Search3D& q = searches[p];
    const Fast3DMatcher& m = *q.m;
    Prepared3D& pr = prep[p];
    const float max_scan_range = std::max(max_point, 3.f * m.resolution);
    const float kSafetyMargin = 1.f - 1e-2f;
    const float step =
        kSafetyMargin * std::acos(1.f - (m.resolution * (m.resolution * 1.f)) /
                                            (2.f * (max_scan_range * (max_scan_range * 1.f))));
    const int angular_window_size = static_cast<int>(std::lround(q.angular_search_window / step));
    CMX_REQUIRE(angular_window_size >= 0 && angular_window_size < (1 << 20), "bad angular window");
    const h3::Rigid node_to_submap = h3::Mul(h3::InverseRigid(q.submap), q.node);
    const float initial_angle = h3::GetYaw(h3::Mul(node_to_submap.q, g_inv));
    YawSweep sweep(m.histogram, scan_hist);
    for (int rz = -angular_window_size; rz <= angular_window_size; ++rz) {
      const float angle = rz * step;
      const float sc = sweep.Score(initial_angle + angle);
      if (sc < m.options.min_rotational_score) continue;
      pr.pose_q.push_back(h3::Mul(h3::Mul(h3::Inverse(q.submap.q),
                                          h3::FromAngleAxisVector({0.f, 0.f, angle})),
                                  q.node.q));
      pr.rotational_score.push_back(sc);
    }
    pr.S = static_cast<int>(pr.pose_q.size());
    pr.pose_t = node_to_submap.t;
    st.num_scans += pr.S;
    // Lowest-resolution candidates (:297-330).
    const int depth = m.options.branch_and_bound_depth;
    const int step_cells = 1 << (depth - 1);
    pr.ncx = (2ll * q.wxy + step_cells) / step_cells;
    pr.ncz = (2ll * q.wz + step_cells) / step_cells;
    pr.per_scan = pr.ncx * pr.ncx * pr.ncz;
    pr.total = pr.per_scan * pr.S;
    CMX_REQUIRE(pr.total < (1ll << 30), "search too large: %lld lowest-resolution candidates",
                pr.total);
    pr.scan_base = scans_total;
    pr.coarse_base = coarse_total;
    scans_total += pr.S;
    coarse_total += static_cast<size_t>(pr.total);
    max_total = std::max(max_total, pr.total);
    // GetPoseFromCandidate (:369-375): Translation(res * offset) * pose renormalises
    // the rotation; Identity * q is exact, the normalisation is not.
    pr.scan_q.resize(pr.S);
    for (int s = 0; s < pr.S; ++s)
      pr.scan_q[s] = h3::Normalized(h3::Mul(h3::Q{1.f, 0.f, 0.f, 0.f}, pr.pose_q[s]));
  }
  if (scans_total == 0) {
    if (stats) *stats = st;
    return;
  }
  CMX_REQUIRE(coarse_total < (size_t(1) << 31) && scans_total * n < (size_t(1) << 31),
              "batch too large");

  lap("prepare");
  WorkspaceLease ws(device);
  {
    const int byte_loads = Debug().fast3d_byte_loads ? 1 : 0;
    static int uploaded = -1;              // (tools only: not meant to be toggled concurrently)
    if (uploaded != byte_loads) {
      CMX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_fast3d_byte_loads), &byte_loads, sizeof(int)));
      uploaded = byte_loads;
    }
  }
  // Everything the call uploads lives in ONE device buffer with ONE pinned mirror, in the order
  //   high-resolution cloud | low-resolution cloud | per-scan poses | misc (counters, problems, states)
  // and goes up in one transfer (they were four copy kernels in a chain of launches that is
  // latency from end to end); the Best3 records behind `misc` are only ever written on the device.
  const auto align256 = [](size_t bytes) { return (bytes + 255) & ~size_t(255); };
  const size_t up_low = align256(3 * sizeof(float) * static_cast<size_t>(n));
  const size_t up_q = up_low + align256(3 * sizeof(float) * static_cast<size_t>(n_low));
  const size_t up_misc = up_q + align256(3 * sizeof(float4) * scans_total);
  // [Counters3 | problems | per problem: best bits, seed count | Best3 per problem]
  const size_t off_problems = sizeof(Counters3);
  const size_t off_state = off_problems + sizeof(Fast3DProblem) * num;
  const size_t off_best = off_state + sizeof(unsigned) * 2 * num;
  const size_t misc_bytes = off_best + sizeof(Best3) * num;
  static_assert(sizeof(Counters3) % 8 == 0 && sizeof(Fast3DProblem) % 8 == 0, "alignment");
  static_assert(sizeof(Best3) % sizeof(unsigned) == 0, "Best3 is copied by dwords");
  char* d_up = static_cast<char*>(ws->dev[0].Reserve(up_misc + misc_bytes));
  char* h_up = static_cast<char*>(ws->pinned[0].Reserve(up_misc + misc_bytes));
  float* d_hi = reinterpret_cast<float*>(d_up);
  float* d_low = reinterpret_cast<float*>(d_up + up_low);
  // per scan: pose rotation | rotation of GetPoseFromCandidate | translation + resolution
  float4* d_pose_q = reinterpret_cast<float4*>(d_up + up_q);
  float4* d_scan_q = d_pose_q + scans_total;
  float4* d_pose_t = d_scan_q + scans_total;
  int4* d_cells = ws->dev[3].ReserveAs<int4>(scans_total * n);
  float* d_coarse = ws->dev[4].ReserveAs<float>(coarse_total);
  // The debug switch frontier_capacity shrinks the frontier buffers (tests only): overflow ->
  // strict retry.
  const int cap_req = Debug().frontier_capacity;
  const int kFrontierCapacity =
      cap_req >= kSubLists3 ? std::min(cap_req, 1 << 21) / kSubLists3 * kSubLists3 : 1 << 21;
  const int kLeafCapacity = 1 << 18;
  Node3D* d_front[2] = {ws->dev[5].ReserveAs<Node3D>(kFrontierCapacity),
                        ws->dev[6].ReserveAs<Node3D>(kFrontierCapacity)};
  Node3D* d_leaves = ws->dev[7].ReserveAs<Node3D>(kLeafCapacity);
  Node3D* d_seeds = ws->dev[8].ReserveAs<Node3D>(static_cast<size_t>(kSeeds3) * num);
  char* d_misc = d_up + up_misc;
  Counters3* d_counters = reinterpret_cast<Counters3*>(d_misc);
  Fast3DProblem* d_problems = reinterpret_cast<Fast3DProblem*>(d_misc + off_problems);
  unsigned* d_state = reinterpret_cast<unsigned*>(d_misc + off_state);   // [num][2]
  Best3* d_best = reinterpret_cast<Best3*>(d_misc + off_best);

  float4* h_q = reinterpret_cast<float4*>(h_up + up_q);
  char* h_misc = h_up + up_misc;
  Counters3* h_counters = reinterpret_cast<Counters3*>(h_misc);
  Fast3DProblem* h_problems = reinterpret_cast<Fast3DProblem*>(h_misc + off_problems);
  unsigned* h_state = reinterpret_cast<unsigned*>(h_misc + off_state);
  std::memset(h_misc, 0, misc_bytes);
  for (int p = 0; p < num; ++p) {
    const Search3D& q = searches[p];
    const Fast3DMatcher& m = *q.m;
    const Prepared3D& pr = prep[p];
    for (int s = 0; s < pr.S; ++s) {
      const size_t k = pr.scan_base + s;
      h_q[k] = make_float4(pr.pose_q[s].x, pr.pose_q[s].y, pr.pose_q[s].z, pr.pose_q[s].w);
      h_q[scans_total + k] =
          make_float4(pr.scan_q[s].x, pr.scan_q[s].y, pr.scan_q[s].z, pr.scan_q[s].w);
      h_q[2 * scans_total + k] = make_float4(pr.pose_t.x, pr.pose_t.y, pr.pose_t.z, m.resolution);
    }
    const float floor_score = std::max(q.min_score, 0.f);
    std::memcpy(&h_state[2 * p], &floor_score, sizeof(float));
    const int depth = m.options.branch_and_bound_depth;
    Fast3DProblem P{};
    for (int d = 0; d < depth; ++d) P.level[d] = m.levels[d]->desc;
    for (size_t d = 0; d < m.oct_desc.size(); ++d) P.oct[d] = m.oct_desc[d];
    P.depth = depth;
    P.full_resolution_depth = m.options.full_resolution_depth;
    P.low = m.low.desc;
    P.low_resolution = m.low_resolution;
    P.resolution = m.resolution;
    P.wxy = q.wxy; P.wz = q.wz;
    P.num_scans = pr.S; P.n = n; P.n_low = n_low;
    P.cells = d_cells + pr.scan_base * n;
    P.low_xyz = d_low;
    P.scan_q = d_scan_q + pr.scan_base;
    P.pose_tx = pr.pose_t.x; P.pose_ty = pr.pose_t.y; P.pose_tz = pr.pose_t.z;
    P.min_score = q.min_score;
    P.min_low_resolution_score = m.options.min_low_resolution_score;
    P.ncx = static_cast<int>(pr.ncx); P.ncy = static_cast<int>(pr.ncx);
    P.ncz = static_cast<int>(pr.ncz);
    P.coarse_score = d_coarse + pr.coarse_base;
    P.best_bits = d_state + 2 * p;
    P.seed_count = reinterpret_cast<int*>(d_state + 2 * p + 1);
    P.seeds = d_seeds + static_cast<size_t>(kSeeds3) * p;
    P.index = p;
    h_problems[p] = P;
  }
  // The high-resolution cloud only ever feeds integer sums (ScoreCandidates), which do
  // not depend on the order of the points.  Upload it sorted along a Morton curve: the
  // 64 points a wavefront gathers together then fall into neighbouring voxels, i.e. into
  // a handful of cache lines instead of 64 (the search is bound by that line traffic).
  float* h_hi = reinterpret_cast<float*>(h_up);
  {
    float lo3[3] = {hi[0], hi[1], hi[2]};
    for (int i = 1; i < n; ++i)
      for (int k = 0; k < 3; ++k) lo3[k] = std::min(lo3[k], hi[3 * i + k]);
    const float inv_cell = 1.f / (2.f * searches[0].m->resolution);
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
  // One upload from the pinned mirror (the caller's low-resolution cloud is copied there
  // first): a copy kernel while it is small (cmx_common.h: SmallCopyAsync).
  float* h_low = reinterpret_cast<float*>(h_up + up_low);
  std::memcpy(h_low, data.low_resolution_point_cloud, 3 * sizeof(float) * n_low);
  SmallCopyAsync(d_up, h_up, up_misc + off_best, true, ws->stream);

  auto front = [&](int stage) {
    return List3{d_front[stage & 1], d_counters->frontier[stage], kFrontierCapacity / kSubLists3};
  };
  const List3 leaf_list{d_leaves, d_counters->leaves, kLeafCapacity / kSubLists3};

  const bool dbg_sync = Debug().sync != 0;
  auto dbg = [&](const char* name) {
    if (!dbg_sync) return;
    fprintf(stderr, "[cmx sync] %s ...\n", name);
    CMX_HIP(hipStreamSynchronize(ws->stream));
  };
  dbg("uploads");
  lap("buffers+uploads");
  StageTrace trace(ws->stream);
  auto mark = [&](const char* name) { trace.Mark(name); };
  mark("begin");
  RecordEvent(ws->ev_begin, ws->stream);
  Discretize3DKernel<<<dim3(DivUp(n, 256), static_cast<unsigned>(scans_total)), 256, 0,
                       ws->stream>>>(d_hi, n, d_pose_q, d_pose_t, d_cells);
  dbg("discretize");
  mark("discretize");
  RecordEvent(ws->ev_k0, ws->stream);
  if (max_total <= 4096)
    ScoreCoarse3DBlockKernel<<<dim3(static_cast<unsigned>(max_total), num), 256, 0,
                               ws->stream>>>(d_problems);
  else
    ScoreCoarse3DKernel<<<dim3(std::min<long long>(8192, DivUp(max_total, 4)), num), 256, 0,
                          ws->stream>>>(d_problems);
  RecordEvent(ws->ev_k1, ws->stream);
  dbg("coarse");
  mark("coarse");

  const int blocks = 2048;
  // Batches: one problem's nodes stay on one XCD (debug switch fast3d_affinity overrides).
  const int affinity_override = Debug().fast3d_affinity;
  const int affinity = affinity_override ? affinity_override - 1 : (num >= 16 ? 1 : 0);
  // fast3d_no_families: every node of a family expanded on its own (A/B runs, parity tests)
  const int families = Debug().fast3d_no_families ? 0 : 1;
  int strict = 0, num_chunks = 1;
  const Best3* h_best = reinterpret_cast<const Best3*>(h_misc + off_best);
  for (;;) {
    if (max_depth == 1) {
      VerifyCoarseLeaves3DKernel<<<dim3(blocks, num), 256, 0, ws->stream>>>(d_problems, leaf_list,
                                                                            d_counters);
    } else {
      if (!strict) {
        // dive: greedy descents from the best lowest-resolution candidates give
        // a verified leaf score to bound the search with.
        SeedSelect3DKernel<<<num, 1024, 0, ws->stream>>>(d_problems);
        dbg("seed");
        mark("seed");
        Dive3DKernel<<<dim3(kSeeds3, num), 256, 0, ws->stream>>>(d_problems, leaf_list,
                                                                  d_counters);
        dbg("dive");
        mark("dive");
      }
      // The lowest-resolution candidates are searched in `num_chunks` interleaved subsets
      // (1 unless an earlier pass overflowed); later chunks profit from the bound the
      // earlier ones raised.
      for (int chunk = 0; chunk < num_chunks; ++chunk) {
        CMX_HIP(hipMemsetAsync(d_counters->frontier, 0, sizeof(d_counters->frontier),
                               ws->stream));
        Filter3DKernel<<<dim3(256, num), 256, 0, ws->stream>>>(
            d_problems, strict, chunk, num_chunks, affinity, front(0), d_counters);
        dbg("filter");
        mark("filter");
        int stage = 0;
        const bool timed = !strict && chunk == 0;      // statistics: the first pass
        if (timed) RecordEvent(ws->ev_x0, ws->stream);
        for (int child = max_depth - 2; child >= 0; --child, ++stage) {
          Expand3DKernel<<<blocks, 256, 0, ws->stream>>>(d_problems, front(stage), strict,
                                                         affinity, families, front(stage + 1),
                                                         leaf_list, d_counters);
          dbg("expand level");
          mark("expand");
        }
        if (timed) {
          RecordEvent(ws->ev_x1, ws->stream);
          st.expansion_launches = stage;
        }
      }
    }
    const bool direct = Debug().no_direct_results == 0;
    SelectBest3DKernel<<<num, 1024, 0, ws->stream>>>(
        leaf_list, d_problems, d_best, reinterpret_cast<const unsigned*>(d_misc),
        direct ? reinterpret_cast<unsigned*>(h_misc) : nullptr,
        static_cast<int>(sizeof(Counters3) / sizeof(unsigned)),
        static_cast<int>(off_state / sizeof(unsigned)), 2 * num,
        static_cast<int>(off_best / sizeof(unsigned)));
    mark("select");
    CMX_HIP(hipGetLastError());
    RecordEvent(ws->ev_end, ws->stream);
    if (!direct) SmallCopyAsync(h_misc, d_misc, misc_bytes, false, ws->stream);
    CMX_HIP(hipStreamSynchronize(ws->stream));
    trace.Report();
    lap("device");
    if (!strict) {
      // (the nodes the expansion kernel took off its lists and found at or above the bound)
      for (int k = 0; k < 16; ++k) st.expansion_nodes += static_cast<int64_t>(h_counters->expanded[k]);
      st.expansion_lookups = st.expansion_nodes * (static_cast<int64_t>(n + 63) / 64 * 64);
    }
    if (!h_counters->overflow || num > 1) break;
    // Something was dropped.  Retry pruning ties (strict) with the bound lowered by one
    // ulp so the best leaf is found again, over four times as many, smaller chunks.
    CMX_REQUIRE(num_chunks < (1 << 12), "branch-and-bound frontier overflow (search too wide)");
    if (strict) num_chunks *= 4;
    strict = 1;
    const float floor_score = std::max(searches[0].min_score, 0.f);
    unsigned floor_bits;
    std::memcpy(&floor_bits, &floor_score, sizeof(float));
    Counters3 reset{};
    std::memcpy(reset.scored, h_counters->scored, sizeof(reset.scored));
    std::memcpy(reset.expanded, h_counters->expanded, sizeof(reset.expanded));
    *h_counters = reset;
    h_state[0] = h_state[0] > floor_bits ? h_state[0] - 1 : floor_bits;
    h_state[1] = 0;
    CMX_HIP(hipMemcpyAsync(d_misc, h_misc, off_best, hipMemcpyHostToDevice, ws->stream));
  }
  st.coarse_candidates = static_cast<int64_t>(coarse_total);
  st.candidates_scored = static_cast<int64_t>(coarse_total);
  for (int k = 0; k < 16; ++k) {
    st.candidates_scored += h_counters->scored[k];
    st.nodes_expanded += h_counters->expanded[k];
  }
  float ms = 0.f;
  ms = ElapsedMs(ws->ev_begin, ws->ev_end);
  st.device_ms = ms;
  ms = ElapsedMs(ws->ev_k0, ws->ev_k1);
  st.dominant_kernel_ms = ms;
  if (st.expansion_launches > 0) {
    ms = ElapsedMs(ws->ev_x0, ws->ev_x1);
    st.expansion_ms = ms;
  }

  if (stats) *stats = st;
  if (num > 1 && h_counters->overflow) {
    // The shared lists dropped nodes: every search again on its own (num == 1 owns the retry).
    for (int p = 0; p < num; ++p) {
      cmx_match_stats again{};
      Match3DMany(searches + p, 1, data, found + p, results + p, &again);
      st.candidates_scored += again.candidates_scored;
      st.nodes_expanded += again.nodes_expanded;
      st.device_ms += again.device_ms;
    }
    if (stats) *stats = st;
    return;
  }

  // Leaves tied for the best score (lazily: one download of the leaf lists for the batch).
  std::vector<Node3D> all_leaves;
  std::vector<float> all_coarse;
  bool have_leaves = false;
  const auto leaves_of = [&](int p, unsigned score_bits) {
    if (!have_leaves) {
      have_leaves = true;
      // one strided copy: the first max-count slots of every sub-list
      int max_count = 0;
      for (int sub = 0; sub < kSubLists3; ++sub)
        max_count = std::max(max_count, std::min(h_counters->leaves[sub * kCountStride3], leaf_list.sub_capacity));
      if (max_count > 0) {
        std::vector<Node3D> rows(static_cast<size_t>(max_count) * kSubLists3);
        CMX_HIP(hipMemcpy2D(rows.data(), max_count * sizeof(Node3D), leaf_list.nodes,
                            leaf_list.sub_capacity * sizeof(Node3D), max_count * sizeof(Node3D),
                            kSubLists3, hipMemcpyDeviceToHost));
        for (int sub = 0; sub < kSubLists3; ++sub) {
          const int count = std::min(h_counters->leaves[sub * kCountStride3], leaf_list.sub_capacity);
          all_leaves.insert(all_leaves.end(), rows.begin() + static_cast<size_t>(sub) * max_count,
                            rows.begin() + static_cast<size_t>(sub) * max_count + count);
        }
      }
    }
    std::vector<Node3D> tied;
    for (const Node3D& nd : all_leaves) {
      unsigned bits;
      std::memcpy(&bits, &nd.score, sizeof(float));
      if (nd.problem == p && bits == score_bits) tied.push_back(nd);
    }
    return tied;
  };

  for (int p = 0; p < num; ++p) {
    const Fast3DMatcher& m = *searches[p].m;
    const Prepared3D& pr = prep[p];
    const long long total = pr.total;
    Best3 best = h_best[p];
    if (best.found && best.ties > 1) {
      // Exact tie resolution (see fast_2d.hip ResolveTies): repeat the reference's
      // std::sort of the lowest-resolution candidates (:352-353) and take the
      // tied leaf its depth-first search meets first.  The dive and the search
      // record the same leaf twice, so first check that distinct leaves tie.
      unsigned best_bits;
      std::memcpy(&best_bits, &best.score, sizeof(float));
      const std::vector<Node3D> tied = leaves_of(p, best_bits);
      bool distinct = false;
      for (const Node3D& nd : tied)
        distinct |= !(nd.scan == tied[0].scan && nd.ox == tied[0].ox && nd.oy == tied[0].oy &&
                      nd.oz == tied[0].oz);
      if (distinct) {
        if (all_coarse.empty()) {
          all_coarse.resize(coarse_total);
          CMX_HIP(hipMemcpy(all_coarse.data(), d_coarse, coarse_total * sizeof(float),
                            hipMemcpyDeviceToHost));
        }
        const float* scores = all_coarse.data() + pr.coarse_base;
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
    if (best.found && best.score > searches[p].min_score) {
      found[p] = 1;
      results[p].score = best.score;
      h3::Rigid pose;
      // Translation(res * offset) * scan.pose
      pose.t = {(pr.pose_t.x + 0.f) + m.resolution * static_cast<float>(best.ox),
                (pr.pose_t.y + 0.f) + m.resolution * static_cast<float>(best.oy),
                (pr.pose_t.z + 0.f) + m.resolution * static_cast<float>(best.oz)};
      pose.q = pr.scan_q[best.scan];
      results[p].pose_estimate = h3::ToPose(pose);
      results[p].rotational_score = pr.rotational_score[best.scan];
      results[p].low_resolution_score = best.low_resolution_score;
    }
  }
  lap("results");
  if (host_trace) fprintf(stderr, "[cmx host] Match3DMany(%d):%s us\n", num, host_report.c_str());
}

}  // namespace
}  // namespace cmx

struct cmx_fast3d {
  cmx::Fast3DMatcher impl;
};

namespace cmx {
// For sharded.hip: the device a 3D matcher's grids live on.
int Fast3DDevice(const cmx_fast3d* matcher) { return matcher->impl.device; }
// For ceres_3d.hip: the raw grids a 3D matcher keeps in HBM.
void Fast3DGrids(const cmx_fast3d* matcher, Brick* high, float* resolution, Brick* low,
                 float* low_resolution) {
  *high = matcher->impl.high.desc;
  *resolution = matcher->impl.resolution;
  *low = matcher->impl.low.desc;
  *low_resolution = matcher->impl.low_resolution;
}
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
    // The raw high-resolution values (level 0 of the stack is their 8-bit quantisation) stay
    // resident for the refinement that follows a match (cmx_fast3d_refine_batch).
    BuildBrickFromVoxels(*ws, voxels, num_voxels, 2, &m.high);
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
      CMX_HIP(hipMalloc(&level->mem, level->bytes + 16));   // (+16: aligned 8-byte reads of the last cells)
      b.cells = level->mem;
      level->desc = b;
      PrecomputeLevel3DKernel<<<DivUp(level->bytes, 256), 256, 0, ws->stream>>>(prev, b, shift,
                                                                               half ? 1 : 0);
      CMX_HIP(hipGetLastError());
      m.levels.push_back(std::move(level));
      last_width = next_width;
    }
    // Octs of every level that can be a child level (debug switch fast3d_no_oct: none, tests).
    {
      const bool build_octs = Debug().fast3d_no_oct == 0;
      const int depth = options->branch_and_bound_depth;
      m.oct_desc.assign(depth, OctDesc{nullptr, 0, 0, 0, 0});
      for (int i = 0; build_octs && i + 1 < depth; ++i) {
        const Brick L = m.levels[i]->desc;
        OctDesc O;
        O.s = 1 << std::min(i, options->full_resolution_depth - 1);
        O.qx = L.nx + O.s; O.qy = L.ny + O.s; O.qz = L.nz + O.s;
        const size_t count = static_cast<size_t>(O.qx) * O.qy * O.qz;
        if (count * sizeof(uint2) >= (size_t(1) << 32)) continue;     // 32-bit offsets elsewhere
        std::unique_ptr<DeviceBrick> mem(new DeviceBrick);
        mem->bytes = count * sizeof(uint2);
        CMX_HIP(hipMalloc(&mem->mem, mem->bytes));
        O.cells = static_cast<const uint2*>(mem->mem);
        BuildOct3DKernel<<<DivUp(count, 256), 256, 0, ws->stream>>>(
            L, O.s, static_cast<uint2*>(mem->mem), O.qx, O.qy, O.qz);
        CMX_HIP(hipGetLastError());
        m.oct_desc[i] = O;
        m.octs.push_back(std::move(mem));
      }
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
// reference runs one thread-pool task per pair; here the pairs of a node are ONE chain of
// launches (Match3DMany: every kernel indexes the search with blockIdx.y or through its nodes,
// frontier and leaf lists are shared), so a level of all searches is one launch instead of
// one short launch per search.  `node_poses[p]` / `submap_poses[p]`: the global poses of pair
// p (only their rotations are read where match_full_submap[p] != 0).
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
    const auto entry_time = std::chrono::steady_clock::now();
    // Match (:127-146) / MatchFullSubmap (:148-170) arguments of every pair, then one chain of
    // launches per device (Match3DMany).
    CMX_REQUIRE(data->high_resolution_point_cloud && data->num_high_resolution_points >= 1,
                "empty high-resolution point cloud");
    float max_point_distance = 0.f;
    for (int i = 0; i < data->num_high_resolution_points; ++i) {
      const float* p = data->high_resolution_point_cloud + 3 * i;
      max_point_distance = std::max(max_point_distance, h3::Norm({p[0], p[1], p[2]}));
    }
    std::vector<Search3D> searches(num_pairs);
    for (int p = 0; p < num_pairs; ++p) {
      const Fast3DMatcher& m = matchers[p]->impl;
      Search3D& q = searches[p];
      q.m = &m;
      q.min_score = min_scores[p];
      if (match_full_submap[p]) {
        const int window = (m.width_in_voxels + 1) / 2 +
                           static_cast<int>(std::lround(max_point_distance / m.resolution + 0.5f));
        q.wxy = q.wz = window;
        q.angular_search_window = M_PI;
        q.node.q = {static_cast<float>(node_poses[p].q[0]), static_cast<float>(node_poses[p].q[1]),
                    static_cast<float>(node_poses[p].q[2]), static_cast<float>(node_poses[p].q[3])};
        q.submap.q = {static_cast<float>(submap_poses[p].q[0]),
                      static_cast<float>(submap_poses[p].q[1]),
                      static_cast<float>(submap_poses[p].q[2]),
                      static_cast<float>(submap_poses[p].q[3])};
      } else {
        q.wxy = static_cast<int>(std::lround(m.options.linear_xy_search_window / m.resolution));
        q.wz = static_cast<int>(std::lround(m.options.linear_z_search_window / m.resolution));
        q.angular_search_window = m.options.angular_search_window;
        q.node = h3::FromPose(node_poses[p]);
        q.submap = h3::FromPose(submap_poses[p]);
      }
    }
    // The debug switch fast3d_batch caps the searches per chain (tools / tests; 1 = one by one).
    const int group = Debug().fast3d_batch > 0 ? Debug().fast3d_batch : 64;
    cmx_match_stats total{};
    std::vector<char> done(num_pairs, 0);
    for (int first = 0; first < num_pairs; ++first) {
      if (done[first]) continue;
      // the not yet searched pairs on this pair's device, `group` at a time
      std::vector<int> idx;
      for (int p = first; p < num_pairs && static_cast<int>(idx.size()) < group; ++p)
        if (!done[p] && searches[p].m->device == searches[first].m->device) idx.push_back(p);
      std::vector<Search3D> part(idx.size());
      std::vector<int32_t> part_found(idx.size(), 0);
      std::vector<cmx_result3d> part_results(idx.size());
      for (size_t k = 0; k < idx.size(); ++k) part[k] = searches[idx[k]];
      cmx_match_stats st{};
      Match3DMany(part.data(), static_cast<int>(part.size()), *data, part_found.data(),
                  part_results.data(), &st);
      for (size_t k = 0; k < idx.size(); ++k) {
        done[idx[k]] = 1;
        found[idx[k]] = part_found[k];
        if (part_found[k]) results[idx[k]] = part_results[k];
      }
      total.candidates_scored += st.candidates_scored;
      total.coarse_candidates += st.coarse_candidates;
      total.nodes_expanded += st.nodes_expanded;
      total.num_scans += st.num_scans;
      total.device_ms += st.device_ms;
      total.dominant_kernel_ms += st.dominant_kernel_ms;
      total.expansion_ms += st.expansion_ms;
      total.expansion_nodes += st.expansion_nodes;
      total.expansion_lookups += st.expansion_lookups;
      total.expansion_launches += st.expansion_launches;
    }
    if (stats) *stats = total;
    if (Debug().host_trace)
      fprintf(stderr, "[cmx host] cmx_fast3d_match_batch(%d): %.0f us\n", num_pairs,
              std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() -
                                                        entry_time).count());
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
