// FastCorrelativeScanMatcher2D on gfx950: precomputation-grid stack
// construction, scan preparation, lowest-resolution scoring and a batched
// branch and bound.
//
// Reference behaviour being replaced:
//   SM2/fast_correlative_scan_matcher_2d.cc:91-186   PrecomputationGrid2D / Stack
//   SM2/correlative_scan_matcher_2d.cc:73-127        ShrinkToFit / GenerateRotatedScans / DiscretizeScans
//   SM2/fast_correlative_scan_matcher_2d.cc:227-378  MatchWithSearchParameters, ScoreCandidates, BranchAndBound
// (SM2 = cartographer/mapping/internal/2d/scan_matching).
//
// Search schedule (any sound schedule returns the reference's best score):
//   1. score every lowest-resolution candidate (reference: :264-274);
//   2. "dive": greedy descents from the best few of them give a real leaf
//      score b0, a valid lower bound;
//   3. every lowest-resolution node whose upper bound reaches the bound is
//      searched depth-first by one workgroup, all workgroups sharing the
//      problem's bound through one atomic word;
//   4. among the leaves with the best score, the one the reference's
//      depth-first search meets first is returned (see SelectBestKernel).
//
// Lowest-resolution scoring ("phase planes").  Lowest-resolution candidates
// of one rotated scan sit on a lattice of pitch w = 2^(depth-1) cells, so for
// a given point p all of them read level cells with the same residue
// (phase) modulo w.  The level is therefore stored a second time as w*w small
// planes, plane(py,px)[J][I] = cell(I*w+px, J*w+py): ONE 64-byte plane holds
// everything a point contributes to all ~13x13 candidates of its scan.  Points
// are bucketed by the lattice block they fall in; within a bucket the lane ->
// candidate map is fixed, so a wave adds planes into registers (one coalesced
// 64 B load + one add per point for ALL candidates) and flushes once per
// bucket.  Out-of-grid lookups (60 % of the reference's reads here) cost
// nothing.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>

#include "scan_matching_2d.h"

namespace cmx {
namespace {

constexpr int kMaxBuckets = 4096;      // LDS histogram size of the point bucketing
constexpr int kMaxPlaneCells = 256;    // plane_i * plane_j
constexpr int kMaxPlaneWidth = 128;    // w; plane index fits 14 bits
constexpr int kMaxCoarsePerScan = 4096;  // lowest-resolution candidates per scan (plane kernel)
constexpr int kMaxAccCells = 12288;      // padded LDS accumulators of the plane kernel (48 KB)
constexpr int kSeedsPerProblem = 64;

// ---------------------------------------------------------------------------
// Precomputation stack
// ---------------------------------------------------------------------------

// Level 0: ComputeCellValue(1 - |cost|)  (SM2/fast_...2d.cc:107-108,163-169)
// with the per-grid cost table of mapping/value_conversion_tables.cc:29-51
// evaluated arithmetically (same f32 expression the table is built from).
__global__ void BuildLevel0Kernel(const uint16_t* __restrict__ cells, int count, float min_cc,
                                  float max_cc, uint8_t* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const unsigned v = cells[i] & 0x7fffu;
  float cost;
  if (v == 0) {
    cost = max_cc;
  } else {
    const float scale = (max_cc - min_cc) / 32766.f;
    cost = static_cast<float>(v) * scale + (min_cc - scale);
  }
  const float probability = 1.f - fabsf(cost);
  const float min_s = 1.f - max_cc, max_s = 1.f - min_cc;
  int value = LRoundF32((probability - min_s) * (255.f / (max_s - min_s)));
  value = min(max(value, 0), 255);
  out[i] = static_cast<uint8_t>(value);
}

// Level w from level w/2: a w x w window is the union of four (w/2) x (w/2)
// windows.  The u8 quantisation is monotone, so max-then-quantise (reference)
// equals quantise-then-max (here).  Windows entirely outside the grid read 0,
// which never wins because at least one of the four overlaps the grid.
__global__ void BuildLevelKernel(const uint8_t* __restrict__ prev, int pwx, int pwy, int half,
                                 uint8_t* __restrict__ out, int wx, int wy) {
  const int X = blockIdx.x * blockDim.x + threadIdx.x;
  const int Y = blockIdx.y;
  if (X >= wx) return;
  // (x0, y0) = (X - (w-1), Y - (w-1)); in the previous level's storage the
  // window at x0 sits at x0 + half - 1 = X - half.
  const int px0 = X - half, py0 = Y - half;
  int best = 0;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int py = py0 + j * half;
    if (static_cast<unsigned>(py) >= static_cast<unsigned>(pwy)) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int px = px0 + i * half;
      if (static_cast<unsigned>(px) >= static_cast<unsigned>(pwx)) continue;
      best = max(best, static_cast<int>(prev[px + py * pwx]));
    }
  }
  out[X + Y * wx] = static_cast<uint8_t>(best);
}

// planes[(py*w + px) * stride + J*PI + I] = level(I*w + px, J*w + py) (0 outside).
__global__ void BuildPlanesKernel(const uint8_t* __restrict__ level, int wx, int wy, int w, int PI,
                                  int PJ, int stride, uint8_t* __restrict__ planes) {
  const int plane = blockIdx.x;             // w*w planes + 1 zero plane
  const int px = plane % w, py = plane / w;
  for (int c = threadIdx.x; c < stride; c += blockDim.x) {
    int v = 0;
    if (plane < w * w && c < PI * PJ) {
      const int I = c % PI, J = c / PI;
      const int x = I * w + px, y = J * w + py;
      if (x < wx && y < wy) v = level[x + y * wx];
    }
    planes[static_cast<size_t>(plane) * stride + c] = static_cast<uint8_t>(v);
  }
}

// out(X, Y) = max of level(X - 2 + a, Y - 2 + b), a, b in [0, 4] (cells outside the level read 0), for
// X in [0, wx + 4), Y in [0, wy + 4): the level dilated by two cells either way, stored two cells
// up so that the border's dilation has a place (the group bounds of the fused front end).
constexpr int kGroupDilation = 2;
__global__ void DilateLevelKernel(const uint8_t* __restrict__ level, int wx, int wy,
                                  uint8_t* __restrict__ out) {
  const int X = blockIdx.x * blockDim.x + threadIdx.x;
  const int Y = blockIdx.y;
  const int ox = wx + 2 * kGroupDilation;
  if (X >= ox) return;
  int best = 0;
  for (int b = -kGroupDilation; b <= kGroupDilation; ++b) {
    const int y = Y - kGroupDilation + b;
    if (static_cast<unsigned>(y) >= static_cast<unsigned>(wy)) continue;
    for (int a = -kGroupDilation; a <= kGroupDilation; ++a) {
      const int x = X - kGroupDilation + a;
      if (static_cast<unsigned>(x) >= static_cast<unsigned>(wx)) continue;
      best = max(best, static_cast<int>(level[x + y * wx]));
    }
  }
  out[X + Y * ox] = static_cast<uint8_t>(best);
}

// quads(x + w, y + w) = level(x, y) | level(x, y+w) << 8 | level(x+w, y) << 16 |
// level(x+w, y+w) << 24 for x in [-w, wx), y in [-w, wy); cells outside the level read 0.
// Tiled storage: QuadOffset (scan_matching_2d.h).
__global__ void BuildQuadsKernel(const uint8_t* __restrict__ level, int wx, int wy, int w,
                                 uint32_t* __restrict__ quads, int qx, int qy, int qtx) {
  const int X = blockIdx.x * blockDim.x + threadIdx.x;
  const int Y = blockIdx.y;
  if (X >= qx) return;
  const int x = X - w, y = Y - w;
  auto at = [&](int cx, int cy) -> uint32_t {
    return (static_cast<unsigned>(cx) < static_cast<unsigned>(wx) &&
            static_cast<unsigned>(cy) < static_cast<unsigned>(wy))
               ? level[cx + cy * wx] : 0u;
  };
  quads[QuadOffset(X, Y, qtx)] =
      at(x, y) | (at(x, y + w) << 8) | (at(x + w, y) << 16) | (at(x + w, y + w) << 24);
}

// ---------------------------------------------------------------------------
// Scan preparation: rotate, translate, discretise, ShrinkToFit, bucket
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
PrepScansKernel(const Fast2DProblem* __restrict__ problems, const float* __restrict__ xyz, int n,
                ProblemState* __restrict__ states, int* __restrict__ counters_words,
                int num_counter_words) {
  // First kernel of a call: it also clears the list counters of the search (saves a
  // memset and its launch gap).
  // (every workgroup clears a slice: the counters with the work queue's control words are 376 KB)
  if (counters_words)
    for (int i = (blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x;
         i < num_counter_words; i += gridDim.x * gridDim.y * blockDim.x)
      counters_words[i] = 0;
  const Fast2DProblem& P = problems[blockIdx.y];
  const int s = blockIdx.x;
  if (s >= P.num_scans || P.use_fused) return;
  const Quat q0{P.init_qw, 0.f, 0.f, P.init_qz};
  const float2 r = P.scan_rot[s];
  const Quat qs{r.x, 0.f, 0.f, r.y};
  uint32_t* out = P.discrete + static_cast<size_t>(s) * n;
  int lo_x = 0, lo_y = 0, hi_x = 0, hi_y = 0, bad = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const F3 p{xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
    F3 a = Rotate(q0, p);                   // rotated_point_cloud (+ zero translation)
    a.x += 0.f; a.y += 0.f; a.z += 0.f;
    F3 b = Rotate(qs, a);                   // GenerateRotatedScans
    b.x += 0.f; b.y += 0.f;
    const float x = (1.f * b.x + 0.f * b.y) + P.tx;   // Affine2f(translation) * v
    const float y = (0.f * b.x + 1.f * b.y) + P.ty;
    // MapLimits::GetCellIndex (mapping/2d/map_limits.h:69-76).
    const int ix = CellIndexF64(P.max_y - static_cast<double>(y), P.res, P.inv_res);
    const int iy = CellIndexF64(P.max_x - static_cast<double>(x), P.res, P.inv_res);
    if (ix < -32768 || ix > 32767 || iy < -32768 || iy > 32767) bad = 1;
    out[i] = (static_cast<uint32_t>(ix) & 0xffffu) | (static_cast<uint32_t>(iy) << 16);
    lo_x = min(lo_x, -ix);
    lo_y = min(lo_y, -iy);
    hi_x = max(hi_x, P.nx - 1 - ix);
    hi_y = max(hi_y, P.ny - 1 - iy);
  }
  __shared__ int red[4][5];
  __shared__ int4 s_bounds;
  __shared__ int2 s_dims;
  lo_x = WaveMin(lo_x); lo_y = WaveMin(lo_y);
  hi_x = WaveMax(hi_x); hi_y = WaveMax(hi_y);
  bad = WaveMax(bad);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    red[wave][0] = lo_x; red[wave][1] = lo_y; red[wave][2] = hi_x; red[wave][3] = hi_y;
    red[wave][4] = bad;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) {
      lo_x = min(lo_x, red[w][0]); lo_y = min(lo_y, red[w][1]);
      hi_x = max(hi_x, red[w][2]); hi_y = max(hi_y, red[w][3]);
      bad = max(bad, red[w][4]);
    }
    // SearchParameters::ShrinkToFit (SM2/correlative_scan_matcher_2d.cc:73-91).
    int4 bd;
    bd.x = max(-P.nl, lo_x);
    bd.y = min(P.nl, hi_x);
    bd.z = max(-P.nl, lo_y);
    bd.w = min(P.nl, hi_y);
    P.bounds[s] = bd;
    // GenerateLowestResolutionCandidates counts (SM2/fast_...2d.cc:279-292).
    const int step = 1 << (P.depth - 1);
    const int2 dims = make_int2((bd.y - bd.x + step) / step, (bd.w - bd.z + step) / step);
    P.coarse_dims[s] = dims;
    s_bounds = bd;
    s_dims = dims;
    if (bad) atomicMax(&states[blockIdx.y].error, 1);
    const int count = dims.x * dims.y;
    if (count > P.coarse_stride || (P.use_planes && count > kMaxCoarsePerScan))
      atomicMax(&states[blockIdx.y].error, 2);
  }
  if (!P.use_planes) return;

  // ---- bucket the points by the lattice block they fall in ---------------
  __shared__ int hist[kMaxBuckets];
  __shared__ int partial[256];
  __syncthreads();
  const int4 bd = s_bounds;
  const int2 dims = s_dims;
  const int shift = P.depth - 1, w = 1 << shift;
  const int BW = dims.x + P.plane_i - 1, BH = dims.y + P.plane_j - 1;
  const int NB = BW * BH;   // <= kMaxBuckets (checked on the host with upper bounds)
  for (int b = threadIdx.x; b < NB; b += blockDim.x) hist[b] = 0;
  __syncthreads();
  auto classify = [&](uint32_t packed, int* bucket, int* plane, uint32_t* block = nullptr) {
    const int U = static_cast<short>(packed & 0xffffu) + bd.x + w - 1;
    const int V = static_cast<short>(packed >> 16) + bd.z + w - 1;
    const int bx = (U >> shift) + dims.x - 1, by = (V >> shift) + dims.y - 1;
    *plane = (V & (w - 1)) * w + (U & (w - 1));
    *bucket = (bx >= 0 && bx < BW && by >= 0 && by < BH) ? by * BW + bx : -1;
    if (block) *block = static_cast<uint32_t>(bx) | (static_cast<uint32_t>(by) << 8);
  };
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    int bucket, plane;
    classify(out[i], &bucket, &plane);
    if (bucket >= 0) atomicAdd(&hist[bucket], 1);
  }
  __syncthreads();
  // exclusive scan of hist[0..NB)
  const int chunk = (NB + 255) / 256;
  const int b0 = min(static_cast<int>(threadIdx.x) * chunk, NB), b1 = min(b0 + chunk, NB);
  int sum = 0;
  for (int b = b0; b < b1; ++b) sum += hist[b];
  partial[threadIdx.x] = sum;
  __syncthreads();
  if (threadIdx.x < 64) {   // exclusive scan of the 256 partials by one wave, 4 per lane
    const int l = threadIdx.x;
    const int a0 = partial[4 * l], a1 = partial[4 * l + 1], a2 = partial[4 * l + 2],
              a3 = partial[4 * l + 3];
    const int mine = a0 + a1 + a2 + a3;
    int incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int o = __shfl_up(incl, off, 64);
      if (l >= off) incl += o;
    }
    const int base = incl - mine;
    partial[4 * l] = base;
    partial[4 * l + 1] = base + a0;
    partial[4 * l + 2] = base + a0 + a1;
    partial[4 * l + 3] = base + a0 + a1 + a2;
    if (l == 63) P.sorted_count[s] = incl;
  }
  __syncthreads();
  int run = partial[threadIdx.x];
  for (int b = b0; b < b1; ++b) { const int v = hist[b]; hist[b] = run; run += v; }
  __syncthreads();
  // Records carry what the plane scorer would otherwise compute per point: the byte
  // offset of the point's plane and the constant bx * pitch + by its lattice block
  // subtracts in the accumulator index (pitch = dims.y + 2 * plane_j - 2).
  uint2* sorted = P.sorted + static_cast<size_t>(s) * n;
  const int pitch = dims.y + 2 * P.plane_j - 2;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    int bucket, plane;
    uint32_t block;
    classify(out[i], &bucket, &plane, &block);
    if (bucket >= 0) {
      const int pos = atomicAdd(&hist[bucket], 1);
      sorted[pos] = make_uint2(static_cast<uint32_t>(plane) * P.plane_stride,
                               (block & 0xffu) * pitch + (block >> 8));
    }
  }
}

// ---------------------------------------------------------------------------
// Scoring
// ---------------------------------------------------------------------------
__device__ __forceinline__ float ToScore(const Fast2DProblem& P, int sum, int n) {
  // ToScore(sum / float(N))  (SM2/fast_...2d.cc:330-331, .h:74-76)
  return P.min_s + (static_cast<float>(sum) / static_cast<float>(n)) * P.score_scale;
}

// An integer >= the sum a node's score was computed from (inverse of ToScore,
// rounded generously upwards; only used to prune).
__device__ __forceinline__ int SumUpperBound(const Fast2DProblem& P, float score, int n) {
  const float s = (score - P.min_s) / P.score_scale * static_cast<float>(n);
  const float ub = ceilf(s * (1.f + 1e-5f)) + 2.f;
  return static_cast<int>(fminf(fmaxf(ub, 0.f), 255.f * static_cast<float>(n)));
}

// Integer sum of one candidate over all points, one wave per candidate
// (SM2/fast_...2d.cc:320-329 with GetValue of .h:56-71).  Generic fallback of
// the lowest resolution when the phase-plane layout does not apply.
__device__ __forceinline__ int ScoreCandidateWave(const LevelDesc& L, int level,
                                                  const uint32_t* __restrict__ scan, int n, int dx,
                                                  int dy, int lane) {
  const int off = (1 << level) - 1;   // -offset_
  const int ax = dx + off, ay = dy + off;
  const auto* cells = AsGlobal(L.cells);
  const auto* gscan = AsGlobal(scan);
  int sum = 0;
#pragma unroll 4
  for (int i = lane; i < n; i += kWave) {
    const uint32_t p = gscan[i];
    const int x = static_cast<short>(p & 0xffffu) + ax;
    const int y = static_cast<short>(p >> 16) + ay;
    const bool ok = static_cast<unsigned>(x) < static_cast<unsigned>(L.wx) &&
                    static_cast<unsigned>(y) < static_cast<unsigned>(L.wy);
    const unsigned v = cells[ok ? x + y * L.wx : 0];   // unconditional load, masked value
    sum += ok ? v : 0u;
  }
  return WaveSum(sum);
}

// Block-wide (sum, local index) maximum, smallest index on ties; result valid
// in thread 0.
__device__ __forceinline__ int2 BlockBest(int sum, int index, int2* scratch /*[4]*/) {
  // sum >= -1 (idle threads pass -1); bias by one so the key is unsigned.
  unsigned long long key =
      (static_cast<unsigned long long>(static_cast<unsigned>(sum + 1)) << 32) |
      static_cast<unsigned>(0x7fffffff - index);
  key = WaveMaxU64(key);
  if ((threadIdx.x & 63) == 0)
    scratch[threadIdx.x >> 6] = make_int2(static_cast<int>(key >> 32) - 1,
                                          0x7fffffff - static_cast<int>(key & 0xffffffffu));
  __syncthreads();
  int2 best = scratch[0];
  if (threadIdx.x == 0) {
    for (int w = 1; w < static_cast<int>(blockDim.x >> 6); ++w) {
      const int2 o = scratch[w];
      if (o.x > best.x || (o.x == best.x && o.y < best.y)) best = o;
    }
  }
  return best;
}

__global__ void __launch_bounds__(256)
ScoreCoarseGenericKernel(const Fast2DProblem* __restrict__ problems, int n,
                         const ProblemState* __restrict__ states) {
  const Fast2DProblem& P = problems[blockIdx.y];
  const int s = blockIdx.x;
  if (s >= P.num_scans || states[blockIdx.y].error || P.use_planes) return;
  const int level = P.depth - 1;
  const int step = 1 << level;
  const int2 dims = P.coarse_dims[s];
  const int4 bd = P.bounds[s];
  const int base = s * P.coarse_stride;
  const uint32_t* scan = P.discrete + static_cast<size_t>(s) * n;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int count = dims.x * dims.y;
  int best_sum = -1, best_index = 0x7ffffff;  // idle threads (sum -1) never win
  for (int c = wave; c < count; c += 4) {
    const int ix = c / dims.y, iy = c - ix * dims.y;   // x outer, y inner (:295-307)
    const int sum = ScoreCandidateWave(P.level[level], level, scan, n, bd.x + ix * step,
                                       bd.z + iy * step, lane);
    if (lane == 0) {
      P.coarse_sum[base + c] = sum;
      P.coarse_score[base + c] = ToScore(P, sum, n);
    }
    if (sum > best_sum) { best_sum = sum; best_index = c; }
  }
  __shared__ int2 scratch[4];
  const int2 best = BlockBest(best_sum, best_index, scratch);
  if (threadIdx.x == 0) P.scan_best[s] = best;
}

// Phase-plane scoring of all lowest-resolution candidates of one scan.
template <int CHUNKS>
__global__ void __launch_bounds__(256)
ScoreCoarsePlanesKernel(const Fast2DProblem* __restrict__ problems, int n,
                        const ProblemState* __restrict__ states) {
  const Fast2DProblem& P = problems[blockIdx.y];
  const int s = blockIdx.x;
  if (s >= P.num_scans || states[blockIdx.y].error || !P.use_planes || P.use_fused) return;
  if ((P.plane_stride >> 6) != CHUNKS) return;
  // Candidate accumulators, padded by the plane extent on every side: a lane's cell
  // (I, J) in lattice block (bx, by) belongs to candidate
  //   (ix, iy) = (I - bx + dims.x - 1, J - by + dims.y - 1),
  // which may lie outside [0, dims); with the padding its accumulator index
  //   (ix + PI - 1) * pitch + (iy + PJ - 1) = lane_const - (bx * pitch + by)
  // is always inside the array, so a flush is one subtract and one LDS add per lane,
  // no bounds logic (out-of-range candidates collect in padding nobody reads).
  extern __shared__ int cand_acc[];
  __shared__ int2 scratch[4];
  const int2 dims = P.coarse_dims[s];
  const int count = dims.x * dims.y;
  const int PI = P.plane_i, PJ = P.plane_j, PIJ = PI * PJ;
  const int pitch = dims.y + 2 * PJ - 2;
  const int acc_cells = (dims.x + 2 * PI - 2) * pitch;
  for (int i = threadIdx.x; i < acc_cells; i += blockDim.x) cand_acc[i] = 0;
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int M = P.sorted_count[s];
  // (x = plane byte offset, y = block constant) as one 64-bit word per record
  const auto* rec = AsGlobal(reinterpret_cast<const unsigned long long*>(P.sorted)) +
                    static_cast<size_t>(s) * n;
  const int waves = blockDim.x >> 6;     // 2..4, chosen by the host (see the launch)
  const int begin = static_cast<int>(static_cast<long long>(M) * wave / waves);
  const int end = static_cast<int>(static_cast<long long>(M) * (wave + 1) / waves);
  const int stride = P.plane_stride;
  const unsigned zero_plane = 1u << (2 * (P.depth - 1));   // index w*w: the all-zero plane

  int acc[CHUNKS], lane_const[CHUNKS];
#pragma unroll
  for (int c = 0; c < CHUNKS; ++c) {
    acc[c] = 0;
    // Lanes past the plane read its zero padding: let them add 0 to the last cell.
    const int cell = min(c * 64 + lane, PIJ - 1);
    lane_const[c] = (cell % PI + dims.x + PI - 2) * pitch + (cell / PI + dims.y + PJ - 2);
  }
  int cur = -1;
  auto flush = [&](int block_const) {
#pragma unroll
    for (int c = 0; c < CHUNKS; ++c) {
      atomicAdd(&cand_acc[lane_const[c] - block_const], acc[c]);
      acc[c] = 0;
    }
  };

  // Records are wave-uniform: 64 of them arrive with one coalesced 8-byte load per lane
  // (the next 64 prefetched meanwhile) and are broadcast with v_readlane (immediate lane
  // index: the batch loops are fully unrolled).  A plane read is a buffer load: lane
  // offset in a VGPR, the record's plane offset in an SGPR, no address arithmetic at all.
  // kBatch plane loads are in flight before the first one is consumed.  Lanes past `end`
  // hold the sentinel (all-zero plane, block -1): adding zeros changes nothing.
  constexpr int kBatch = CHUNKS == 1 ? 32 : (CHUNKS == 2 ? 16 : 8);
  const int kSentinelBlock = -1;
  const unsigned long long sentinel =
      (static_cast<unsigned long long>(static_cast<uint32_t>(kSentinelBlock)) << 32) |
      static_cast<uint32_t>(zero_plane * stride);
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint8_t*>(P.planes), 0, static_cast<int>((zero_plane + 1) * stride), 0x00020000);
  unsigned long long mine = sentinel;
  if (begin + lane < end) mine = rec[begin + lane];
  for (int base_i = begin; base_i < end; base_i += 64) {
    unsigned long long next = sentinel;
#pragma unroll
    for (int j0 = 0; j0 < 64; j0 += kBatch) {
      if (base_i + j0 >= end) break;          // wave-uniform
      int block[kBatch];
      int v[kBatch][CHUNKS];
#pragma unroll
      for (int k = 0; k < kBatch; ++k) {
        const int plane_offset =
            __builtin_amdgcn_readlane(static_cast<int>(mine & 0xffffffffu), j0 + k);
        block[k] = __builtin_amdgcn_readlane(static_cast<int>(mine >> 32), j0 + k);
#pragma unroll
        for (int c = 0; c < CHUNKS; ++c)
          v[k][c] = __builtin_amdgcn_raw_buffer_load_b8(rsrc, lane + c * 64, plane_offset, 0);
      }
      if (j0 == 0) {   // prefetch the next 64 records behind this batch's plane loads
        const int nidx = base_i + 64 + lane;
        next = rec[min(nidx, end - 1)];
        if (nidx >= end) next = sentinel;
      }
#pragma unroll
      for (int k = 0; k < kBatch; ++k) {
        if (block[k] != cur) {
          if (cur >= 0) flush(cur);
          cur = block[k];     // the sentinel block (-1) only ever follows real ones
        }
#pragma unroll
        for (int c = 0; c < CHUNKS; ++c) acc[c] += v[k][c];
      }
    }
    mine = next;
  }
  if (cur >= 0) flush(cur);
  __syncthreads();

  const int base = s * P.coarse_stride;
  auto* coarse_sum = AsGlobal(P.coarse_sum) + base;
  auto* coarse_score = AsGlobal(P.coarse_score) + base;
  int best_sum = -1, best_index = 0x7ffffff;  // idle threads (sum -1) never win
  for (int i = threadIdx.x; i < count; i += blockDim.x) {
    const int ix = i / dims.y, iy = i - ix * dims.y;
    const int sum = cand_acc[(ix + PI - 1) * pitch + (iy + PJ - 1)];
    coarse_sum[i] = sum;
    coarse_score[i] = ToScore(P, sum, n);
    if (sum > best_sum) { best_sum = sum; best_index = i; }
  }
  const int2 best = BlockBest(best_sum, best_index, scratch);
  if (threadIdx.x == 0) P.scan_best[s] = best;
}

// The same scoring for 64-byte planes (plane_i * plane_j <= 64, the usual case) with DWORD
// gathers.  A wave-wide `buffer_load_ubyte` costs the texture-address path ~12 cycles
// however few cache lines it touches (2.3 M of them were the 45 us of the byte variant:
// SQ/TA counters in profiles/HISTORY.md); here a lane fetches four plane cells at once, sixteen lanes
// cover a plane, and one instruction serves FOUR records (lane group g = lane / 16 takes
// records 4t + g).  Groups sit in different lattice blocks, so the block bookkeeping is
// per lane: packed 16-bit partial sums (cells 0|2 and 1|3), flushed to the LDS
// accumulators when the lane's block changes or after 256 records.
__global__ void __launch_bounds__(256)
ScoreCoarsePlanesDwordKernel(const Fast2DProblem* __restrict__ problems, int n,
                             const ProblemState* __restrict__ states) {
  const Fast2DProblem& P = problems[blockIdx.y];
  const int s = blockIdx.x;
  if (s >= P.num_scans || states[blockIdx.y].error || !P.use_planes || P.use_fused) return;
  if (P.plane_stride != 64) return;
  extern __shared__ int cand_acc[];
  __shared__ int2 scratch[4];
  const int2 dims = P.coarse_dims[s];
  const int count = dims.x * dims.y;
  const int PI = P.plane_i, PJ = P.plane_j, PIJ = PI * PJ;
  const int pitch = dims.y + 2 * PJ - 2;
  const int acc_cells = (dims.x + 2 * PI - 2) * pitch;
  for (int i = threadIdx.x; i < acc_cells; i += blockDim.x) cand_acc[i] = 0;
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int group = lane >> 4, sub = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int M = P.sorted_count[s];
  const auto* rec = AsGlobal(reinterpret_cast<const unsigned long long*>(P.sorted)) +
                    static_cast<size_t>(s) * n;
  const int waves = blockDim.x >> 6;
  const int begin = static_cast<int>(static_cast<long long>(M) * wave / waves);
  const int end = static_cast<int>(static_cast<long long>(M) * (wave + 1) / waves);
  const unsigned zero_plane = 1u << (2 * (P.depth - 1));   // index w*w: the all-zero plane

  int lane_const[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    // Cells past the plane are zero padding: they add 0 to the last cell.
    const int cell = min(4 * sub + j, PIJ - 1);
    lane_const[j] = (cell % PI + dims.x + PI - 2) * pitch + (cell / PI + dims.y + PJ - 2);
  }
  int cur = -1, pending = 0;
  uint32_t even = 0, odd = 0;               // cells 0 | 2 << 16 and 1 | 3 << 16
  const auto flush = [&]() {
    const int a0 = even & 0xffffu, a2 = even >> 16, a1 = odd & 0xffffu, a3 = odd >> 16;
    if (a0) atomicAdd(&cand_acc[lane_const[0] - cur], a0);
    if (a1) atomicAdd(&cand_acc[lane_const[1] - cur], a1);
    if (a2) atomicAdd(&cand_acc[lane_const[2] - cur], a2);
    if (a3) atomicAdd(&cand_acc[lane_const[3] - cur], a3);
    even = odd = 0;
    pending = 0;
  };

  constexpr int kSteps = 8;                 // gathers (of four records each) in flight
  const unsigned long long sentinel = (0xffffffffull << 32) | (zero_plane * 64u);
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint8_t*>(P.planes), 0, static_cast<int>((zero_plane + 1) * 64), 0x00020000);
  unsigned long long mine = sentinel;
  if (begin + lane < end) mine = rec[begin + lane];
  for (int base_i = begin; base_i < end; base_i += 64) {
    unsigned long long next = sentinel;
#pragma unroll
    for (int t0 = 0; t0 < 16; t0 += kSteps) {
      if (base_i + 4 * t0 >= end) break;      // wave-uniform
      int block[kSteps];
      uint32_t q[kSteps];
#pragma unroll
      for (int k = 0; k < kSteps; ++k) {
        const int src = 4 * (t0 + k) + group;                 // this lane group's record
        const int plane_offset = __shfl(static_cast<int>(mine & 0xffffffffu), src, 64);
        block[k] = __shfl(static_cast<int>(mine >> 32), src, 64);
        q[k] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, plane_offset + 4 * sub, 0, 0);
      }
      if (t0 == 0) {   // prefetch the next 64 records behind the first gathers
        const int nidx = base_i + 64 + lane;
        next = rec[min(nidx, end - 1)];
        if (nidx >= end) next = sentinel;
      }
#pragma unroll
      for (int k = 0; k < kSteps; ++k) {
        if (block[k] != cur) {                // per lane group
          if (cur >= 0) flush();
          cur = block[k];                     // -1 (sentinel) only ever follows real blocks
        }
        even += q[k] & 0x00ff00ffu;
        odd += (q[k] >> 8) & 0x00ff00ffu;
        if (++pending == 256) flush();        // 16-bit partial sums: 256 x 255 fits
      }
    }
    mine = next;
  }
  if (cur >= 0) flush();
  __syncthreads();

  const int base = s * P.coarse_stride;
  auto* coarse_sum = AsGlobal(P.coarse_sum) + base;
  auto* coarse_score = AsGlobal(P.coarse_score) + base;
  int best_sum = -1, best_index = 0x7ffffff;  // idle threads (sum -1) never win
  for (int i = threadIdx.x; i < count; i += blockDim.x) {
    const int ix = i / dims.y, iy = i - ix * dims.y;
    const int sum = cand_acc[(ix + PI - 1) * pitch + (iy + PJ - 1)];
    coarse_sum[i] = sum;
    coarse_score[i] = ToScore(P, sum, n);
    if (sum > best_sum) { best_sum = sum; best_index = i; }
  }
  const int2 best = BlockBest(best_sum, best_index, scratch);
  if (threadIdx.x == 0) P.scan_best[s] = best;
}

// ---------------------------------------------------------------------------
// Fused front end (the usual case: 64-byte phase planes, the scan fits in LDS)
// ---------------------------------------------------------------------------
// One block per rotated scan does everything the reference does for that scan before
// branch and bound -- GenerateRotatedScans + DiscretizeScans + ShrinkToFit
// (SM2/correlative_scan_matcher_2d.cc:73-127), GenerateLowestResolutionCandidates and
// their ScoreCandidates (SM2/fast_correlative_scan_matcher_2d.cc:264-333) -- with the
// discretised scan staged in LDS only.  As separate launches the same work wrote 27 MB per
// match (discrete scans + 64-bit bucketed records) and the scorer fetched 20 MB of it back;
// here only the candidates' scores (1.7 MB) leave the chip -- the tree search re-derives the
// cells of the scans it descends into (ScanCell) -- and the per-scan candidate layout needs
// no prefix sum: scan s owns [s * coarse_stride, (s + 1) * coarse_stride).
//
// Unlike the separate launches this kernel does NOT sort the points by lattice block.  The
// sort (histogram, scan, scatter: seven barriers) was a third of a block's latency, and it
// buys little: a range scan is spatially coherent -- consecutive returns fall into the same
// 2^(depth-1)-cell lattice block for dozens of points -- so scoring in point order flushes
// the register accumulators only when a lane group's block really changes.  (An unordered
// cloud stays correct: it flushes more often.)  Every lane classifies its own point of a
// 64-point chunk (plane, lattice block); lane group g = lane / 16 takes points 4 t + g, so
// one buffer_load_dword still serves four points, sixteen of them in flight per wave.
// (Wider gathers do not help: the plane reads run at ~8 B/clk per CU whatever the
// instruction width -- buffer_load_dwordx4, sixteen points per instruction, was slower --
// because every point touches its own 64-byte half of a 128-byte L2 line.)
// The integer sums are order-free: results are bit-identical to the sorted variant
// (ScoreCoarsePlanesDwordKernel, kept for CMX_FUSED=0 and for problems this kernel does not
// take).
// Dynamic LDS: pts[group][n_pad] u32 | misc[kFusedMisc] | cand_acc[acc_cap] | point words[waves][64].
constexpr int kFusedMaxPoints = 4096;    // = kPointCache of the tree search
constexpr int kFusedGroup = 3;           // rotations per workgroup under group bounds (see the kernel)
constexpr int kFusedMisc = 128;          // ints of bookkeeping between the cells and the accumulators

// Points kFirst .. kFirst + 7 of a lane group (LDS words at a stride of 16 bytes from `base`): the
// low halves into lo[0..7], the high halves into hi[0..7], zero-extended; returns when they landed.
template <int kFirst>
__device__ __forceinline__ void ReadHalves8(unsigned base, uint32_t* lo, uint32_t* hi) {
  constexpr int o = 16 * kFirst;
  asm volatile(
      "ds_read_u16 %0, %16 offset:%17\n\tds_read_u16 %8, %16 offset:%18\n\t"
      "ds_read_u16 %1, %16 offset:%19\n\tds_read_u16 %9, %16 offset:%20\n\t"
      "ds_read_u16 %2, %16 offset:%21\n\tds_read_u16 %10, %16 offset:%22\n\t"
      "ds_read_u16 %3, %16 offset:%23\n\tds_read_u16 %11, %16 offset:%24\n\t"
      "ds_read_u16 %4, %16 offset:%25\n\tds_read_u16 %12, %16 offset:%26\n\t"
      "ds_read_u16 %5, %16 offset:%27\n\tds_read_u16 %13, %16 offset:%28\n\t"
      "ds_read_u16 %6, %16 offset:%29\n\tds_read_u16 %14, %16 offset:%30\n\t"
      "ds_read_u16 %7, %16 offset:%31\n\tds_read_u16 %15, %16 offset:%32\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(lo[0]), "=&v"(lo[1]), "=&v"(lo[2]), "=&v"(lo[3]), "=&v"(lo[4]), "=&v"(lo[5]),
        "=&v"(lo[6]), "=&v"(lo[7]), "=&v"(hi[0]), "=&v"(hi[1]), "=&v"(hi[2]), "=&v"(hi[3]),
        "=&v"(hi[4]), "=&v"(hi[5]), "=&v"(hi[6]), "=&v"(hi[7])
      : "v"(base), "n"(o), "n"(o + 2), "n"(o + 16), "n"(o + 18), "n"(o + 32), "n"(o + 34),
        "n"(o + 48), "n"(o + 50), "n"(o + 64), "n"(o + 66), "n"(o + 80), "n"(o + 82),
        "n"(o + 96), "n"(o + 98), "n"(o + 112), "n"(o + 114)
      : "memory");
}

// The sums of a unit of PrepScoreFusedKernel (below, where the terms are explained): the cells of ONE
// rotation of the unit summed over the phase planes -- the dilated level's under group bounds, the
// level's own otherwise -- and the sums handed to the rotations they stand for.  Everything it needs
// comes out of the block's bookkeeping words in LDS, so that nothing but those is live across the
// gather loop (which fills the 64 VGPRs of eight wavefronts per SIMD on its own: no scratch).
template <bool kTimeline>
__device__ __forceinline__ void FusedPass(const Fast2DProblem& P, ProblemState* state, int n,
                                          int acc_cap, int s0, int timeline_block) {
  extern __shared__ __attribute__((aligned(16))) unsigned char fused_smem[];
  const auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
  const int n_pad = (n + 63) & ~63;
  const int G = P.group > 1 ? kFusedGroup : 1;
  auto* pts_all = reinterpret_cast<uint32_t*>(fused_smem);        // [G][n_pad]
  int* misc = reinterpret_cast<int*>(pts_all + G * n_pad);
  int* cand_acc = misc + kFusedMisc;
  const int T = blockDim.x;
  const int waves = T >> 6;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const auto stamp = [&](int k) {
    if constexpr (kTimeline) Stamp(P.timeline, timeline_block, k);
  };
  const int gcount = uni(misc[1]), gm = uni(misc[2]);
  const bool far = uni(misc[7]) != 0;       // the premise of the group bound failed for this unit
  const int2 dims_all = make_int2(uni(misc[4]), uni(misc[5]));
  const int PI = P.plane_i, PJ = P.plane_j, PIJ = PI * PJ;
  const int shift = P.depth - 1, w = 1 << shift;
  const unsigned zero_plane = 1u << (2 * (P.depth - 1));
  const int group = lane >> 4, sub = lane & 15;
  const int begin = static_cast<int>(static_cast<long long>(n) * wave / waves);
  const int end = static_cast<int>(static_cast<long long>(n) * (wave + 1) / waves);
  constexpr int kSteps = 16;                // all gathers of a 64-point chunk in flight
  uint32_t* const wave_words = reinterpret_cast<uint32_t*>(cand_acc + acc_cap) + 64 * wave;
  const unsigned group_base = static_cast<unsigned>(reinterpret_cast<uintptr_t>(
      (__attribute__((address_space(3))) uint32_t*)(wave_words + group)));
  int2* const scratch = reinterpret_cast<int2*>(misc + 8);      // [4]
  const bool verify = (P.group_verify & 1) != 0;
    const bool group_pass = G > 1;
    const int gp = group_pass ? gm : 0;                  // whose cells are summed
    const uint32_t* const pts = pts_all + gp * n_pad;
    const int* const mine = misc + 16 + 8 * gp;
    const int4 bd = make_int4(uni(mine[0]), uni(mine[1]), uni(mine[2]), uni(mine[3]));
    const int2 dims = group_pass ? dims_all : make_int2(uni(mine[4]), uni(mine[5]));
    const int lift = group_pass ? kGroupDilation : 0;    // (the dilated level is stored two cells up)
    const int pitch = dims.y + 2 * PJ - 2;
    const int BW = dims.x + PI - 1, BH = dims.y + PJ - 1;

  // ---- score in point order (cf. ScoreCoarsePlanesDwordKernel) ------------------------
  int lane_const[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int cell = min(4 * sub + j, PIJ - 1);
    lane_const[j] = (cell % PI + dims.x + PI - 2) * pitch + (cell / PI + dims.y + PJ - 2);
  }
  // Four 32-bit running sums, one per plane cell of the lane's dword (round 4's per-chunk stamps:
  // a step of this loop is ~15 issued instructions on a SIMD shared by 4.5 wavefronts -- 1.9 of a
  // chunk's 2.1 us, the gathers themselves land in 0.16 -- so the packed 16-bit pairs, whose
  // overflow guard cost a counter, a compare and a branch per step, are gone: byte k of the dword
  // is added with one (SDWA) instruction each).
  uint32_t cur = 0;                         // lattice block + 1 of the running sums; 0: none yet
  uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  const auto flush = [&]() {
    const int at = static_cast<int>(cur) - 1;
    if (a0) atomicAdd(&cand_acc[lane_const[0] - at], static_cast<int>(a0));
    if (a1) atomicAdd(&cand_acc[lane_const[1] - at], static_cast<int>(a1));
    if (a2) atomicAdd(&cand_acc[lane_const[2] - at], static_cast<int>(a2));
    if (a3) atomicAdd(&cand_acc[lane_const[3] - at], static_cast<int>(a3));
    a0 = a1 = a2 = a3 = 0;
  };
  // (the selected pointer made uniform by hand: the compiler turns the selection into ONE vector
  // load from a selected address, and a resource out of vector registers costs a waterfall loop
  // around every gather)
  const unsigned long long planes_bits =
      reinterpret_cast<unsigned long long>(group_pass ? P.planes_group : P.planes);
  // (readfirstlane returns an int: through `unsigned`, or the low half sign-extends over the high one)
  const unsigned planes_lo = static_cast<unsigned>(
      __builtin_amdgcn_readfirstlane(static_cast<int>(static_cast<unsigned>(planes_bits))));
  const unsigned planes_hi = static_cast<unsigned>(
      __builtin_amdgcn_readfirstlane(static_cast<int>(static_cast<unsigned>(planes_bits >> 32))));
  const uint8_t* const planes_uniform = reinterpret_cast<const uint8_t*>(
      (static_cast<unsigned long long>(planes_hi) << 32) | planes_lo);
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint8_t*>(planes_uniform), 0, static_cast<int>((zero_plane + 1) * 64), 0x00020000);
  for (int base_i = begin; base_i < end; base_i += 64) {
    // (instrumented instantiation only -- wavefront 0's first chunk step by step: [8] chunk
    // begins, [9] its sixteen gathers issued, [10] all of them landed, [11] consumed; [12..15]:
    // the next four chunks begin.  profiles/HISTORY.md 5.1: which part of a chunk takes its 2.25 us)
    const int chunk_index = (base_i - begin) >> 6;
    if constexpr (kTimeline) {
      if (chunk_index == 0) stamp(8);
      else if (chunk_index <= 4) stamp(11 + chunk_index);
    }
    // This lane's point of the chunk: byte offset of its phase plane and the constant
    // bx * pitch + by of its lattice block (-1: no candidate of this scan can reach it).
    // ONE word per point: plane index (low half; the zero plane for a point no candidate
    // reaches) and lattice block + 1 (high half; BW, BH <= 255 and the host keeps the pitch so
    // that it fits).  The wavefront parks its 64 words in LDS and a lane group reads point
    // 4 k + group of the chunk at the IMMEDIATE offset 16 k from its own base, as two 16-bit
    // halves: the plane index needs one shift-or to become the gather's offset and the block goes
    // straight into the compare.  (Before: two ds_bpermute and an address add per step; as ONE
    // packed word a shift, a mask and a decrement more -- on the unit the loop is bound by.)
    uint32_t my_word = zero_plane;
    if (base_i + lane < end) {
      const uint32_t packed = pts[base_i + lane];
      const int U = static_cast<short>(packed & 0xffffu) + bd.x + w - 1 + lift;
      const int V = static_cast<short>(packed >> 16) + bd.z + w - 1 + lift;
      const int bx = (U >> shift) + dims.x - 1, by = (V >> shift) + dims.y - 1;
      if (bx >= 0 && bx < BW && by >= 0 && by < BH)
        my_word = static_cast<uint32_t>((V & (w - 1)) * w + (U & (w - 1))) |
                  (static_cast<uint32_t>(bx * pitch + by + 1) << 16);
    }
    wave_words[lane] = my_word;
    __builtin_amdgcn_wave_barrier();
    uint32_t block[kSteps];                            // lattice block + 1; 0: a skipped point
    uint32_t plane[kSteps];
    uint32_t q[kSteps];
    // (inline assembly: written as 16-bit loads in C++, the compiler merges the two halves of a
    // word into one ds_read_b32 and takes them apart again with a mask and a shift per step)
    ReadHalves8<0>(group_base, plane, block);
    ReadHalves8<8>(group_base, plane + 8, block + 8);
#pragma unroll
    for (int k = 0; k < kSteps; ++k)
      q[k] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, (plane[k] << 6) | (4 * sub), 0, 0);
    __builtin_amdgcn_wave_barrier();                   // (the next chunk overwrites the words)
    if constexpr (kTimeline) {
      if (chunk_index == 0) {
        stamp(9);
        __builtin_amdgcn_s_waitcnt(0x0f70);      // vmcnt(0): the gathers' latency on its own
        stamp(10);
      }
    }
#pragma unroll
    for (int k = 0; k < kSteps; ++k) {
      if (block[k] != cur) {                // per lane group; 0 = skipped point (adds zeros)
        // (cur == 0: nothing has been added but bytes of the zero plane, every sum is 0 and
        // flush() issues no addition -- no second test per step)
        flush();
        cur = block[k];
      }
      a0 += q[k] & 0xffu;
      a1 += (q[k] >> 8) & 0xffu;
      a2 += (q[k] >> 16) & 0xffu;
      a3 += q[k] >> 24;
    }
    if constexpr (kTimeline) {
      if (chunk_index == 0) stamp(11);
    }
  }
  if (cur != 0) flush();
  stamp(4);      // wave 0 done gathering
  __syncthreads();
  stamp(5);      // all waves done

    // ---- the sums of this pass to the rotations they stand for ------------------------------
    // (a unit whose premise failed -- a point's cell, or a bound, further than one from the middle
    // rotation's: not seen so far, the angular step excludes it up to rounding -- keeps the middle
    // rotation's bound, which holds whatever the others do, and gives every candidate of the other
    // rotations the largest sum there is: nothing of them is excluded up here)
    for (int t = 0; t < gcount; ++t) {
      const int s = s0 + t;
      const int2 tdims = make_int2(misc[16 + 8 * t + 4], misc[16 + 8 * t + 5]);
      const int count = tdims.x * tdims.y;
      const int base = s * P.coarse_stride;
      auto* coarse_sum = AsGlobal(P.coarse_sum) + base;
      auto* coarse_score = AsGlobal(P.coarse_score) + base;
      const bool unbounded = far && t != gm;
      int best_sum = -1, best_index = 0x7ffffff;
      for (int i = threadIdx.x; i < count; i += T) {
        const int ix = i / tdims.y, iy = i - ix * tdims.y;
        const int csum = unbounded ? 255 * n : cand_acc[(ix + PI - 1) * pitch + (iy + PJ - 1)];
        if (group_pass) {
          // fast2d_group_verify: the exact sums of an earlier launch (group = 1) are in place
          if (verify && coarse_sum[i] > csum) atomicMax(&state->error, 3);
        } else if (P.write_all_discrete || verify) {
          coarse_sum[i] = csum;     // introspection only
        }
        coarse_score[i] = ToScore(P, csum, n);
        if (csum > best_sum) { best_sum = csum; best_index = i; }
      }
      const int2 best = BlockBest(best_sum, best_index, scratch);
      if (threadIdx.x == 0) P.scan_best[s] = best;
      stamp(6);      // scores written
      // The discretised scan stays on chip: the tree search re-derives the cells of the few scans
      // it descends into (ScanCell).  Only the introspection entry point asks for the array.
      // Batches (store_scans): a scan whose best candidate reaches the initial bound may enter the
      // tree search, where several nodes per scan are expanded by independent wavefronts; its
      // cells are written for them (a superset of what the coarse filter keeps: the bound only
      // rises).  Re-deriving the cells per node made that expansion VALU-bound.
      bool keep_cells = P.write_all_discrete != 0;
      if (!keep_cells && P.store_scans) {
        int top_sum = scratch[0].x;
        for (int k = 1; k < T >> 6; ++k) top_sum = max(top_sum, scratch[k].x);
        keep_cells = !(ToScore(P, top_sum, n) < fmaxf(P.min_score, 0.f));
      }
      if (keep_cells) {
        auto* out = AsGlobal(P.discrete) + static_cast<size_t>(s) * n;
        const uint32_t* const cells = pts_all + t * n_pad;
        for (int i = threadIdx.x; i < n; i += T) out[i] = cells[i];
      }
      __syncthreads();                       // (the next rotation's BlockBest reuses the scratch)
    }
}

template <bool kTimeline>    // (true: the debug switch `timeline`; the shipped instantiation has no stamps)
__global__ void __launch_bounds__(256, 8)   // (eight wavefronts per SIMD: at most 64 VGPRs)
PrepScoreFusedKernel(const Fast2DProblem* __restrict__ problems, const float* __restrict__ xyz,
                     int n, ProblemState* __restrict__ states, int acc_cap,
                     int* __restrict__ counters_words, int num_counter_words) {
  extern __shared__ __attribute__((aligned(16))) unsigned char fused_smem[];
  // First kernel of a fully fused batch: it also clears the list counters of the search.
  // (every workgroup clears a slice: the counters with the work queue's control words are 376 KB)
  if (counters_words)
    for (int i = (blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x;
         i < num_counter_words; i += gridDim.x * gridDim.y * blockDim.x)
      counters_words[i] = 0;
  const Fast2DProblem& P = problems[blockIdx.y];
  // GROUP BOUNDS (round 6).  Neighbouring rotations move a point by at most one cell (the angular
  // step is chosen so, SM2/correlative_scan_matcher_2d.cc:31-44), and their search bounds -- the
  // minimum over the points -- by at most one with it.  So for the G = 3 rotations g of a unit and
  // the middle one m, the cell a lowest-resolution candidate (kx, ky) of rotation g reads for point
  // p lies within two cells (per axis) of the cell candidate (kx, ky) of rotation m reads for p,
  // and ONE sum of m's cells over the level DILATED by two cells bounds the score of (kx, ky) of
  // all three from above.  Everything behind the front end (dive, filter, tree search) takes a
  // lowest-resolution score as the upper bound of the subtree below it and nothing else, so the
  // bound takes the score's place: a third of the gathers.  What needs the exact scores -- the
  // replay of the reference's std::sort when leaves tie (ResolveTies), depth 1, the introspection
  // entry point -- runs this kernel (again) with group = 1.  The premise is CHECKED per unit (every
  // point's cells, every bound): in a unit that fails it the outer rotations get the largest sum
  // there is, i.e. no bound (FusedPass).  fast2d_group_verify: every bound against the exact sums
  // of a launch with group = 1, on the device.
  const int G = P.group > 1 ? kFusedGroup : 1;
  // Units u, u + 256, u + 512, ... tend to share a CU (u % 8 picks the XCD, round-robin
  // within it): give them ADJACENT rotations.  Neighbouring rotations move a point by less
  // than a cell, so co-resident blocks gather the same or the neighbouring phase plane at
  // about the same time and meet in the CU's L1 instead of each going to L2.  (Any bijection
  // is correct; only speed depends on the dispatch order.)
  const int slots = (gridDim.x + 255) >> 8;
  const int unit = (blockIdx.x & 255) * slots + (blockIdx.x >> 8);
  const int s0 = unit * G;
  if (!P.use_fused || s0 >= P.num_scans) return;
  const int gcount = min(G, P.num_scans - s0);
  const int gm = gcount == 3 ? 1 : 0;       // the rotation of the unit whose cells are summed
  const int n_pad = (n + 63) & ~63;
  auto* pts_all = reinterpret_cast<uint32_t*>(fused_smem);        // [G][n_pad]
  int* misc = reinterpret_cast<int*>(pts_all + G * n_pad);
  int* cand_acc = misc + kFusedMisc;
  // misc: [0, 8) the pass (bounds, dims, ok, grouped) | [8, 16) BlockBest | [16 + 8 g, ...) bounds
  // and dims of rotation g | [40 + 20 g + 5 wave, ...) partials | [100 + wave] cell deltas
  const int T = blockDim.x;              // 128, 192 or 256
  const int waves = T >> 6;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const auto stamp = [&](int k) {
    if constexpr (kTimeline) Stamp(P.timeline, blockIdx.y * gridDim.x + blockIdx.x, k);
  };
  stamp(0);

  // ---- rotate, translate, discretise (PrepScansKernel's arithmetic) ----------
  const bool identity_q0 = P.init_qw == 1.f && P.init_qz == 0.f;
  // The cell of a point from an f32 ESTIMATE of the value the reference rounds,
  //     t = (max - translation) / res - 0.5 - (rotated coordinate) / res,
  // in two FMAs per coordinate (the real-time matcher's discretisation, rt_2d_tiles.hip, where
  // the error bound is derived: the estimate differs from GetCellIndex over RotateZ's f32 chain
  // by less than 2^-24 [((k_z + 4) (|ax| + |ay|) + |translation|) / res + 3 |K|], k_z =
  // max(2 + 4 z^2, 1 + 6 |z|) for this scan's rotation (w, z)); when it lies further than
  // 1.25 x that from every half-integer its rounding IS the reference's cell.  Otherwise -- three
  // points in a thousand at 60 m (20 M random points over the full circle: 0 wrong cells among
  // the decided ones) -- the exact expressions below run for that lane.  ~30 instead of ~110
  // vector instructions per point (a third of this kernel's instructions) -- and no measurable
  // change of its duration (same-box A/B: 132.4 -> 130.4 - 132.5 us per search): the kernel is
  // not bound by instruction issue but by the plane gathers below.
  const double inv_res_d = P.inv_res;
  const double Kyd = (P.max_y - static_cast<double>(P.ty)) * inv_res_d - 0.5;
  const double Kxd = (P.max_x - static_cast<double>(P.tx)) * inv_res_d - 0.5;
  const float Ky = static_cast<float>(Kyd), Kx = static_cast<float>(Kxd);
  const float bound_fixed = static_cast<float>(
      1.25 * 0x1p-24 * (inv_res_d * fmax(fabs(static_cast<double>(P.tx)), fabs(static_cast<double>(P.ty))) +
                        3.0 * fmax(fabs(Kxd), fabs(Kyd)) + 1.0));
  for (int g = 0; g < gcount; ++g) {
    const float2 r = P.scan_rot[s0 + g];
    const double zd = r.y;
    const float Ci = static_cast<float>((1.0 - 2.0 * zd * zd) * inv_res_d);
    const float Si = static_cast<float>(2.0 * static_cast<double>(r.x) * zd * inv_res_d);
    const float bound_per_m = static_cast<float>(
        1.25 * 0x1p-24 * inv_res_d * (4.0 + fmax(2.0 + 4.0 * zd * zd, 1.0 + 6.0 * fabs(zd))));
    uint32_t* const pts = pts_all + g * n_pad;
    int lo_x = 0, lo_y = 0, hi_x = 0, hi_y = 0, bad = 0;
    for (int i0 = threadIdx.x; i0 < n; i0 += 4 * T) {
      // Four points' loads in flight before the first is used.
      F3 p[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int j = min(i0 + k * T, n - 1);
        p[k] = F3{xyz[3 * j], xyz[3 * j + 1], xyz[3 * j + 2]};
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = i0 + k * T;
        if (i >= n) break;
        // Two yaw rotations (initial estimate, then this scan's perturbation), then the
        // translation: Rotate / `+ 0.f` / `1.f * x + 0.f * y` of PrepScansKernel without the
        // terms that are exactly zero (RotateZ, cmx_device.h).  A full-submap search starts
        // from yaw 0: its first rotation is the identity.
        float ax = p[k].x, ay = p[k].y;
        if (!identity_q0) RotateZ(P.init_qw, P.init_qz, p[k].x, p[k].y, &ax, &ay);
        const float tY = fmaf(-Ci, ay, fmaf(-Si, ax, Ky));    // cell x index from the map's y
        const float tX = fmaf(-Ci, ax, fmaf(Si, ay, Kx));
        const float nY = rintf(tY), nX = rintf(tX);
        const float margin = fminf(0.5f - fabsf(tY - nY), 0.5f - fabsf(tX - nX));
        const float bound = fmaf(fabsf(ax) + fabsf(ay), bound_per_m, bound_fixed);
        int ix, iy;
        if (margin > bound && fabsf(tY) < 1e6f && fabsf(tX) < 1e6f) {     // (NaN: not greater)
          ix = static_cast<int>(nY);
          iy = static_cast<int>(nX);
        } else {
          float bx, by;
          RotateZ(r.x, r.y, ax, ay, &bx, &by);
          const float x = bx + P.tx;
          const float y = by + P.ty;
          // lround((max - v) / res - 0.5), exact (cmx_device.h)
          ix = CellIndexFast(P.max_y, y, P.res, P.inv_res);
          iy = CellIndexFast(P.max_x, x, P.res, P.inv_res);
        }
        if (ix < -32768 || ix > 32767 || iy < -32768 || iy > 32767) bad = 1;
        pts[i] = (static_cast<uint32_t>(ix) & 0xffffu) | (static_cast<uint32_t>(iy) << 16);
        lo_x = min(lo_x, -ix);
        lo_y = min(lo_y, -iy);
        hi_x = max(hi_x, P.nx - 1 - ix);
        hi_y = max(hi_y, P.ny - 1 - iy);
      }
    }
    lo_x = WaveMinDpp(lo_x); lo_y = WaveMinDpp(lo_y);
    hi_x = WaveMaxDpp(hi_x); hi_y = WaveMaxDpp(hi_y);
    bad = WaveMaxDpp(bad);
    if (lane == 0) {
      int* red = misc + 40 + 20 * g + wave * 5;      // [G][4][5]
      red[0] = lo_x; red[1] = lo_y; red[2] = hi_x; red[3] = hi_y; red[4] = bad;
    }
  }
  for (int i = threadIdx.x; i < acc_cap; i += T) cand_acc[i] = 0;
  stamp(1);      // points discretised
  __syncthreads();
  if (gcount > 1) {            // (uniform)
    // the premise of the group bound: no point's cell further than one from the middle rotation's
    int far = 0;
    const uint32_t* const mid = pts_all + gm * n_pad;
    for (int g = 0; g < gcount; ++g) {
      if (g == gm) continue;
      const uint32_t* const other = pts_all + g * n_pad;
      for (int i = threadIdx.x; i < n; i += T) {
        const uint32_t a = mid[i], b = other[i];
        const int dx = static_cast<short>(a & 0xffffu) - static_cast<short>(b & 0xffffu);
        const int dy = static_cast<short>(a >> 16) - static_cast<short>(b >> 16);
        far |= (dx < -1 || dx > 1 || dy < -1 || dy > 1) ? 1 : 0;
      }
    }
    far = WaveMaxDpp(far);
    if (lane == 0) misc[100 + wave] = far;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const int step = 1 << (P.depth - 1);
    int bad = 0, far = 0;
    int2 dims_all = make_int2(0, 0);
    for (int g = 0; g < gcount; ++g) {
      const int* red = misc + 40 + 20 * g;
      int lo_x = red[0], lo_y = red[1], hi_x = red[2], hi_y = red[3];
      bad = max(bad, red[4]);
      for (int w = 1; w < waves; ++w) {
        red += 5;
        lo_x = min(lo_x, red[0]); lo_y = min(lo_y, red[1]);
        hi_x = max(hi_x, red[2]); hi_y = max(hi_y, red[3]);
        bad = max(bad, red[4]);
      }
      int4 bd;   // ShrinkToFit
      bd.x = max(-P.nl, lo_x);
      bd.y = min(P.nl, hi_x);
      bd.z = max(-P.nl, lo_y);
      bd.w = min(P.nl, hi_y);
      P.bounds[s0 + g] = bd;
      const int2 dims = make_int2((bd.y - bd.x + step) / step, (bd.w - bd.z + step) / step);
      P.coarse_dims[s0 + g] = dims;
      int* mine = misc + 16 + 8 * g;
      mine[0] = bd.x; mine[1] = bd.y; mine[2] = bd.z; mine[3] = bd.w;
      mine[4] = dims.x; mine[5] = dims.y;
      dims_all.x = max(dims_all.x, dims.x);
      dims_all.y = max(dims_all.y, dims.y);
    }
    if (gcount > 1) {
      far = (P.group_verify & 2) ? 1 : 0;       // (tests: every unit as if its premise had failed)
      for (int w = 0; w < waves; ++w) far |= misc[100 + w];
      for (int g = 0; g < gcount; ++g)
        far |= (abs(misc[16 + 8 * g] - misc[16 + 8 * gm]) > 1 ||
                abs(misc[16 + 8 * g + 2] - misc[16 + 8 * gm + 2]) > 1) ? 1 : 0;
    }
    if (bad) atomicMax(&states[blockIdx.y].error, 1);
    // (checked on the largest candidate grid of the unit: the accumulators of a group pass hold it)
    const int count = dims_all.x * dims_all.y;
    const int BW = dims_all.x + P.plane_i - 1, BH = dims_all.y + P.plane_j - 1;
    // (block + 1 = bx * pitch + by + 1 travels in 16 bits, see the scoring loop)
    const int ok = count <= P.coarse_stride && count <= kMaxCoarsePerScan && BW <= 255 &&
                   BH <= 255 && (BW - 1) * (dims_all.y + 2 * P.plane_j - 2) + BH <= 65535 &&
                   (dims_all.x + 2 * P.plane_i - 2) * (dims_all.y + 2 * P.plane_j - 2) <= acc_cap;
    misc[1] = gcount; misc[2] = gm;
    misc[4] = dims_all.x; misc[5] = dims_all.y;
    misc[6] = ok;
    misc[7] = far;
    if (far) atomicAdd(&states[blockIdx.y].done_top, 1);      // (statistics: units without a group bound)
    if (!ok) {
      atomicMax(&states[blockIdx.y].error, 2);
      for (int g = 0; g < gcount; ++g) P.scan_best[s0 + g] = make_int2(0, 0);
    }
  }
  __syncthreads();
  if (!misc[6]) return;
  stamp(2);      // bounds known
  FusedPass<kTimeline>(P, &states[blockIdx.y], n, acc_cap, s0, blockIdx.y * gridDim.x + blockIdx.x);
  stamp(7);
}

// ---------------------------------------------------------------------------
// Branch and bound
// ---------------------------------------------------------------------------
// Node lists (frontiers, leaves) are split into kSubLists sub-lists, each with
// its own counter, so that thousands of blocks appending at once do not
// serialise on one atomic word (one word sustains only ~90 atomics/us).
constexpr int kSubLists = 64;

constexpr int kMaxStages = kMaxDepth + 2;   // one frontier counter array per search stage

// Sub-list counters sit one per 128-byte line: returning atomics on words of the same line
// queue behind each other in L2 (64 adjacent counters = 2 lines took every list reservation of
// a batch through two queues).
constexpr int kCountStride = 32;
// The work queue of TreeQueueKernel (below): kQueues sub-queues, control words one 128-byte line
// per sub-queue (a word takes ~90 atomics per microsecond, and atomics on one line queue behind
// each other).
constexpr int kQueues = 2048;
struct alignas(8) QueueCtl {
  int head;           // nodes taken (CAS)
  int reserved;       // slots reserved by producers (every one of them gets published)
  int pad[30];        // (a pop reads the pair with ONE 8-byte load: both only grow, so halves of
};                    // different ages are harmless -- the CAS on `head` decides)
struct Counters {           // device, zeroed per call
  int frontier[kMaxStages][kSubLists * kCountStride];
  int leaves[kSubLists * kCountStride];
  int frontier_overflow;
  int leaf_overflow;
  unsigned wave_gathers;      // 64-lane quad gathers issued by the wave-per-node expansion (statistics)
  int blocks_done;            // TreeQueueKernel: workgroups that have run out of work
  int pad[28];
  unsigned gathers_shard[16 * kCountStride];   // TreeQueueKernel's share of wave_gathers, by workgroup
  unsigned queue_stats[16 * kCountStride];     // [shard][8] trace counters of the queue (CountersSummary)
  QueueCtl queue[kQueues];
};

// What the host needs of the counters, written next to the results by the last kernel of a
// search (the padded Counters are 120 KB: not something to copy back per match).
struct CountersSummary {
  int leaves[kSubLists];
  int frontier_total[kMaxStages];
  int frontier_overflow;
  int leaf_overflow;
  unsigned wave_gathers;
  int pad;
  unsigned queue_stats[8];    // trace: pops, lost races, slot re-reads, pushed nodes, list nodes, chains, -, -
};

struct NodeList {
  Node2D* nodes;      // [kSubLists][sub_capacity]
  int* counts;        // [kSubLists] at stride kCountStride
  int sub_capacity;
};

// Reserves `m` consecutive slots of sub-list `sub`; returns the first slot.
__device__ __forceinline__ int ListReserve(const NodeList& list, int sub, int m) {
  return atomicAdd(&list.counts[sub * kCountStride], m);
}
__device__ __forceinline__ bool ListStore(const NodeList& list, int sub, int slot,
                                          const Node2D& nd) {
  if (slot >= list.sub_capacity) return false;
  list.nodes[static_cast<size_t>(sub) * list.sub_capacity + slot] = nd;
  return true;
}
// Largest sub-list length (wave-uniform); every lane must call it.
__device__ __forceinline__ int ListMaxCount(const NodeList& list) {
  const int lane = threadIdx.x & 63;
  return WaveMax(min(list.counts[lane * kCountStride], list.sub_capacity));
}

__device__ __forceinline__ int NodeProblem(const Node2D& nd) { return nd.problem & 0xffffff; }
__device__ __forceinline__ int NodeLevel(const Node2D& nd) { return nd.problem >> 24; }

__device__ __forceinline__ Node2D CoarseNode(const Fast2DProblem& P, int problem, int s,
                                             int local) {
  const int2 dims = P.coarse_dims[s];
  const int4 bd = P.bounds[s];
  const int step = 1 << (P.depth - 1);
  const int ix = local / dims.y, iy = local - ix * dims.y;
  const int c = s * P.coarse_stride + local;
  Node2D nd;
  nd.problem = problem | ((P.depth - 1) << 24);
  nd.scan = s;
  nd.dx = bd.x + ix * step;
  nd.dy = bd.z + iy * step;
  nd.score = P.coarse_score[c];
  nd.coarse_index = c;
  nd.path = 0;
  nd.coarse_score = nd.score;
  return nd;
}

// Seeds of the dive: the best candidate of each of the ~64 best scans (histogram
// threshold on the per-scan maxima).  Every dive block repeats the selection for its own
// problem -- 18 KB of per-scan maxima, L2-resident after the first block -- and takes
// seed number `want` in scan order: no separate launch (a kernel that does this alone
// costs ~5 us plus its boundary), no cross-block hand-off.  When want == 0 the block also
// totals the problem's lowest-resolution candidates (the layout needs no prefix sum).
struct SeedScratch {
  int hist[1024];
  int wave_total[4];
  int threshold_bin;
  int found_scan;
  int total;
};

__device__ __forceinline__ bool PickSeed(const Fast2DProblem& P, int n, int want,
                                         SeedScratch* sh, int* scan_out) {
  const int tid = threadIdx.x, lane = tid & 63, T = blockDim.x;
  for (int i = tid; i < 1024; i += T) sh->hist[i] = 0;
  if (tid == 0) { sh->found_scan = -1; sh->total = 0; }
  __syncthreads();
  const int S = P.num_scans;
  const long long range = 255ll * n + 1;
  const auto* scan_best = AsGlobal(P.scan_best);
  const auto* coarse_dims = AsGlobal(P.coarse_dims);
  // Thread t owns scans t, t + T, ...: coalesced, independent loads (a contiguous chunk per
  // thread was a chain of L2 round trips); the first kOwn maxima stay in registers for the
  // second pass.
  // Under group bounds the rotations of a unit share their sums: only the one whose cells were
  // summed (the middle one) stands for its unit, or the seeds would be a third as many units.
  const bool grouped = P.group > 1;
  const auto best_of = [&](int s) -> int {
    if (grouped) {
      const int first = s - s % kFusedGroup;
      if (s != first + (S - first >= kFusedGroup ? 1 : 0)) return -1;
    }
    return scan_best[s].x;
  };
  constexpr int kOwn = 16;
  int own[kOwn];
#pragma unroll
  for (int k = 0; k < kOwn; ++k) {
    const int s = tid + k * T;
    own[k] = s < S ? best_of(s) : -1;
  }
  int total = 0;
#pragma unroll
  for (int k = 0; k < kOwn; ++k)
    if (own[k] >= 0) atomicAdd(&sh->hist[static_cast<int>(own[k] * 1024ll / range)], 1);
  for (int s = tid + kOwn * T; s < S; s += T) {
    const int b = best_of(s);
    if (b >= 0) atomicAdd(&sh->hist[static_cast<int>(b * 1024ll / range)], 1);
  }
  if (want == 0) {
    for (int s = tid; s < S; s += T) total += coarse_dims[s].x * coarse_dims[s].y;
    total = WaveSum(total);
    if (lane == 0 && total) atomicAdd(&sh->total, total);
  }
  __syncthreads();
  if (tid < 64) {
    // Highest bin b >= 1 with (number of scans in bins >= b) >= kSeedsPerProblem, else 0.
    // Lane l owns bins 1023 - 16 l down to 1008 - 16 l.
    int mine = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) mine += sh->hist[1023 - 16 * lane - k];
    int incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int o = __shfl_up(incl, off, 64);
      if (lane >= off) incl += o;
    }
    const unsigned long long reached = __ballot(incl >= kSeedsPerProblem);
    int tb = 0;
    if (reached) {
      const int first = __ffsll(static_cast<long long>(reached)) - 1;
      if (lane == first) {
        int acc = incl - mine;
        for (int k = 0; k < 16; ++k) {
          const int b = 1023 - 16 * lane - k;
          acc += sh->hist[b];
          if (acc >= kSeedsPerProblem) { tb = b; break; }   // b == 0 only in lane 63: tb = 0
        }
        sh->threshold_bin = tb;
      }
    } else if (lane == 0) {
      sh->threshold_bin = 0;
    }
  }
  __syncthreads();
  const int tb = sh->threshold_bin;
  const auto qualifies = [&](int best) {
    return best >= 0 && static_cast<int>(best * 1024ll / range) >= tb &&
           ToScore(P, best, n) > P.min_score;
  };
  // Seeds are numbered thread-major: thread t's qualifying scans (in its own order) follow
  // those of threads < t.
  int count = 0;
#pragma unroll
  for (int k = 0; k < kOwn; ++k) count += qualifies(own[k]) ? 1 : 0;
  for (int s = tid + kOwn * T; s < S; s += T) count += qualifies(best_of(s)) ? 1 : 0;
  int incl = count;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int o = __shfl_up(incl, off, 64);
    if (lane >= off) incl += o;
  }
  if (lane == 63) sh->wave_total[tid >> 6] = incl;
  __syncthreads();
  int before = incl - count;
  for (int w = 0; w < (tid >> 6); ++w) before += sh->wave_total[w];
  if (want >= before && want < before + count) {
    int k = before, found = -1;
#pragma unroll
    for (int j = 0; j < kOwn; ++j) {
      if (found < 0 && qualifies(own[j])) {
        if (k == want) found = tid + j * T;
        ++k;
      }
    }
    for (int s = tid + kOwn * T; s < S && found < 0; s += T) {
      if (!qualifies(best_of(s))) continue;
      if (k == want) found = s;
      ++k;
    }
    sh->found_scan = found;
  }
  __syncthreads();
  *scan_out = sh->found_scan;
  return sh->found_scan >= 0;
}

// Cell of point i of rotated scan `rot` = P.scan_rot[scan], packed (x | y << 16): the fused
// front end's arithmetic (RotateZ twice, translation, CellIndexFast), so bit-identical to what
// it scored -- and to PrepScansKernel's `discrete` array.
__device__ __forceinline__ uint32_t ScanCell(const Fast2DProblem& P, float2 rot, int i) {
  const float* __restrict__ xyz = P.xyz;
  const float px = xyz[3 * i], py = xyz[3 * i + 1];
  float ax = px, ay = py;
  if (!(P.init_qw == 1.f && P.init_qz == 0.f)) RotateZ(P.init_qw, P.init_qz, px, py, &ax, &ay);
  float bx, by;
  RotateZ(rot.x, rot.y, ax, ay, &bx, &by);
  const float x = bx + P.tx;
  const float y = by + P.ty;
  const int ix = CellIndexFast(P.max_y, y, P.res, P.inv_res);
  const int iy = CellIndexFast(P.max_x, x, P.res, P.inv_res);
  return (static_cast<uint32_t>(ix) & 0xffffu) | (static_cast<uint32_t>(iy) << 16);
}

// Problem- and scan-invariant data a block keeps on chip while it works on
// nodes of one rotated scan: level descriptors and score constants (so that a
// node expansion starts without dependent global loads), the discretised
// points (LDS when they fit) and the search bounds of the scan.
constexpr int kPointCache = 4096;   // points kept in LDS (16 KB)

struct BlockContext {
  LevelDesc level[kMaxDepth];
  uint32_t cache[kPointCache];
  const uint32_t* global_pts;
  float min_s, score_scale, min_score;
  int n, cached;
  int max_x, max_y;    // linear_bounds[scan].max_x / max_y
};

__device__ __forceinline__ void LoadContext(const Fast2DProblem& P, int n, int scan,
                                            BlockContext* ctx, bool stored = false) {
  const uint32_t* pts = P.discrete + static_cast<size_t>(scan) * n;
  const bool cached = n <= kPointCache;
  if (P.recompute_scans && !stored) {   // (implies n <= kFusedMaxPoints = kPointCache)
    const float2 rot = P.scan_rot[scan];
    for (int i = threadIdx.x; i < n; i += blockDim.x) ctx->cache[i] = ScanCell(P, rot, i);
  } else if (cached) {
    const auto* gp = AsGlobal(pts);
    for (int i = threadIdx.x; i < n; i += blockDim.x) ctx->cache[i] = gp[i];
  }
  if (threadIdx.x < kMaxDepth && static_cast<int>(threadIdx.x) < P.depth)
    ctx->level[threadIdx.x] = P.level[threadIdx.x];
  if (threadIdx.x == 64) {
    const int4 bd = P.bounds[scan];
    ctx->max_x = bd.y;
    ctx->max_y = bd.w;
    ctx->global_pts = pts;
    ctx->cached = cached;
    ctx->n = n;
    ctx->min_s = P.min_s;
    ctx->score_scale = P.score_scale;
    ctx->min_score = P.min_score;
  }
  __syncthreads();
}

__device__ __forceinline__ float ToScoreCtx(const BlockContext& ctx, int sum) {
  return ctx.min_s + (static_cast<float>(sum) / static_cast<float>(ctx.n)) * ctx.score_scale;
}

// Scores the <=4 children of a node (SM2/fast_...2d.cc:351-368) with the whole
// 256-thread block: every wave takes a quarter of the points and gathers all
// four children per point.  child_score[k] < 0 marks a child outside the
// search bounds; rank[k] is the position of child k in the reference's stable
// descending sort of the children.
struct ChildScratch {
  int partial[4][4];
  float child_score[4];
  int rank[4];
  int nvalid;
};

__device__ __forceinline__ void ScoreChildren(const BlockContext& ctx, int dx, int dy,
                                              int child_level, ChildScratch* sh) {
  const int n = ctx.n;
  const LevelDesc L = ctx.level[child_level];
  const int half = 1 << child_level;
  const int off = half - 1;
  const bool vx = dx + half <= ctx.max_x, vy = dy + half <= ctx.max_y;  // `break`s at :356,361
  const bool cached = ctx.cached;
  const auto* gpts = AsGlobal(ctx.global_pts);
  const auto* quads = AsGlobal(L.quads);
  // Children beyond the search bounds (`break`s at :356,361) are masked out of the quad.
  const uint32_t child_mask = (vx ? 0xffffffffu : 0x0000ffffu) & (vy ? 0xffffffffu : 0x00ff00ffu);
  // Packed accumulators: (s00 | s10 << 16) and (s01 | s11 << 16); a lane adds at most
  // 255 per point, so 256 points fit before the halves are widened.
  int s00 = 0, s01 = 0, s10 = 0, s11 = 0;   // s[x-step][y-step]
  // The gathers of the unrolled loop are meant to be in flight together.  Until the end of
  // round 3 they were not: `cached ? ctx.cache[i] : gpts[i]` inside the body gave every unrolled
  // iteration a branch and a basic block of its own, closed with s_waitcnt vmcnt(0) lgkmcnt(0) --
  // which also waited for the previous iteration's gather -- and the plain
  // `quads[inside ? offset : 0]` became a load under an exec mask.  Now: one loop per source of
  // the cells (compile-time), quads by buffer loads (out-of-range offsets read 0; a level of more
  // than 2 GB of quads keeps plain loads).
  // (the level comes out of LDS: the compiler does not take it for wavefront-uniform and would
  // wrap every buffer load in a waterfall loop -- readfirstlane says it is)
  const unsigned long long quads_address = reinterpret_cast<unsigned long long>(L.quads);
  // (readfirstlane returns an int: through `unsigned`, or a low word with its top bit set
  // sign-extends over the high word -- the first version of this faulted on exactly that)
  const unsigned long long quads_uniform =
      static_cast<unsigned long long>(static_cast<unsigned>(
          __builtin_amdgcn_readfirstlane(static_cast<unsigned>(quads_address)))) |
      (static_cast<unsigned long long>(static_cast<unsigned>(
           __builtin_amdgcn_readfirstlane(static_cast<unsigned>(quads_address >> 32)))) << 32);
  const unsigned long long quad_bytes =
      static_cast<unsigned long long>((__builtin_amdgcn_readfirstlane(L.qy) + 3) >> 2) *
      static_cast<unsigned>(__builtin_amdgcn_readfirstlane(L.qtx)) * 128ull;
  const bool quads_by_buffer = quad_bytes < (1ull << 31);
  const __amdgpu_buffer_rsrc_t quad_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<uint32_t*>(quads_uniform), 0,
      quads_by_buffer ? static_cast<int>(quad_bytes) : 0, 0x00020000);
  const auto walk = [&](auto cached_tag, auto buffer_tag) {
    constexpr bool kCached = decltype(cached_tag)::value;
    constexpr bool kBuffer = decltype(buffer_tag)::value;
    for (int base = threadIdx.x; base < n; base += 256 * 256) {
      uint32_t even = 0, odd = 0;
      const int stop = min(n, base + 256 * 256);
#pragma unroll 4
      for (int i = base; i < stop; i += 256) {
        uint32_t p;
        if constexpr (kCached) p = ctx.cache[i];
        else p = gpts[i];
        const int X = static_cast<short>(p & 0xffffu) + dx + off + half;
        const int Y = static_cast<short>(p >> 16) + dy + off + half;
        const bool inside = static_cast<unsigned>(X) < static_cast<unsigned>(L.qx) &&
                            static_cast<unsigned>(Y) < static_cast<unsigned>(L.qy);
        uint32_t v;
        if constexpr (kBuffer) {
          v = __builtin_amdgcn_raw_buffer_load_b32(
                  quad_rsrc, inside ? QuadOffset(X, Y, L.qtx) * 4u : 0xfffffff0u, 0, 0) &
              child_mask;
        } else {
          const uint32_t q = quads[inside ? QuadOffset(X, Y, L.qtx) : 0u];
          v = inside ? (q & child_mask) : 0u;
        }
        even += v & 0x00ff00ffu;          // byte 0 (x0,y0) and byte 2 (x1,y0)
        odd += (v >> 8) & 0x00ff00ffu;    // byte 1 (x0,y1) and byte 3 (x1,y1)
      }
      s00 += even & 0xffffu; s10 += even >> 16;
      s01 += odd & 0xffffu;  s11 += odd >> 16;
    }
  };
  if (quads_by_buffer) {
    if (cached) walk(std::true_type{}, std::true_type{});
    else walk(std::false_type{}, std::true_type{});
  } else {
    if (cached) walk(std::true_type{}, std::false_type{});
    else walk(std::false_type{}, std::false_type{});
  }
  s00 = WaveSum(s00); s01 = WaveSum(s01); s10 = WaveSum(s10); s11 = WaveSum(s11);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    sh->partial[wave][0] = s00; sh->partial[wave][1] = s01;
    sh->partial[wave][2] = s10; sh->partial[wave][3] = s11;
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    // Lanes 0..3 finish child k = 2*x-step + y-step (generation order: x outer,
    // y inner) and rank them through wave shuffles.
    const int k = threadIdx.x & 3;
    const bool valid = ((k >> 1) == 0 || vx) && ((k & 1) == 0 || vy);
    const int total = sh->partial[0][k] + sh->partial[1][k] + sh->partial[2][k] + sh->partial[3][k];
    const float mine = valid ? ToScoreCtx(ctx, total) : -1.f;
    int rank = 0, nvalid = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float other = __shfl(mine, j, 64);
      if (other >= 0.f) ++nvalid;
      if (j != k && other >= 0.f && (other > mine || (other == mine && j < k))) ++rank;
    }
    if (threadIdx.x < 4) {
      sh->child_score[k] = mine;
      sh->rank[k] = rank;
      if (k == 0) sh->nvalid = nvalid;
    }
  }
  __syncthreads();
}

__device__ __forceinline__ Node2D MakeChild(const Node2D& nd, int k, int child_level,
                                            const ChildScratch& sh) {
  Node2D child = nd;
  child.problem = NodeProblem(nd) | (child_level << 24);
  child.dx = nd.dx + (k >> 1) * (1 << child_level);
  child.dy = nd.dy + (k & 1) * (1 << child_level);
  child.score = sh.child_score[k];
  child.path = nd.path | (static_cast<unsigned>(sh.rank[k]) << (2 * child_level));
  return child;
}

__device__ __forceinline__ void RecordLeaf(const Node2D& leaf, const NodeList& leaves,
                                           Counters* __restrict__ counters) {
  const int sub = blockIdx.x & (kSubLists - 1);
  if (!ListStore(leaves, sub, ListReserve(leaves, sub, 1), leaf)) counters->leaf_overflow = 1;
}

// One greedy descent per seed: always continue with the best child.  The leaf
// reached is a real candidate, so its score is a valid bound.
__global__ void __launch_bounds__(256, 4)
DiveKernel(const Fast2DProblem* __restrict__ problems, ProblemState* __restrict__ states, int n,
           NodeList leaves, Counters* __restrict__ counters) {
  const int problem = blockIdx.y;
  const Fast2DProblem& P = problems[problem];
  ProblemState& st = states[problem];
  if (st.error) return;
  __shared__ BlockContext ctx;
  __shared__ ChildScratch sh;
  __shared__ Node2D cur;
  __shared__ SeedScratch seed_scratch;
  // Under group bounds a seed is a unit of three rotations (PickSeed) whose bound says nothing
  // about which of them holds the good leaf: a dive per rotation (blocks 3 k, 3 k + 1, 3 k + 2).
  const int per_seed = P.group > 1 ? kFusedGroup : 1;
  if (static_cast<int>(blockIdx.x) >= kSeedsPerProblem * per_seed) return;
  int seed_scan;
  const bool have_seed = PickSeed(P, n, blockIdx.x / per_seed, &seed_scratch, &seed_scan);
  if (blockIdx.x == 0 && threadIdx.x == 0) st.coarse_total = seed_scratch.total;
  if (!have_seed) return;
  if (per_seed > 1) {
    seed_scan = seed_scan - seed_scan % kFusedGroup + static_cast<int>(blockIdx.x) % per_seed;
    if (seed_scan >= P.num_scans) return;
  }
  const Node2D seed = CoarseNode(P, problem, seed_scan, P.scan_best[seed_scan].y);
  if (threadIdx.x == 0) cur = seed;
  LoadContext(P, n, seed.scan, &ctx);
  const int depth = NodeLevel(seed) + 1;
  unsigned long long scored = 0;
  for (int child_level = depth - 2; child_level >= 0; --child_level) {
    const Node2D nd = cur;
    ScoreChildren(ctx, nd.dx, nd.dy, child_level, &sh);
    scored += sh.nvalid;
    if (threadIdx.x == 0) {
      for (int k = 0; k < 4; ++k)
        if (sh.child_score[k] >= 0.f && sh.rank[k] == 0) cur = MakeChild(nd, k, child_level, sh);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const Node2D leaf = cur;
    if (leaf.score > ctx.min_score) {
      RecordLeaf(leaf, leaves, counters);
      atomicMax(&st.best_bits, __float_as_uint(leaf.score));
    }
    atomicAdd(&st.scored_shard[blockIdx.x & (kStatShards - 1)], scored);
    atomicAdd(&st.expanded_shard[blockIdx.x & (kStatShards - 1)],
              static_cast<unsigned long long>(depth - 1));
  }
}

// Lowest-resolution nodes that can still matter (reference: :346-350): one
// block per scan; whole scans are skipped through their best candidate.
// strict = 0 keeps nodes equal to the bound so that every leaf tied for the
// best score is found.
__global__ void __launch_bounds__(256)
FilterCoarseKernel(const Fast2DProblem* __restrict__ problems,
                   const ProblemState* __restrict__ states, int n, int chunk, int num_chunks,
                   int strict, int affinity, NodeList out, Counters* __restrict__ counters) {
  // One WAVEFRONT per rotated scan (grid: ceil(scans / 4) x problems): a scan's filter is a
  // chain of dependent loads (problem, best candidate, dimensions, scores, list slot) over
  // ~190 candidates; a block per scan kept 8 of those chains in flight per CU, and 36 k blocks
  // of a 16-submap batch took 320 us to dispatch and drain.
  const int problem = blockIdx.y;
  const Fast2DProblem& P = problems[problem];
  const int lane = threadIdx.x & 63;
  const int s = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (s >= P.num_scans || states[problem].error) return;
  if (s % num_chunks != chunk) return;
  const float best = __uint_as_float(states[problem].best_bits);
  const float top = ToScore(P, P.scan_best[s].x, n);
  if (strict ? !(top > best) : (top < best)) return;
  const int2 dims = P.coarse_dims[s];
  const int count = dims.x * dims.y;
  const int base = s * P.coarse_stride;
  // The first 256 scores are fetched together (the passes below would otherwise be a chain of
  // load -> ballot -> list reservation round trips).
  const auto* scores = AsGlobal(P.coarse_score) + base;
  float ahead[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) ahead[k] = k * kWave + lane < count ? scores[k * kWave + lane] : 0.f;
  for (int c0 = 0; c0 < count; c0 += kWave) {
    // One reservation per wave; consecutive waves use consecutive sub-lists, so that one
    // rotation's survivors do not all queue in the same one.
    // With `affinity` a problem's nodes only go to the sub-lists the workgroups of ONE XCD
    // read (sub % 8 == problem % 8, see ExpandWaveKernel): the level data of that problem
    // then lives in one L2 instead of eight.
    const int sub = affinity ? (problem & 7) + 8 * ((s + (c0 >> 6)) & 7)
                             : (s + blockIdx.y + (c0 >> 6)) & (kSubLists - 1);
    const int c = c0 + lane;
    bool keep = false;
    if (c < count) {
      float score;
      switch (c0 >> 6) {
        case 0: score = ahead[0]; break;
        case 1: score = ahead[1]; break;
        case 2: score = ahead[2]; break;
        case 3: score = ahead[3]; break;
        default: score = scores[c];
      }
      keep = strict ? (score > best) : (score >= best);
    }
    // One reservation per wave: slots go to the kept lanes in lane order.
    const unsigned long long mask = __ballot(keep);
    if (mask == 0) continue;
    int first = 0;
    if (lane == 0) first = ListReserve(out, sub, __popcll(mask));
    first = __builtin_amdgcn_readfirstlane(first);
    if (keep) {
      const int slot = first + __popcll(mask & ((1ull << lane) - 1));
      if (!ListStore(out, sub, slot, CoarseNode(P, problem, s, c))) counters->frontier_overflow = 1;
    }
  }
}

// Level-synchronous expansion near the top of the tree, one WAVE per node
// (four independent nodes per block, no LDS, no barriers).  The top levels
// hold thousands of nodes most of which die after one expansion, so what
// matters there is how many nodes are in flight, not the latency of one.
// Children that can still matter go to `out`; child_level is always >= 1 here.
constexpr int kWaveStatProblems = 1024;   // problems whose work counters a block keeps in LDS

__global__ void __launch_bounds__(256)
ExpandWaveKernel(const Fast2DProblem* __restrict__ problems, ProblemState* __restrict__ states,
                 int n, NodeList in, int strict, int affinity, NodeList out,
                 Counters* __restrict__ counters) {
  // Work counters (candidates scored / nodes expanded per problem) are collected in LDS and
  // flushed once per block: one global atomic pair PER NODE -- half a million nodes of 16
  // problems hammering 32 cache lines -- was 64 % of this kernel on a 16-submap batch
  // (3.36 -> 1.21 ms, profiles/r02_c3_wave_atomics.txt).
  __shared__ unsigned stat_scored[kWaveStatProblems], stat_expanded[kWaveStatProblems];
  __shared__ unsigned stat_gathers;
  if (threadIdx.x == 0) stat_gathers = 0;
  for (int i = threadIdx.x; i < kWaveStatProblems; i += blockDim.x) {
    stat_scored[i] = 0;
    stat_expanded[i] = 0;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int max_count = ListMaxCount(in);
  // Workgroups go to the XCDs round-robin (blockIdx.x % 8).  With `affinity` the waves of XCD
  // x read and write only the sub-lists with sub % 8 == x (grid: a multiple of 16 blocks), so
  // that a node's children are expanded on the XCD whose L2 already holds that problem.
  int first = blockIdx.x * 4 + wave;
  if (affinity) {
    const int slot = (blockIdx.x >> 3) * 4 + wave;            // wave index within the XCD
    first = (slot >> 3) * kSubLists + (blockIdx.x & 7) + 8 * (slot & 7);
  }
  const int out_sub = first & (kSubLists - 1);
  for (int i = first; i < max_count * kSubLists; i += gridDim.x * 4) {
    const int in_sub = i & (kSubLists - 1), j = i / kSubLists;
    if (j >= in.counts[in_sub * kCountStride]) continue;   // wave-uniform
    const Node2D nd = in.nodes[static_cast<size_t>(in_sub) * in.sub_capacity + j];
    const int problem = NodeProblem(nd);
    const Fast2DProblem& P = problems[problem];
    ProblemState& st = states[problem];
    const float best = __uint_as_float(st.best_bits);
    if (strict ? !(nd.score > best) : (nd.score < best)) continue;
    const int child_level = NodeLevel(nd) - 1;
    const LevelDesc L = P.level[child_level];
    const int4 bd = P.bounds[nd.scan];
    const int half = 1 << child_level, off = half - 1;
    const bool vx = nd.dx + half <= bd.y, vy = nd.dy + half <= bd.w;
    const auto* pts = AsGlobal(P.discrete) + static_cast<size_t>(nd.scan) * n;
    const bool recompute = P.recompute_scans != 0 && P.store_scans == 0;
    const float2 rot = recompute ? P.scan_rot[nd.scan] : make_float2(1.f, 0.f);
    // Early exit.  A level-(l+1) cell is the maximum of the four level-l cells its
    // children read (the 2h window is tiled by four h windows), so for every point
    // max(children) <= parent value and
    //   child_k total <= partial_k + (parent total - sum of max(children) so far).
    // Most frontier nodes pass their own bound only marginally: after a few dozen
    // points no child can reach the bound any more and the rest of the gathers
    // (the expensive part: 64 distinct cache lines each) is skipped.  The outcome
    // is the same as scoring all points: no child would have been kept.
    const int parent_ub = SumUpperBound(P, nd.score, n);
    // (the tiled quad array: ceil(qy / 4) rows of qtx tiles of 32 dwords; a buffer resource
    // addresses up to 2 GB of it -- a level of more than 23 000 x 23 000 cells keeps plain loads)
    const unsigned long long quad_bytes =
        static_cast<unsigned long long>((L.qy + 3) >> 2) * static_cast<unsigned>(L.qtx) * 128ull;
    const bool quads_by_buffer = quad_bytes < (1ull << 31);
    const __amdgpu_buffer_rsrc_t quad_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint32_t*>(L.quads), 0, quads_by_buffer ? static_cast<int>(quad_bytes) : 0,
        0x00020000);
    const auto* quads = AsGlobal(L.quads);
    const uint32_t child_mask =
        (vx ? 0xffffffffu : 0x0000ffffu) & (vy ? 0xffffffffu : 0x00ff00ffu);
    int s00 = 0, s01 = 0, s10 = 0, s11 = 0, seen_max = 0;
    int groups = 0;
    bool dead = false;
    // 64-point iterations gathered between two bound checks (1, 2 and 4 measure the same on a
    // 16-submap batch; 16, i.e. everything in flight at once, was no faster for single
    // searches: 37 vs 33 us).
    constexpr int kIters = 4;
    constexpr int kGroup = kIters * kWave;
    // (The loop is instantiated twice, for stored and for re-derived scan cells.  With the choice
    // made per point -- `recompute ? ScanCell(...) : pts[...]` inside the unrolled body, as it
    // stood until the end of round 3 -- every one of the four "gathers in flight" began with a
    // branch and a basic block of its own, and the compiler closed each with s_waitcnt vmcnt(0):
    // ONE gather in flight per wavefront, sixteen dependent round trips per node.)
    const auto walk = [&](auto recompute_tag, auto buffer_tag) {
      constexpr bool kRecompute = decltype(recompute_tag)::value;
      constexpr bool kBuffer = decltype(buffer_tag)::value;
      for (int q0 = 0; q0 < n; q0 += kGroup) {
        uint32_t cell[kIters];
#pragma unroll
        for (int u = 0; u < kIters; ++u) {      // the four cells first: independent loads
          const int q = q0 + u * kWave + lane;
          const int at = q < n ? q : 0;
          if constexpr (kRecompute) cell[u] = ScanCell(P, rot, at);
          else cell[u] = pts[at];
        }
        uint32_t v[kIters];
#pragma unroll
        for (int u = 0; u < kIters; ++u) {
          const int q = q0 + u * kWave + lane;
          const bool live = q < n;
          const uint32_t p = cell[u];
          const int X = static_cast<short>(p & 0xffffu) + nd.dx + off + half;
          const int Y = static_cast<short>(p >> 16) + nd.dy + off + half;
          const bool inside = live && static_cast<unsigned>(X) < static_cast<unsigned>(L.qx) &&
                              static_cast<unsigned>(Y) < static_cast<unsigned>(L.qy);
          // one gather, four children.  A buffer load: the compiler turned the plain
          // `quads[inside ? offset : 0]` into a load under an exec mask with its own s_waitcnt --
          // the four gathers went out one after the other.  Out-of-range offsets read 0.
          if constexpr (kBuffer) {
            const uint32_t quad = __builtin_amdgcn_raw_buffer_load_b32(
                quad_rsrc, inside ? QuadOffset(X, Y, L.qtx) * 4u : 0xfffffff0u, 0, 0);
            v[u] = quad & child_mask;
          } else {
            const uint32_t quad = quads[inside ? QuadOffset(X, Y, L.qtx) : 0u];
            v[u] = inside ? (quad & child_mask) : 0u;
          }
        }
#pragma unroll
        for (int u = 0; u < kIters; ++u) {
          const int a00 = v[u] & 0xff, a01 = (v[u] >> 8) & 0xff;
          const int a10 = (v[u] >> 16) & 0xff, a11 = v[u] >> 24;
          s00 += a00; s01 += a01; s10 += a10; s11 += a11;
          seen_max += max(max(a00, a01), max(a10, a11));
        }
        ++groups;
        if (q0 + kGroup < n) {
          // max_k sum_lanes(s_k) <= sum_lanes(max_k s_k): ONE wavefront reduction (of what the
          // best child of each lane's points still lacks to the parent) instead of five -- a
          // slightly looser bound, the same results (the check only skips work that cannot
          // matter), 8 of 54 vector instructions per gather less.
          const int lacking = seen_max - max(max(s00, s01), max(s10, s11));
          // kept children satisfy score >= best (> best in strict mode)
          if (ToScore(P, parent_ub - WaveSum(lacking), n) < best) { dead = true; break; }
        }
      }
    };
    if (quads_by_buffer) {
      if (recompute) walk(std::true_type{}, std::true_type{});
      else walk(std::false_type{}, std::true_type{});
    } else {
      if (recompute) walk(std::true_type{}, std::false_type{});
      else walk(std::false_type{}, std::false_type{});
    }
    const auto count_node = [&](int nvalid) {     // lane 0
      atomicAdd(&stat_gathers, static_cast<unsigned>(min(groups * kIters, (n + kWave - 1) / kWave)));
      if (problem < kWaveStatProblems) {
        atomicAdd(&stat_scored[problem], static_cast<unsigned>(nvalid));
        atomicAdd(&stat_expanded[problem], 1u);
      } else {
        atomicAdd(&st.scored_shard[out_sub & (kStatShards - 1)],
                  static_cast<unsigned long long>(nvalid));
        atomicAdd(&st.expanded_shard[out_sub & (kStatShards - 1)], 1ull);
      }
    };
    if (dead) {   // every child provably below the bound: counted, not kept
      if (lane == 0) count_node((1 + (vx ? 1 : 0)) * (1 + (vy ? 1 : 0)));
      continue;
    }
    const int total[4] = {WaveSum(s00), WaveSum(s01), WaveSum(s10), WaveSum(s11)};
    ChildScratch cs;   // wave-uniform, in registers
    int nvalid = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const bool valid = ((k >> 1) == 0 || vx) && ((k & 1) == 0 || vy);
      cs.child_score[k] = valid ? ToScore(P, total[k], n) : -1.f;
      nvalid += valid;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int rank = 0;
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        if (o != k && cs.child_score[o] >= 0.f &&
            (cs.child_score[o] > cs.child_score[k] ||
             (cs.child_score[o] == cs.child_score[k] && o < k)))
          ++rank;
      }
      cs.rank[k] = rank;
    }
    if (lane == 0) {
      count_node(nvalid);
      int keep_mask = 0, m = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float sc = cs.child_score[k];
        if (sc < 0.f) continue;
        if (strict ? !(sc > best) : (sc < best)) continue;
        keep_mask |= 1 << k;
        ++m;
      }
      if (m) {
        int slot = ListReserve(out, out_sub, m);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (!(keep_mask >> k & 1)) continue;
          if (!ListStore(out, out_sub, slot, MakeChild(nd, k, child_level, cs)))
            counters->frontier_overflow = 1;
          ++slot;
        }
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0 && stat_gathers) atomicAdd(&counters->wave_gathers, stat_gathers);
  for (int i = threadIdx.x; i < kWaveStatProblems; i += blockDim.x) {
    if (stat_expanded[i] == 0) continue;
    ProblemState& st = states[i];
    atomicAdd(&st.scored_shard[blockIdx.x & (kStatShards - 1)],
              static_cast<unsigned long long>(stat_scored[i]));
    atomicAdd(&st.expanded_shard[blockIdx.x & (kStatShards - 1)],
              static_cast<unsigned long long>(stat_expanded[i]));
  }
}

// Depth-first search of the subtree below each frontier node by one block,
// pruned with the problem's shared bound (reference: :335-378).  Nodes
// reaching `stop_level` (> 0) are handed to `out` instead of being searched
// (used once near the top to create enough independent roots); stop_level = 0
// searches down to the leaves, where only the first-best child can be
// returned by the reference (:340-343 after the stable sort of :331-332).
// strict = 0 keeps nodes / leaves EQUAL to the bound, so every leaf tied for
// the best score is recorded and the reference's visiting order can be
// reproduced.  The bound is re-read from global memory once per root and every
// 8 expansions; in between the block uses its own copy, raised by its own
// leaves (a stale bound only costs extra work).
__global__ void __launch_bounds__(256, 4)
SubtreeKernel(const Fast2DProblem* __restrict__ problems, ProblemState* __restrict__ states, int n,
              NodeList in, int stop_level, int strict, NodeList out, NodeList leaves,
              Counters* __restrict__ counters) {
  __shared__ BlockContext ctx;
  __shared__ ChildScratch cs;
  __shared__ Node2D stack[kMaxDepth * 3 + 4];
  __shared__ Node2D cur;
  __shared__ int sp, have;
  __shared__ float s_best;
  const int max_count = ListMaxCount(in);
  const int out_sub = blockIdx.x & (kSubLists - 1);
  for (int i = blockIdx.x; i < max_count * kSubLists; i += gridDim.x) {
    const int in_sub = i & (kSubLists - 1), j = i / kSubLists;
    if (j >= in.counts[in_sub * kCountStride]) continue;   // block-uniform
    const Node2D root = in.nodes[static_cast<size_t>(in_sub) * in.sub_capacity + j];
    const int problem = NodeProblem(root);
    const Fast2DProblem& P = problems[problem];
    ProblemState& st = states[problem];
    if (threadIdx.x == 0) {
      stack[0] = root;
      sp = 1;
      s_best = __uint_as_float(__hip_atomic_load(&st.best_bits, __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT));
    }
    LoadContext(P, n, root.scan, &ctx, P.store_scans != 0);   // ends with __syncthreads()
    unsigned long long scored = 0, expanded = 0;
    for (;;) {
      if (threadIdx.x == 0) {
        have = sp > 0;
        if (have) cur = stack[--sp];
        if (have && (expanded & 7) == 7)
          s_best = fmaxf(s_best, __uint_as_float(__hip_atomic_load(
                                     &st.best_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
      }
      __syncthreads();
      if (!have) break;
      const Node2D nd = cur;
      const float best = s_best;
      const bool skip = strict ? !(nd.score > best) : (nd.score < best);
      if (!skip) {
        const int child_level = NodeLevel(nd) - 1;
        ScoreChildren(ctx, nd.dx, nd.dy, child_level, &cs);
        scored += cs.nvalid;
        ++expanded;
        if (threadIdx.x == 0) {
          if (child_level == 0) {
            for (int k = 0; k < 4; ++k) {
              if (cs.child_score[k] >= 0.f && cs.rank[k] == 0) {
                const Node2D leaf = MakeChild(nd, k, 0, cs);
                const bool keep = strict ? (leaf.score > best) : (leaf.score >= best);
                if (leaf.score > ctx.min_score && keep) {
                  RecordLeaf(leaf, leaves, counters);
                  atomicMax(&st.best_bits, __float_as_uint(leaf.score));
                  s_best = fmaxf(s_best, leaf.score);
                }
              }
            }
          } else if (child_level == stop_level) {
            // Hand the surviving children to the next stage: one slot
            // reservation per expansion.
            int keep_mask = 0, m = 0;
            for (int k = 0; k < 4; ++k) {
              const float sc = cs.child_score[k];
              if (sc < 0.f) continue;
              if (strict ? !(sc > best) : (sc < best)) continue;
              keep_mask |= 1 << k;
              ++m;
            }
            if (m) {
              int slot = ListReserve(out, out_sub, m);
              for (int k = 0; k < 4; ++k) {
                if (!(keep_mask >> k & 1)) continue;
                if (!ListStore(out, out_sub, slot, MakeChild(nd, k, child_level, cs)))
                  counters->frontier_overflow = 1;
                ++slot;
              }
            }
          } else {
            // Push worst first so the best child is searched next.
            for (int r = 3; r >= 0; --r) {
              for (int k = 0; k < 4; ++k) {
                if (cs.child_score[k] < 0.f || cs.rank[k] != r) continue;
                const float sc = cs.child_score[k];
                if (strict ? !(sc > best) : (sc < best)) continue;
                stack[sp++] = MakeChild(nd, k, child_level, cs);
              }
            }
          }
        }
      }
      __syncthreads();
    }
    if (threadIdx.x == 0 && expanded) {
      atomicAdd(&st.scored_shard[blockIdx.x & (kStatShards - 1)], scored);
      atomicAdd(&st.expanded_shard[blockIdx.x & (kStatShards - 1)], expanded);
    }
    __syncthreads();
  }
}

// Best-leaf selection in the reference's depth-first visiting order among
// equal scores: higher-scoring lowest-resolution ancestor first (the sorted
// order of :331-332; equal ancestors fall back to generation order), then the
// sibling ranks down the tree.  One block; the leaf list is short.
struct SelectState {         // per problem, device
  unsigned best_coarse_bits;
  int ties;
  unsigned long long key;    // (coarse_index << 32) | path, minimised
};

// A word another workgroup of the SAME launch may have written (atomics, agent-scope stores):
// read past this CU's L1 (global_load ... sc1).
template <typename T>
__device__ __forceinline__ T LoadAgent(const T* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// kSameLaunch: the leaves, counters and problem states were written by other workgroups of the
// launch this runs in (TreeQueueKernel's last workgroup), not by an earlier launch: every word of
// them is read with agent-scope loads (the producers stored / updated them with agent-scope
// stores and atomics and drained their stores before they arrived on `blocks_done`).
template <bool kSameLaunch>
__device__ __forceinline__ void
SelectBestBody(NodeList leaves, const ProblemState* __restrict__ states,
               SelectState* __restrict__ sel, BestLeaf* __restrict__ best, int num_problems,
               ProblemState* __restrict__ states_out, const Counters* __restrict__ counters,
               CountersSummary* __restrict__ summary) {
  if constexpr (kSameLaunch) {
    static_assert(sizeof(ProblemState) % sizeof(unsigned) == 0, "copied by words");
    constexpr int kWords = sizeof(ProblemState) / sizeof(unsigned);
    const unsigned* from = reinterpret_cast<const unsigned*>(states);
    unsigned* to = reinterpret_cast<unsigned*>(states_out);
    for (int i = threadIdx.x; i < num_problems * kWords; i += blockDim.x) to[i] = LoadAgent(&from[i]);
  } else {
    for (int p = threadIdx.x; p < num_problems; p += blockDim.x) states_out[p] = states[p];
  }
  const auto word = [](const auto* p) {
    if constexpr (kSameLaunch) return LoadAgent(p);
    else return *p;
  };
  if (threadIdx.x < kSubLists) summary->leaves[threadIdx.x] = word(&counters->leaves[threadIdx.x * kCountStride]);
  if (threadIdx.x >= 64 && threadIdx.x < 64 + kMaxStages) {
    const int st = threadIdx.x - 64;
    int total = 0;
    for (int k = 0; k < kSubLists; ++k) total += word(&counters->frontier[st][k * kCountStride]);
    summary->frontier_total[st] = total;
  }
  if (threadIdx.x == 128) {
    summary->frontier_overflow = word(&counters->frontier_overflow);
    summary->leaf_overflow = word(&counters->leaf_overflow);
    unsigned gathers = word(&counters->wave_gathers);
    for (int k = 0; k < 16; ++k) gathers += word(&counters->gathers_shard[k * kCountStride]);
    summary->wave_gathers = gathers;
  }
  if (threadIdx.x >= 192 && threadIdx.x < 200) {
    unsigned total = 0;
    for (int k = 0; k < 16; ++k) total += word(&counters->queue_stats[k * kCountStride + (threadIdx.x - 192)]);
    summary->queue_stats[threadIdx.x - 192] = total;
  }
  __syncthreads();      // (states_out is complete: the selection below reads it, not `states`)
  states = states_out;
  int max_count;
  {
    const int lane = threadIdx.x & 63;
    max_count = WaveMax(min(word(&leaves.counts[lane * kCountStride]), leaves.sub_capacity));
  }
  const int total = max_count * kSubLists;
  auto leaf_at = [&](int i, Node2D* nd) {
    const int sub = i & (kSubLists - 1), j = i / kSubLists;
    if (j >= min(word(&leaves.counts[sub * kCountStride]), leaves.sub_capacity)) return false;
    const Node2D* at = &leaves.nodes[static_cast<size_t>(sub) * leaves.sub_capacity + j];
    if constexpr (kSameLaunch) {
      const unsigned long long* w = reinterpret_cast<const unsigned long long*>(at);
      unsigned long long v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = LoadAgent(&w[k]);
      __builtin_memcpy(nd, v, sizeof(Node2D));
    } else {
      *nd = *at;
    }
    return true;
  };
  // Common case (a handful of leaves, a few problems): every thread keeps its leaf in
  // registers and the four selection rounds run on LDS atomics -- one trip to memory
  // instead of five passes of global atomics and fences.
  constexpr int kFastProblems = 64;
  if (total <= static_cast<int>(blockDim.x) && num_problems <= kFastProblems) {
    __shared__ unsigned s_best_bits[kFastProblems], s_coarse[kFastProblems];
    __shared__ unsigned long long s_key[kFastProblems];
    __shared__ int s_ties[kFastProblems], s_scan[kFastProblems], s_dx[kFastProblems],
        s_dy[kFastProblems];
    if (static_cast<int>(threadIdx.x) < num_problems) {
      const int p = threadIdx.x;
      s_best_bits[p] = states[p].best_bits;
      s_coarse[p] = 0;
      s_key[p] = ~0ull;
      s_ties[p] = 0;
      s_scan[p] = -1; s_dx[p] = 0; s_dy[p] = 0;
    }
    Node2D nd;
    const bool have = leaf_at(threadIdx.x, &nd);
    const int p = have ? NodeProblem(nd) : 0;
    __syncthreads();
    const bool tied = have && __float_as_uint(nd.score) == s_best_bits[p];
    if (tied) {
      atomicMax(&s_coarse[p], __float_as_uint(nd.coarse_score));
      atomicAdd(&s_ties[p], 1);
    }
    __syncthreads();
    const unsigned long long key =
        (static_cast<unsigned long long>(static_cast<unsigned>(nd.coarse_index)) << 32) | nd.path;
    const bool top = tied && __float_as_uint(nd.coarse_score) == s_coarse[p];
    if (top) atomicMin(&s_key[p], key);
    __syncthreads();
    if (top && key == s_key[p]) {     // duplicates (dive + search) carry identical content
      s_scan[p] = nd.scan; s_dx[p] = nd.dx; s_dy[p] = nd.dy;
      BestLeaf b;
      b.score = nd.score; b.scan = nd.scan; b.dx = nd.dx; b.dy = nd.dy;
      b.found = 1; b.ties = 1; b.pad0 = b.pad1 = 0;
      best[p] = b;
    }
    __syncthreads();
    if (static_cast<int>(threadIdx.x) < num_problems) s_ties[threadIdx.x] = 0;
    __syncthreads();
    // ties = 1 + tied records that are a DIFFERENT leaf than the chosen one.
    if (tied && (nd.scan != s_scan[p] || nd.dx != s_dx[p] || nd.dy != s_dy[p]))
      atomicAdd(&s_ties[p], 1);
    __syncthreads();
    if (static_cast<int>(threadIdx.x) < num_problems) {
      const int q = threadIdx.x;
      sel[q].best_coarse_bits = s_coarse[q];
      sel[q].key = s_key[q];
      sel[q].ties = s_ties[q];
      if (s_scan[q] < 0) {
        BestLeaf b{};
        best[q] = b;
      } else {
        best[q].ties = 1 + s_ties[q];
      }
    }
    return;
  }
  for (int p = threadIdx.x; p < num_problems; p += blockDim.x) {
    sel[p].best_coarse_bits = 0;
    sel[p].ties = 0;
    sel[p].key = ~0ull;
    BestLeaf b{};
    best[p] = b;
  }
  __threadfence();
  __syncthreads();
  for (int i = threadIdx.x; i < total; i += blockDim.x) {
    Node2D nd;
    if (!leaf_at(i, &nd)) continue;
    const int p = NodeProblem(nd);
    if (__float_as_uint(nd.score) == states[p].best_bits) {
      atomicMax(&sel[p].best_coarse_bits, __float_as_uint(nd.coarse_score));
      atomicAdd(&sel[p].ties, 1);
    }
  }
  __threadfence();
  __syncthreads();
  for (int i = threadIdx.x; i < total; i += blockDim.x) {
    Node2D nd;
    if (!leaf_at(i, &nd)) continue;
    const int p = NodeProblem(nd);
    if (__float_as_uint(nd.score) == states[p].best_bits &&
        __float_as_uint(nd.coarse_score) ==
            __hip_atomic_load(&sel[p].best_coarse_bits, __ATOMIC_RELAXED,
                              __HIP_MEMORY_SCOPE_AGENT)) {
      const unsigned long long key =
          (static_cast<unsigned long long>(static_cast<unsigned>(nd.coarse_index)) << 32) | nd.path;
      atomicMin(&sel[p].key, key);
    }
  }
  __threadfence();
  __syncthreads();
  for (int i = threadIdx.x; i < total; i += blockDim.x) {
    Node2D nd;
    if (!leaf_at(i, &nd)) continue;
    const int p = NodeProblem(nd);
    const unsigned long long key =
        (static_cast<unsigned long long>(static_cast<unsigned>(nd.coarse_index)) << 32) | nd.path;
    if (__float_as_uint(nd.score) == states[p].best_bits &&
        __float_as_uint(nd.coarse_score) ==
            __hip_atomic_load(&sel[p].best_coarse_bits, __ATOMIC_RELAXED,
                              __HIP_MEMORY_SCOPE_AGENT) &&
        key == __hip_atomic_load(&sel[p].key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
      BestLeaf b;
      b.score = nd.score; b.scan = nd.scan; b.dx = nd.dx; b.dy = nd.dy;
      b.found = 1;
      b.ties = 1;
      b.pad0 = b.pad1 = 0;
      best[p] = b;   // duplicates (dive + search) carry identical content
    }
  }
  __threadfence();
  __syncthreads();
  // ties = 1 + number of tied records that are a DIFFERENT leaf (the dive and
  // the search record the best leaf twice; that is not a tie).
  for (int i = threadIdx.x; i < total; i += blockDim.x) {
    Node2D nd;
    if (!leaf_at(i, &nd)) continue;
    const int p = NodeProblem(nd);
    if (__float_as_uint(nd.score) != states[p].best_bits) continue;
    const int scan = __hip_atomic_load(&best[p].scan, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int dx = __hip_atomic_load(&best[p].dx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int dy = __hip_atomic_load(&best[p].dy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (nd.scan != scan || nd.dx != dx || nd.dy != dy) atomicAdd(&best[p].ties, 1);
  }
}

// The one block that selects also PUBLISHES: everything the host reads after a search (the
// counters' summary, selection states, best leaves, problem states: `tail_words` dwords behind
// the counters) goes from device memory straight into the caller's pinned buffer (mapped into
// the device's address space) -- no copy kernel behind this one in a chain of launches that is
// latency from end to end.  tail_host == nullptr: the host fetches the tail itself.
__global__ void __launch_bounds__(1024)
SelectBestKernel(NodeList leaves, const ProblemState* __restrict__ states,
                 SelectState* __restrict__ sel, BestLeaf* __restrict__ best, int num_problems,
                 ProblemState* __restrict__ states_out, const Counters* __restrict__ counters,
                 CountersSummary* __restrict__ summary, const unsigned* __restrict__ tail_dev,
                 unsigned* __restrict__ tail_host, int tail_words) {
  SelectBestBody<false>(leaves, states, sel, best, num_problems, states_out, counters, summary);
  if (tail_host == nullptr) return;
  __threadfence();
  __syncthreads();
  for (int i = threadIdx.x; i < tail_words; i += blockDim.x)
    tail_host[i] = __hip_atomic_load(&tail_dev[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------
// Branch and bound through a work queue (round 6): ONE launch behind the dive and the
// lowest-resolution filter instead of the wave / subtree / select chain of launches.
//
// A worker is a WAVEFRONT.  It takes a node from the queue and walks a CHAIN from it: expand
// (one quad gather per point, the node's scan held in registers: 16 cells per lane), continue
// with the best child, hand the other children that can still matter to the queue; a chain ends
// at a leaf (which raises the problem's bound) or when no child reaches the bound.  Every chain is
// a greedy dive, so the bound tightens as fast as the reference's depth-first search tightens it
// (SM2/fast_...2d.cc:335-378), while thousands of chains run at once.  The level-synchronous
// kernels this replaces expanded a whole level against the bound the dive had left: 93 000 node
// expansions on a hard scan where the reference's own order needs 40 000.
//
// The queue: kQueues sub-queues (a control word takes ~90 atomics per microsecond).  A slot is
// eight 8-byte granules {tag = the call's epoch, word of the node}, each written by ONE
// agent-scope store and polled with agent-scope loads: the data is its own flag, no fence
// (MI355X guide, inter-workgroup hand-off R2).  push: one atomic add on `reserved`, then the
// granules.  pop: look at every sub-queue (one wave-wide load), CAS `head` of one that has
// something -- never a ticket for a node that does not exist yet, so a worker NEVER waits for
// work: a wavefront that finds every sub-queue empty is done for good.  That is safe because
// whoever publishes a node looks again when its own chain has ended, and it is what makes eight
// such launches share the chip: no wavefront spins on another one's progress (only, briefly,
// on the granules of a slot that has been reserved and is being written).  The last workgroup to
// run out of work selects the best leaves and publishes the results (SelectBestBody).
// Slots are never reused within a call; a sub-queue that fills up sets frontier_overflow and
// the host repeats the search on the strict, chunked path below.
// ---------------------------------------------------------------------------
struct TreeQueue {
  unsigned long long* slots;     // [kQueues][capacity][8] granules
  int capacity;                  // nodes per sub-queue
  unsigned epoch;                // tag of this call (never 0; the buffer only holds older tags)
};

__device__ __forceinline__ unsigned long long* QueueSlot(const TreeQueue& Q, int q, int slot) {
  return Q.slots + (static_cast<size_t>(q) * Q.capacity + slot) * 8;
}

__device__ __forceinline__ void StoreGranule(unsigned long long* g, unsigned epoch, unsigned value) {
  __hip_atomic_store(g, (static_cast<unsigned long long>(epoch) << 32) | value, __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}

// One lane writes one node (the calling lanes hold different nodes).
__device__ __forceinline__ void StoreNodeGranules(const TreeQueue& Q, int q, int slot,
                                                  const Node2D& nd) {
  unsigned long long* g = QueueSlot(Q, q, slot);
  StoreGranule(g + 0, Q.epoch, static_cast<unsigned>(nd.problem));
  StoreGranule(g + 1, Q.epoch, static_cast<unsigned>(nd.scan));
  StoreGranule(g + 2, Q.epoch, static_cast<unsigned>(nd.dx));
  StoreGranule(g + 3, Q.epoch, static_cast<unsigned>(nd.dy));
  StoreGranule(g + 4, Q.epoch, __float_as_uint(nd.score));
  StoreGranule(g + 5, Q.epoch, static_cast<unsigned>(nd.coarse_index));
  StoreGranule(g + 6, Q.epoch, nd.path);
  StoreGranule(g + 7, Q.epoch, __float_as_uint(nd.coarse_score));
}

// Wave-wide push of the lanes with `keep` to sub-queue q: one reservation, slots in lane order.
// A full sub-queue raises the overflow flag (the reservation stands: poppers never look beyond
// `capacity`; slots below it are still filled, so every reserved slot below it IS published).
__device__ __forceinline__ void QueuePush(const TreeQueue& Q, Counters* __restrict__ counters,
                                          int q, bool keep, const Node2D& nd) {
  const int lane = threadIdx.x & 63;
  const unsigned long long mask = __ballot(keep);
  if (mask == 0) return;
  const int m = __popcll(mask);
  int first = 0;
  if (lane == 0)
    first = __hip_atomic_fetch_add(&counters->queue[q].reserved, m, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
  first = __builtin_amdgcn_readfirstlane(first);
  // (agent-scope stores: the selecting workgroup of this launch reads the flag)
  if (first + m > Q.capacity && lane == 0)
    __hip_atomic_store(&counters->frontier_overflow, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (keep) {
    const int slot = first + __popcll(mask & ((1ull << lane) - 1));
    if (slot < Q.capacity) StoreNodeGranules(Q, q, slot, nd);
  }
}

// Lanes 0..7 sweep the granules of a slot that has been taken -- it is reserved, i.e. written or
// being written -- until every tag is this call's; the node comes back wave-uniform.
__device__ __forceinline__ void ReadSlot(const TreeQueue& Q, int q, int slot, Node2D* out,
                                         unsigned* slot_rereads) {
  const int lane = threadIdx.x & 63;
  const unsigned long long* g = QueueSlot(Q, q, slot);
  unsigned word = 0;
  for (;;) {
    bool ok = true;
    if (lane < 8) {
      const unsigned long long x = LoadAgent(&g[lane]);
      word = static_cast<unsigned>(x);
      ok = static_cast<unsigned>(x >> 32) == Q.epoch;
    }
    if (__all(ok)) break;
    ++*slot_rereads;
    __builtin_amdgcn_s_sleep(2);
  }
  out->problem = static_cast<int>(__builtin_amdgcn_readlane(word, 0));
  out->scan = static_cast<int>(__builtin_amdgcn_readlane(word, 1));
  out->dx = static_cast<int>(__builtin_amdgcn_readlane(word, 2));
  out->dy = static_cast<int>(__builtin_amdgcn_readlane(word, 3));
  out->score = __uint_as_float(__builtin_amdgcn_readlane(word, 4));
  out->coarse_index = static_cast<int>(__builtin_amdgcn_readlane(word, 5));
  out->path = __builtin_amdgcn_readlane(word, 6);
  out->coarse_score = __uint_as_float(__builtin_amdgcn_readlane(word, 7));
}

__device__ __forceinline__ bool TakeSlot(Counters* __restrict__ counters, int q, int slot) {
  int got = 0;
  if ((threadIdx.x & 63) == 0) {
    int expected = slot;
    got = __hip_atomic_compare_exchange_strong(&counters->queue[q].head, &expected, slot + 1,
                                               __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT) ? 1 : 0;
  }
  return __builtin_amdgcn_readfirstlane(got) != 0;
}

// The wavefront's HOME sub-queue (the only one it pushes to).  false: it is empty -- and, no
// push of this wavefront being outstanding, it stays empty: the wavefront may leave.
__device__ __forceinline__ bool PopHome(const TreeQueue& Q, Counters* __restrict__ counters,
                                        int home, Node2D* out, unsigned* lost_races,
                                        unsigned* slot_rereads) {
  for (;;) {
    const unsigned long long hr =
        LoadAgent(reinterpret_cast<const unsigned long long*>(&counters->queue[home].head));
    const int head = static_cast<int>(hr & 0xffffffffu);
    const int reserved = min(static_cast<int>(hr >> 32), Q.capacity);
    if (reserved <= head) return false;
    if (TakeSlot(counters, home, head)) {
      ReadSlot(Q, home, head, out, slot_rereads);
      return true;
    }
    ++*lost_races;          // a thief was faster: look again (the sub-queue only loses nodes that way)
  }
}

// Somebody else's node: up to `windows` windows of 64 sub-queues from a start that differs per
// wavefront and attempt, one of the busy sub-queues of a window (not the first: every thief
// would race for the same word).  A thief that loses `max_lost` races, or finds its windows
// empty, gives up: supply is short then, and whoever published a node takes it in the end.
__device__ __forceinline__ bool Steal(const TreeQueue& Q, Counters* __restrict__ counters,
                                      int home, int queues, unsigned salt, int windows,
                                      int max_lost, Node2D* out, unsigned* lost_races,
                                      unsigned* slot_rereads) {
  const int lane = threadIdx.x & 63;
  int lost = 0;
  for (int w = 0; w < windows; ++w) {
    const int base = static_cast<int>((static_cast<unsigned>(home) * 2654435761u + salt * 40503u + w * 64u) %
                                      static_cast<unsigned>(queues));
    const int mine = (base + lane) % queues;
    const unsigned long long hr =
        LoadAgent(reinterpret_cast<const unsigned long long*>(&counters->queue[mine].head));
    const int head = static_cast<int>(hr & 0xffffffffu);
    const int reserved = min(static_cast<int>(hr >> 32), Q.capacity);
    const unsigned long long avail = __ballot(reserved > head);
    if (avail == 0) continue;
    int skip = static_cast<int>((salt + home) % static_cast<unsigned>(__popcll(avail)));
    unsigned long long rest = avail;
    while (skip-- > 0) rest &= rest - 1;
    const int l = __ffsll(static_cast<long long>(rest)) - 1;
    const int q = (base + l) % queues;
    const int slot = __builtin_amdgcn_readlane(head, l);
    if (TakeSlot(counters, q, slot)) {
      ReadSlot(Q, q, slot, out, slot_rereads);
      return true;
    }
    ++*lost_races;
    if (++lost >= max_lost) return false;
    --w;                    // the same window again (another sub-queue of it: salt)
    ++salt;
  }
  return false;
}

constexpr int kChainCells = 16;     // cells per lane held in registers: clouds of up to 1024 points
constexpr uint32_t kNoCell = 0x80008000u;   // (x, y) = (-32768, -32768): outside every level

// One leaf record, stored so that the selecting workgroup of the SAME launch reads it.
__device__ __forceinline__ void RecordLeafAgent(const Node2D& leaf, const NodeList& leaves,
                                                int sub, Counters* __restrict__ counters) {
  const int slot = ListReserve(leaves, sub, 1);
  if (slot >= leaves.sub_capacity) {
    __hip_atomic_store(&counters->leaf_overflow, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  unsigned long long v[4];
  __builtin_memcpy(v, &leaf, sizeof(Node2D));
  unsigned long long* at = reinterpret_cast<unsigned long long*>(
      &leaves.nodes[static_cast<size_t>(sub) * leaves.sub_capacity + slot]);
#pragma unroll
  for (int k = 0; k < 4; ++k)
    __hip_atomic_store(&at[k], v[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void __launch_bounds__(256)
TreeQueueKernel(const Fast2DProblem* __restrict__ problems, ProblemState* __restrict__ states, int n,
                NodeList in, TreeQueue Q, NodeList leaves, Counters* __restrict__ counters,
                SelectState* __restrict__ sel, BestLeaf* __restrict__ best_out, int num_problems,
                ProblemState* __restrict__ states_out, CountersSummary* __restrict__ summary,
                const unsigned* __restrict__ tail_dev, unsigned* __restrict__ tail_host,
                int tail_words, int counters_trace, int max_lost) {
  static_assert(sizeof(Node2D) == 32, "a node is eight words: eight granules, four leaf stores");
  const int lane = threadIdx.x & 63;
  const int wave_id = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int num_waves = gridDim.x * 4;
  const int queues = min(num_waves, kQueues);          // sub-queues in use
  const int home = wave_id % queues;
  unsigned steals = 0;
  unsigned long long scored = 0, expanded = 0;
  unsigned gathers = 0;
  unsigned q_pops = 0, q_lost = 0, q_rereads = 0, q_pushed = 0, q_listed = 0, q_chains = 0;
  int stat_problem = -1;
  const auto flush_stats = [&]() {
    if (lane == 0 && stat_problem >= 0 && expanded) {
      ProblemState& st = states[stat_problem];
      atomicAdd(&st.scored_shard[wave_id & (kStatShards - 1)], scored);
      atomicAdd(&st.expanded_shard[wave_id & (kStatShards - 1)], expanded);
    }
    scored = expanded = 0;
  };
  // Phase 1: this wavefront's share of the filter's list (written by the launch before: plain
  // loads), node i of the interleaved sub-lists for i = wave, wave + W, ...  Phase 2: the queue.
  // A static share is safe here because nobody ever waits for anybody: a workgroup that is not
  // resident yet simply walks its share when it gets there.
  // BEST FIRST within the share: 64 nodes of it at a time, one per lane (one wave-wide load), the
  // highest score among them taken next -- and the round dropped as soon as that score is below
  // the bound.  Every wavefront starts with the best node it owns, so the first thousand chains
  // are dives from the best thousand nodes of the list and the bound is close to its final value
  // when they end; the rest of the list then dies at a compare.  (In list order -- the filter's,
  // i.e. by rotation -- a hard scan expanded 23 000 nodes where 2 400 suffice with the final
  // bound known from the start: profiles/r06_queue_development.txt.)
  const int in_slots = ListMaxCount(in) * kSubLists;
  int round = 0;
  Node2D mine{};               // this lane's node of the current round
  unsigned mine_bits = 0;      // its score's bits; 0: none (taken, or no such node)
  bool round_loaded = false, share_done = false;
  bool pushed_any = false;
  Node2D nd;
  for (;;) {
    bool have = false;
    while (!have && !share_done) {
      if (!round_loaded) {
        // (wave-uniform: the first lane's index is the smallest of the round)
        if (wave_id + static_cast<long long>(round) * 64 * num_waves >= in_slots) { share_done = true; break; }
        const long long i = wave_id + (static_cast<long long>(round) * 64 + lane) * num_waves;
        ++round;
        mine_bits = 0;
        if (i < in_slots) {
          const int in_sub = static_cast<int>(i & (kSubLists - 1)), j = static_cast<int>(i / kSubLists);
          if (j < min(in.counts[in_sub * kCountStride], in.sub_capacity)) {
            mine = in.nodes[static_cast<size_t>(in_sub) * in.sub_capacity + j];
            mine_bits = max(__float_as_uint(fmaxf(mine.score, 0.f)), 1u);
          }
        }
        round_loaded = true;
      }
      const unsigned top = static_cast<unsigned>(WaveMax(static_cast<int>(mine_bits)));
      const float bound_now = __uint_as_float(LoadAgent(&states[0].best_bits));
      // (the bound of problem 0 only prunes rounds of a single search; batches check per node)
      if (top == 0 || (num_problems == 1 && __uint_as_float(top) < bound_now)) {
        round_loaded = false;      // nothing left in this round that can matter
        continue;
      }
      const int l = __ffsll(static_cast<long long>(__ballot(mine_bits == top))) - 1;
      nd.problem = __builtin_amdgcn_readlane(mine.problem, l);
      nd.scan = __builtin_amdgcn_readlane(mine.scan, l);
      nd.dx = __builtin_amdgcn_readlane(mine.dx, l);
      nd.dy = __builtin_amdgcn_readlane(mine.dy, l);
      nd.score = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(mine.score), l));
      nd.coarse_index = __builtin_amdgcn_readlane(mine.coarse_index, l);
      nd.path = __builtin_amdgcn_readlane(mine.path, l);
      nd.coarse_score =
          __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(mine.coarse_score), l));
      if (lane == l) mine_bits = 0;
      have = true;
      ++q_listed;
    }
    if (!have) {
      // its own sub-queue first (whoever publishes nodes is who guarantees that they are taken:
      // by a thief, or in the end by itself), then somebody else's; nothing there, or the races
      // lost: this wavefront is done -- its own sub-queue is empty and stays so
      if (!(pushed_any && PopHome(Q, counters, home, &nd, &q_lost, &q_rereads)) &&
          !Steal(Q, counters, home, queues, ++steals, 2, max_lost, &nd, &q_lost, &q_rereads))
        break;
      ++q_pops;
    }
    const int problem = __builtin_amdgcn_readfirstlane(NodeProblem(nd));
    const Fast2DProblem& P = problems[problem];
    ProblemState& st = states[problem];
    float best = __uint_as_float(LoadAgent(&st.best_bits));
    if (nd.score < best) continue;                       // the bound has risen since the push
    if (problem != stat_problem) { flush_stats(); stat_problem = problem; }
    ++q_chains;
    // the node's scan: 16 cells per lane, kept for the whole chain
    const auto* pts = AsGlobal(P.discrete) + static_cast<size_t>(nd.scan) * n;
    uint32_t cell[kChainCells];
#pragma unroll
    for (int j = 0; j < kChainCells; ++j) {
      const int i = j * kWave + lane;
      cell[j] = kNoCell;
      if (j * kWave < n) cell[j] = i < n ? pts[i] : kNoCell;
    }
    const int4 bd = P.bounds[nd.scan];
    const float min_score = P.min_score;
    for (;;) {                                           // the chain
      // (everything about the node is wavefront-uniform, and the compiler is TOLD so: a buffer
      // resource it cannot prove uniform gets every one of the sixteen gathers wrapped in a
      // readfirstlane loop of thirty instructions)
      const int child_level = __builtin_amdgcn_readfirstlane(NodeLevel(nd) - 1);
      const LevelDesc& Lm = P.level[child_level];
      const int half = 1 << child_level;
      const int ndx = __builtin_amdgcn_readfirstlane(nd.dx), ndy = __builtin_amdgcn_readfirstlane(nd.dy);
      const bool vx = ndx + half <= bd.y, vy = ndy + half <= bd.w;
      const int ax = ndx + 2 * half - 1, ay = ndy + 2 * half - 1;
      const int parent_ub = SumUpperBound(P, nd.score, n);
      struct { int qx, qy, qtx; } L;
      L.qx = __builtin_amdgcn_readfirstlane(Lm.qx);
      L.qy = __builtin_amdgcn_readfirstlane(Lm.qy);
      L.qtx = __builtin_amdgcn_readfirstlane(Lm.qtx);
      const unsigned long long quads_address = reinterpret_cast<unsigned long long>(Lm.quads);
      const unsigned long long quads_uniform =
          static_cast<unsigned long long>(static_cast<unsigned>(
              __builtin_amdgcn_readfirstlane(static_cast<unsigned>(quads_address)))) |
          (static_cast<unsigned long long>(static_cast<unsigned>(
               __builtin_amdgcn_readfirstlane(static_cast<unsigned>(quads_address >> 32)))) << 32);
      const unsigned long long quad_bytes =
          static_cast<unsigned long long>((L.qy + 3) >> 2) * static_cast<unsigned>(L.qtx) * 128ull;
      // (levels beyond the 2 GB a buffer resource addresses do not come here: see queue_ok)
      const __amdgpu_buffer_rsrc_t quad_rsrc = __builtin_amdgcn_make_buffer_rsrc(
          reinterpret_cast<uint32_t*>(quads_uniform), 0, static_cast<int>(quad_bytes), 0x00020000);
      // packed 16-bit sums: (child 00 | child 10 << 16) and (child 01 | child 11 << 16); a lane adds
      // at most 16 x 255.  Children beyond the search bounds (`break`s at :356,361) are summed
      // like the others and dropped when the scores are formed: the level-(l+1) cell is the
      // maximum of all four level-l cells whatever the bounds say, so the early-exit bound below
      // holds with them in it.
      typedef unsigned short Halves __attribute__((ext_vector_type(2)));
      uint32_t even = 0, odd = 0;
      int seen_max = 0;
      bool dead = false;
      // The first 256 points, then -- unless no child can reach the bound any more -- ALL the
      // others in flight at once: two trips to memory per expansion at most (a check after every
      // 256 points was four).
      const auto gather = [&](auto first_tag, auto count_tag) {
        constexpr int kFirst = decltype(first_tag)::value, kCount = decltype(count_tag)::value;
        uint32_t v[kCount];
#pragma unroll
        for (int u = 0; u < kCount; ++u) {
          const uint32_t p = cell[kFirst + u];
          const unsigned X = static_cast<unsigned>(static_cast<short>(p & 0xffffu) + ax);
          const unsigned Y = static_cast<unsigned>(static_cast<short>(p >> 16) + ay);
          const bool inside = X < static_cast<unsigned>(L.qx) && Y < static_cast<unsigned>(L.qy);
          // QuadOffset(X, Y, qtx) * 4, branch-free, the tile index by a 24-bit multiply-add
          // (inside: Y >> 2 and qtx are far below 2^24)
          const unsigned tile = __umul24(Y >> 2, static_cast<unsigned>(L.qtx)) + (X >> 3);
          const unsigned byte = (tile << 7) | ((Y & 3u) << 5) | ((X & 7u) << 2);
          v[u] = __builtin_amdgcn_raw_buffer_load_b32(quad_rsrc, inside ? byte : 0xfffffff0u, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < kCount; ++u) {
          const uint32_t e = v[u] & 0x00ff00ffu, o = (v[u] >> 8) & 0x00ff00ffu;
          even += e;
          odd += o;
          Halves eh, oh;
          __builtin_memcpy(&eh, &e, 4);
          __builtin_memcpy(&oh, &o, 4);
          const Halves m = __builtin_elementwise_max(eh, oh);
          seen_max += static_cast<int>(max(m.x, m.y));
        }
      };
      gather(std::integral_constant<int, 0>{}, std::integral_constant<int, 4>{});
      gathers += 4;
      if (n > 4 * kWave) {
        // what the best child of this lane's points still lacks to the parent (see
        // ExpandWaveKernel): no child can reach the bound -> the rest of the gathers is skipped
        const int most = static_cast<int>(max(max(even & 0xffffu, even >> 16),
                                              max(odd & 0xffffu, odd >> 16)));
        if (ToScore(P, parent_ub - WaveSum(seen_max - most), n) < best) {
          dead = true;
        } else {
          if (n > 8 * kWave) {
            gather(std::integral_constant<int, 4>{}, std::integral_constant<int, 12>{});
            gathers += 12;
          } else {
            gather(std::integral_constant<int, 4>{}, std::integral_constant<int, 4>{});
            gathers += 4;
          }
        }
      }
      ++expanded;
      scored += (1 + (vx ? 1 : 0)) * (1 + (vy ? 1 : 0));
      if (dead) break;
      const int total[4] = {WaveSum(static_cast<int>(even & 0xffffu)), WaveSum(static_cast<int>(odd & 0xffffu)),
                            WaveSum(static_cast<int>(even >> 16)), WaveSum(static_cast<int>(odd >> 16))};
      // lane k < 4 is child k = 2 * x-step + y-step (generation order: x outer, y inner)
      float sc[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const bool valid = ((k >> 1) == 0 || vx) && ((k & 1) == 0 || vy);
        sc[k] = valid ? ToScore(P, total[k], n) : -1.f;
      }
      const int k = lane & 3;
      const float mine = k == 0 ? sc[0] : (k == 1 ? sc[1] : (k == 2 ? sc[2] : sc[3]));
      int rank = 0;
#pragma unroll
      for (int o = 0; o < 4; ++o)
        if (o != k && sc[o] >= 0.f && (sc[o] > mine || (sc[o] == mine && o < k))) ++rank;
      best = fmaxf(best, __uint_as_float(LoadAgent(&st.best_bits)));
      const bool keep = lane < 4 && mine >= 0.f && mine >= best;
      // the best valid child (rank 0; child 0 is always valid)
      const int kbest = __ffsll(static_cast<long long>(__ballot(lane < 4 && mine >= 0.f && rank == 0))) - 1;
      Node2D child;
      child.problem = problem | (child_level << 24);
      child.scan = nd.scan;
      child.dx = nd.dx + (k >> 1) * half;
      child.dy = nd.dy + (k & 1) * half;
      child.score = mine;
      child.coarse_index = nd.coarse_index;
      child.path = nd.path | (static_cast<unsigned>(rank) << (2 * child_level));
      child.coarse_score = nd.coarse_score;
      const bool best_kept = (__ballot(keep) >> kbest) & 1ull;
      if (child_level == 0) {
        // leaves: only the first-best child can be returned by the reference (:340-343 after the
        // stable sort of :331-332)
        if (best_kept && lane == kbest && child.score > min_score) {
          RecordLeafAgent(child, leaves, wave_id & (kSubLists - 1), counters);
          atomicMax(&st.best_bits, __float_as_uint(child.score));
        }
        break;
      }
      if (!best_kept) break;                              // the best child is below the bound: all are
      // the other children that can still matter go to the queue ...
      const bool others = keep && lane != kbest;
      if (__ballot(others)) {
        QueuePush(Q, counters, home, others, child);
        pushed_any = true;
        q_pushed += __popcll(__ballot(others));
      }
      // ... and the chain goes on with the best one
      nd.problem = child.problem;
      nd.dx = __builtin_amdgcn_readlane(child.dx, kbest);
      nd.dy = __builtin_amdgcn_readlane(child.dy, kbest);
      nd.score = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(child.score), kbest));
      nd.path = __builtin_amdgcn_readlane(child.path, kbest);
    }
  }
  // ---- out of work.  The work counters of the four wavefronts go out as one set of atomics per
  // workgroup (one per WAVEFRONT on a word per problem took 20 us of a 2048-wavefront launch: a
  // word takes ~90 atomics per microsecond), then the last workgroup to get here selects and
  // publishes. ---------------------------------------------------------------------------------
  __shared__ unsigned long long s_scored[4], s_expanded[4];
  __shared__ unsigned s_gathers[4];
  __shared__ unsigned s_queue_stats[8];
  __shared__ int s_problem[4];
  if (threadIdx.x < 8) s_queue_stats[threadIdx.x] = 0;
  __syncthreads();
  if (lane == 0 && counters_trace) {
    atomicAdd(&s_queue_stats[0], q_pops);
    atomicAdd(&s_queue_stats[1], q_lost);
    atomicAdd(&s_queue_stats[2], q_rereads);
    atomicAdd(&s_queue_stats[3], q_pushed);
    atomicAdd(&s_queue_stats[4], q_listed);
    atomicAdd(&s_queue_stats[5], q_chains);
  }
  __shared__ int s_last;
  if (lane == 0) {
    const int w = threadIdx.x >> 6;
    s_scored[w] = scored;
    s_expanded[w] = expanded;
    s_gathers[w] = gathers;
    s_problem[w] = expanded ? stat_problem : -1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wavefront's stores have landed
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned g = 0;
    for (int w = 0; w < 4; ++w) {
      g += s_gathers[w];
      if (s_problem[w] < 0) continue;
      unsigned long long sc = s_scored[w], ex = s_expanded[w];
      for (int v = w + 1; v < 4; ++v)
        if (s_problem[v] == s_problem[w]) { sc += s_scored[v]; ex += s_expanded[v]; s_problem[v] = -1; }
      ProblemState& st = states[s_problem[w]];
      atomicAdd(&st.scored_shard[blockIdx.x & (kStatShards - 1)], sc);
      atomicAdd(&st.expanded_shard[blockIdx.x & (kStatShards - 1)], ex);
    }
    if (g) atomicAdd(&counters->gathers_shard[(blockIdx.x & 15) * kCountStride], g);
    if (counters_trace)
      for (int k = 0; k < 6; ++k)
        if (s_queue_stats[k])
          atomicAdd(&counters->queue_stats[(blockIdx.x & 15) * kCountStride + k], s_queue_stats[k]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    s_last = __hip_atomic_fetch_add(&counters->blocks_done, 1, __ATOMIC_RELAXED,
                                    __HIP_MEMORY_SCOPE_AGENT) == static_cast<int>(gridDim.x) - 1;
  }
  __syncthreads();
  if (!s_last) return;
  SelectBestBody<true>(leaves, states, sel, best_out, num_problems, states_out, counters, summary);
  if (tail_host == nullptr) return;
  __threadfence();
  __syncthreads();
  for (int i = threadIdx.x; i < tail_words; i += blockDim.x)
    tail_host[i] = __hip_atomic_load(&tail_dev[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// depth == 1: the lowest-resolution candidates are the leaves
// (BranchAndBound returns candidates[0], SM2/fast_...2d.cc:340-343).
__global__ void __launch_bounds__(1024)
SelectDepthOneKernel(const Fast2DProblem* __restrict__ problems,
                     const ProblemState* __restrict__ states, int n, BestLeaf* __restrict__ best,
                     ProblemState* __restrict__ states_out) {
  const int problem = blockIdx.x;
  const Fast2DProblem& P = problems[problem];
  __shared__ unsigned long long keys[16];
  __shared__ int s_total;
  if (threadIdx.x == 0) s_total = 0;
  __syncthreads();
  unsigned long long key = 0;
  int total = 0;
  for (int s = threadIdx.x; s < P.num_scans; s += blockDim.x) {
    total += P.coarse_dims[s].x * P.coarse_dims[s].y;
    const int2 b = P.scan_best[s];
    // larger sum first, then smaller scan index (generation order)
    const unsigned long long k = (static_cast<unsigned long long>(static_cast<unsigned>(b.x)) << 32) |
                                 static_cast<unsigned>(0x7fffffff - s);
    key = k > key ? k : key;
  }
  key = WaveMaxU64(key);
  total = WaveSum(total);
  if ((threadIdx.x & 63) == 0) {
    keys[threadIdx.x >> 6] = key;
    if (total) atomicAdd(&s_total, total);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    ProblemState st = states[problem];
    st.coarse_total = s_total;
    states_out[problem] = st;
    for (int w = 1; w < 16; ++w) key = keys[w] > key ? keys[w] : key;
    BestLeaf b{};
    if (!states[problem].error && P.num_scans > 0) {
      const int s = 0x7fffffff - static_cast<int>(key & 0xffffffffu);
      const int2 sb = P.scan_best[s];
      const float score = ToScore(P, sb.x, n);
      if (score > P.min_score) {
        const int2 dims = P.coarse_dims[s];
        const int4 bd = P.bounds[s];
        b.found = 1; b.score = score; b.scan = s;
        b.dx = bd.x + sb.y / dims.y;
        b.dy = bd.z + sb.y % dims.y;
        b.ties = 1;
      }
    }
    best[problem] = b;
  }
}

}  // namespace

// ---------------------------------------------------------------------------
// Fast2DMatcher (host)
// ---------------------------------------------------------------------------
Fast2DMatcher::Fast2DMatcher(const cmx_fast2d_options& options, const cmx_grid2d_limits& limits,
                             const uint16_t* cells, int device)
    : options_(options), limits_(limits), device_(device) {
  // CHECKs of the reference: SM2/fast_...2d.cc:100-102,174; map_limits.h:45-47;
  // grid_2d.cc:73.
  CMX_REQUIRE(cells != nullptr, "cells is null");
  CMX_REQUIRE(options.branch_and_bound_depth >= 1 && options.branch_and_bound_depth <= kMaxDepth,
              "branch_and_bound_depth %d outside [1,%d]", options.branch_and_bound_depth,
              kMaxDepth);
  CMX_REQUIRE(limits.resolution > 0., "resolution must be > 0");
  CMX_REQUIRE(limits.num_x_cells >= 1 && limits.num_y_cells >= 1, "empty cell limits");
  CMX_REQUIRE(limits.num_x_cells <= 16384 && limits.num_y_cells <= 16384,
              "grid larger than 16384 cells per side is unsupported");
  CMX_REQUIRE(limits.min_correspondence_cost < limits.max_correspondence_cost,
              "min_correspondence_cost must be < max_correspondence_cost");
  WorkspaceLease ws(device);
  const int nx = limits.num_x_cells, ny = limits.num_y_cells;
  const int depth = options.branch_and_bound_depth;
  size_t total = 0;
  level_offsets_.resize(depth);
  levels_.resize(depth);
  for (int i = 0; i < depth; ++i) {
    const int w = 1 << i;
    level_offsets_[i] = total;
    levels_[i].wx = nx + w - 1;
    levels_[i].wy = ny + w - 1;
    total += (static_cast<size_t>(levels_[i].wx) * levels_[i].wy + 255) & ~size_t(255);
  }
  CMX_HIP(hipMalloc(&stack_mem_, total));
  for (int i = 0; i < depth; ++i)
    levels_[i].cells = static_cast<uint8_t*>(stack_mem_) + level_offsets_[i];
  min_s_ = 1.f - limits.max_correspondence_cost;
  const float max_s = 1.f - limits.min_correspondence_cost;
  score_scale_ = (max_s - min_s_) / 255.f;

  const size_t count = static_cast<size_t>(nx) * ny;
  CMX_HIP(hipMalloc(reinterpret_cast<void**>(&grid_cells_), count * sizeof(uint16_t)));
  uint16_t* d_cells = grid_cells_;
  CMX_HIP(hipMemcpyAsync(d_cells, cells, count * sizeof(uint16_t), hipMemcpyHostToDevice,
                         ws->stream));
  BuildLevel0Kernel<<<DivUp(count, 256), 256, 0, ws->stream>>>(
      d_cells, static_cast<int>(count), limits.min_correspondence_cost,
      limits.max_correspondence_cost, const_cast<uint8_t*>(levels_[0].cells));
  for (int i = 1; i < depth; ++i) {
    const LevelDesc& prev = levels_[i - 1];
    const LevelDesc& cur = levels_[i];
    BuildLevelKernel<<<dim3(DivUp(cur.wx, 256), cur.wy), 256, 0, ws->stream>>>(
        prev.cells, prev.wx, prev.wy, 1 << (i - 1), const_cast<uint8_t*>(cur.cells), cur.wx,
        cur.wy);
  }
  // Quad layouts of every level that can be a child level (0 .. depth-2).
  {
    size_t quad_total = 0;
    std::vector<size_t> quad_off(depth, 0);
    for (int i = 0; i + 1 < depth; ++i) {
      const int w = 1 << i;
      levels_[i].qx = levels_[i].wx + w;
      levels_[i].qy = levels_[i].wy + w;
      levels_[i].qtx = (levels_[i].qx + 7) / 8;
      quad_off[i] = quad_total;
      // whole tiles of 32 dwords (128 bytes)
      quad_total += static_cast<size_t>(levels_[i].qtx) * ((levels_[i].qy + 3) / 4) * 128;
    }
    levels_[depth - 1].quads = nullptr;
    levels_[depth - 1].qx = levels_[depth - 1].qy = levels_[depth - 1].qtx = 0;
    if (quad_total) {
      CMX_HIP(hipMalloc(&quads_mem_, quad_total));
      for (int i = 0; i + 1 < depth; ++i) {
        LevelDesc& L = levels_[i];
        uint32_t* q = reinterpret_cast<uint32_t*>(static_cast<char*>(quads_mem_) + quad_off[i]);
        L.quads = q;
        BuildQuadsKernel<<<dim3(DivUp(L.qx, 256), L.qy), 256, 0, ws->stream>>>(
            L.cells, L.wx, L.wy, 1 << i, q, L.qx, L.qy, L.qtx);
      }
    }
  }
  // Phase planes of the lowest-resolution level.
  {
    const int w = 1 << (depth - 1);
    const LevelDesc& top = levels_[depth - 1];
    const int PI = (top.wx + w - 1) / w, PJ = (top.wy + w - 1) / w;
    if (w <= kMaxPlaneWidth && PI * PJ <= kMaxPlaneCells) {
      plane_i_ = PI;
      plane_j_ = PJ;
      plane_stride_ = (PI * PJ + 63) & ~63;
      const size_t bytes = static_cast<size_t>(w * w + 1) * plane_stride_;
      CMX_HIP(hipMalloc(reinterpret_cast<void**>(&planes_), bytes));
      BuildPlanesKernel<<<w * w + 1, 64, 0, ws->stream>>>(top.cells, top.wx, top.wy, w, PI, PJ,
                                                          plane_stride_, planes_);
      // The same planes of the level dilated by two cells (group bounds of the fused front end),
      // where the dilated image still fits the planes' PI x PJ lattice cells.
      const int dwx = top.wx + 2 * kGroupDilation, dwy = top.wy + 2 * kGroupDilation;
      if (plane_stride_ == 64 && depth > 1 && dwx <= PI * w && dwy <= PJ * w) {
        uint8_t* dilated = ws->dev[0].ReserveAs<uint8_t>(static_cast<size_t>(dwx) * dwy);
        DilateLevelKernel<<<dim3(DivUp(dwx, 256), dwy), 256, 0, ws->stream>>>(top.cells, top.wx,
                                                                             top.wy, dilated);
        CMX_HIP(hipMalloc(reinterpret_cast<void**>(&planes_group_), bytes));
        BuildPlanesKernel<<<w * w + 1, 64, 0, ws->stream>>>(dilated, dwx, dwy, w, PI, PJ,
                                                            plane_stride_, planes_group_);
      }
    }
  }
  CMX_HIP(hipGetLastError());
  CMX_HIP(hipStreamSynchronize(ws->stream));
}

Fast2DMatcher::~Fast2DMatcher() {
  (void)hipSetDevice(device_);
  if (stack_mem_) (void)hipFree(stack_mem_);
  if (quads_mem_) (void)hipFree(quads_mem_);
  if (planes_) (void)hipFree(planes_);
  if (planes_group_) (void)hipFree(planes_group_);
  if (grid_cells_) (void)hipFree(grid_cells_);
}

void FillRotationTable(double step, int num_angular, float2* out) {
  // delta_theta accumulates in f64, each angle is narrowed to f32 for AngleAxisf.
  const int num_scans = 2 * num_angular + 1;
  double delta_theta = -num_angular * step;
  for (int s = 0; s < num_scans; ++s, delta_theta += step) {
    const float ha = 0.5f * static_cast<float>(delta_theta);
    out[s] = make_float2(std::cos(ha), std::sin(ha) * 1.f);
  }
}

std::shared_ptr<const std::vector<float2>> HostRotationTable(double step, int num_angular) {
  struct Entry {
    double step;
    int num_angular;
    std::shared_ptr<const std::vector<float2>> table;
  };
  static std::mutex mu;
  static std::vector<Entry>* cache = new std::vector<Entry>;   // most recent last
  {
    std::lock_guard<std::mutex> lock(mu);
    for (size_t i = cache->size(); i-- > 0;) {
      if ((*cache)[i].step == step && (*cache)[i].num_angular == num_angular)
        return (*cache)[i].table;
    }
  }
  const int num_scans = 2 * num_angular + 1;
  auto table = std::make_shared<std::vector<float2>>(num_scans);
  FillRotationTable(step, num_angular, table->data());
  std::lock_guard<std::mutex> lock(mu);
  if (cache->size() >= 64) cache->erase(cache->begin());      // bound the cache
  cache->push_back(Entry{step, num_angular, table});
  return table;
}

namespace {

// SearchParameters ctor (SM2/correlative_scan_matcher_2d.cc:27-55), host side.
struct HostSearch {
  int num_angular;
  double step;
  int num_scans;
  int nl;
};
HostSearch MakeSearch(double linear_window, double angular_window, float max_range_xy,
                      double resolution) {
  float max_scan_range = 3.f * resolution;
  max_scan_range = std::max(max_range_xy, max_scan_range);
  const double kSafetyMargin = 1. - 1e-3;
  const float range_sq = max_scan_range * (max_scan_range * 1.f);
  const double res_sq = resolution * (resolution * 1.);
  HostSearch h;
  h.step = kSafetyMargin * std::acos(1. - res_sq / (2. * range_sq));
  h.num_angular = std::ceil(angular_window / h.step);
  h.num_scans = 2 * h.num_angular + 1;
  h.nl = std::ceil(linear_window / resolution);
  return h;
}

float MaxRangeXY(const float* xyz, int n) {
  float m = 0.f;
  for (int i = 0; i < n; ++i) {
    const float x = xyz[3 * i], y = xyz[3 * i + 1];
    m = std::max(m, std::sqrt(x * x + y * y));
  }
  return m;
}

struct PreparedBatch {
  StageTrace* trace = nullptr;
  int num_problems = 0;
  int n = 0;
  int max_scans = 0;
  long long plane_acc_cells = 0;   // LDS accumulators the plane kernel needs (upper bound)
  std::vector<HostSearch> search;
  std::vector<cmx_pose2d> initial;
  Fast2DProblem* d_problems = nullptr;
  ProblemState* d_states = nullptr;
  std::vector<Fast2DProblem> h_problems;
  // Search scratch carved before the first kernel so that it can clear the counters.
  char* d_misc = nullptr;          // Counters | SelectState[num] | BestLeaf[num] | ProblemState[num]
  bool write_all_discrete = false; // debug entry point: keep every discretised scan
  unsigned long long* d_timeline = nullptr;   // CMX_TIMELINE=1
  int timeline_blocks = 0;
  // The fused front end's launch, kept for the exact re-run of a problem whose leaves tie
  // (RescoreExact): under group bounds the lowest-resolution scores are bounds.
  bool any_group = false;
  size_t fused_lds = 0;
  int fused_acc = 0, fused_threads = 0;
  const float* d_xyz = nullptr;
};

// The debug switch fast2d_unfused routes every problem through the separate prep / score
// launches (the fallback of problems the fused kernel does not take); parity tests run both.
bool FusedEnabled() { return Debug().fast2d_unfused == 0; }

// Blocks of PrepScoreFusedKernel the whole chip holds at once (occupancy query, cached).
long long FusedResidentBlocks(int device, int threads, size_t lds_bytes) {
  struct Key { int device, threads; size_t lds; long long blocks; };
  static std::mutex mu;
  static std::vector<Key>* cache = new std::vector<Key>;
  const size_t lds = (lds_bytes + 1023) & ~size_t(1023);
  {
    std::lock_guard<std::mutex> lock(mu);
    for (const Key& k : *cache)
      if (k.device == device && k.threads == threads && k.lds == lds) return k.blocks;
  }
  int per_cu = 0, cus = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, PrepScoreFusedKernel<false>, threads, lds) !=
          hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  // (one fewer per CU than the API says: it over-reports by one for some SGPR counts,
  // MI355X_MICROARCH.md "Residency and cooperative launch")
  const long long blocks = static_cast<long long>(std::max(per_cu - 1, 0)) * cus;
  std::lock_guard<std::mutex> lock(mu);
  if (cache->size() < 256) cache->push_back(Key{device, threads, lds, blocks});
  return blocks;
}

// Uploads problem descriptors, carves scratch and runs the preparation +
// lowest-resolution scoring kernels.  `d_xyz` is the device point cloud.
// The work-queue tree search (TreeQueueKernel) holds a scan in registers, 16 cells per lane.
// Debug switch fast2d_queue: 2 = the chain of level-synchronous launches of rounds 2 - 5 (the
// parity partner, and the path a queue overflow falls back to).
// Batches of four or more problems stay on the level-synchronous launches: wide frontiers of many
// problems keep the chip busy there, and they measure faster (16 submaps: 1.4 against 2.7 ms).
bool QueueSearchWanted(int n, int num) {
  return Debug().fast2d_queue != 2 && n <= kChainCells * kWave && (num < 4 || Debug().fast2d_queue == 1);
}

void PrepareAndScoreCoarse(Workspace& ws, const Fast2DMatcher* const* matchers, int num,
                           const cmx_pose2d* initial_or_null, bool full_submap,
                           const float* d_xyz, int n, float max_range_xy, float min_score,
                           PreparedBatch* out, const int32_t* full_flags = nullptr,
                           const float* min_scores = nullptr) {
  // Mixed batches (the ConstraintBuilder front): per-problem full-submap flag and
  // acceptance threshold override the uniform ones.
  const auto is_full = [&](int p) { return full_flags ? full_flags[p] != 0 : full_submap; };
  const auto min_of = [&](int p) { return min_scores ? min_scores[p] : min_score; };
  out->num_problems = num;
  out->n = n;
  out->search.resize(num);
  out->initial.resize(num);
  out->h_problems.resize(num);

  // Per-problem search parameters and scratch sizes.
  size_t discrete_total = 0, scans_total = 0, coarse_total = 0;
  // Rotation tables (host libm values, cached process-wide) of the distinct
  // (step, num_angular) pairs of this batch; they travel in the problem upload.
  struct Rotation { double step; int num_angular; std::shared_ptr<const std::vector<float2>> table; size_t offset; };
  std::vector<Rotation> rotations;
  std::vector<int> rotation_of(num);
  size_t rotation_floats = 0;
  const bool fused_enabled = FusedEnabled();
  const int n_pad = (n + 63) & ~63;
  long long fused_acc = 0;
  bool any_fused = false, any_unfused = false;
  for (int p = 0; p < num; ++p) {
    const Fast2DMatcher& m = *matchers[p];
    const cmx_grid2d_limits& lim = m.limits();
    HostSearch h;
    cmx_pose2d init;
    if (is_full(p)) {
      // SM2/fast_...2d.cc:213-222.
      h = MakeSearch(1e6 * lim.resolution, M_PI, max_range_xy, lim.resolution);
      init.x = lim.max_x - 0.5 * lim.resolution * lim.num_y_cells;
      init.y = lim.max_y - 0.5 * lim.resolution * lim.num_x_cells;
      init.theta = 0.;
    } else {
      h = MakeSearch(m.options().linear_search_window, m.options().angular_search_window,
                     max_range_xy, lim.resolution);
      init = initial_or_null[p];
    }
    CMX_REQUIRE(h.num_scans >= 1 && h.num_scans < (1 << 20), "unsupported number of scans %d",
                h.num_scans);
    out->search[p] = h;
    out->initial[p] = init;
    int r = -1;
    for (size_t k = 0; k < rotations.size(); ++k)
      if (rotations[k].step == h.step && rotations[k].num_angular == h.num_angular) r = static_cast<int>(k);
    if (r < 0) {
      r = static_cast<int>(rotations.size());
      rotations.push_back(Rotation{h.step, h.num_angular, HostRotationTable(h.step, h.num_angular),
                                   rotation_floats});
      rotation_floats += 2 * static_cast<size_t>(h.num_scans);
    }
    rotation_of[p] = r;
    discrete_total += static_cast<size_t>(h.num_scans) * n;
    scans_total += h.num_scans + 1;
    // Upper bound of lowest-resolution candidates per scan: the shrunk window
    // never exceeds nx-1 plus the cell spread of the scan, nor 2*nl.
    const int step = 1 << (m.depth() - 1);
    const double spread_cells = 2.0 * (std::max(max_range_xy, 0.f) / lim.resolution + 2.0);
    auto per_axis = [&](int cells) {
      const double width = std::min(2.0 * h.nl, cells - 1 + spread_cells);
      return static_cast<long long>(width / step) + 2;
    };
    const long long ax = per_axis(lim.num_x_cells), ay = per_axis(lim.num_y_cells);
    const long long cap = ax * ay * h.num_scans;
    CMX_REQUIRE(cap < (1ll << 30), "search too large: %lld lowest-resolution candidates", cap);
    Fast2DProblem& P = out->h_problems[p];
    P.coarse_capacity = static_cast<int>(cap);
    P.coarse_stride = static_cast<int>(ax * ay);
    P.use_planes = m.planes() != nullptr && ax * ay <= kMaxCoarsePerScan &&
                   (ax + m.plane_i() - 1) * (ay + m.plane_j() - 1) <= kMaxBuckets &&
                   ax + m.plane_i() - 1 <= 255 && ay + m.plane_j() - 1 <= 255 &&   // 8-bit bx, by
                   (ax + 2 * m.plane_i() - 2) * (ay + 2 * m.plane_j() - 2) <= kMaxAccCells &&
                   m.depth() > 1;
    const long long acc = (ax + 2 * m.plane_i() - 2) * (ay + 2 * m.plane_j() - 2);
    if (P.use_planes)
      out->plane_acc_cells = std::max<long long>(out->plane_acc_cells, acc);
    // Fused front end: 64-byte planes, the scan + the accumulators within the 64 KB of
    // dynamic LDS a launch gets without opting in to more.
    // (... and the lattice block of a point + 1 within 16 bits: the fused kernel's point words)
    P.use_fused = fused_enabled && P.use_planes && m.plane_stride() == 64 &&
                  n <= kFusedMaxPoints && 4ll * n_pad + 4 * (kFusedMisc + acc) + 1024 <= 64 * 1024 &&
                  (ax + m.plane_i() - 2) * (ay + 2 * m.plane_j() - 2) + (ay + m.plane_j() - 1) <= 65535;
    if (P.use_fused) {
      any_fused = true;
      fused_acc = std::max(fused_acc, acc);
    } else {
      any_unfused = true;
    }
    P.write_all_discrete = out->write_all_discrete ? 1 : 0;
    // Group bounds: three rotations per workgroup, one sum over the dilated level (the kernel's
    // long comment).  Not for the callers that need every exact lowest-resolution score
    // (introspection, depth 1), not where two cells of dilation are a large part of the
    // lowest-resolution window (below 16 cells the bounds stop excluding anything).
    // fast2d_group: 1 never, 2 whenever the planes exist.
    {
      const int sw = Debug().fast2d_group;
      const bool wanted = sw == 1 ? false : sw == 2 ? true : m.depth() >= 5;
      P.group = (wanted && P.use_fused && m.planes_group() != nullptr && m.depth() > 1 &&
                 !out->write_all_discrete && h.num_scans >= kFusedGroup &&
                 4ll * kFusedGroup * n_pad + 4 * (kFusedMisc + acc) + 1024 <= 64 * 1024)
                    ? kFusedGroup : 1;
      P.group_verify = Debug().fast2d_group_verify;
      if (P.group > 1) out->any_group = true;
    }
    P.timeline = nullptr;
    coarse_total += cap;
  }

  // Scratch carving.  The bucketed records exist in HBM only for unfused problems.
  uint32_t* d_discrete =
      ws.dev[2].ReserveAs<uint32_t>(discrete_total + (any_unfused ? 2 * discrete_total + 2 : 0));
  uint2* d_sorted = reinterpret_cast<uint2*>(d_discrete + discrete_total + (discrete_total & 1));
  int4* d_bounds = ws.dev[3].ReserveAs<int4>(scans_total);
  int2* d_dims = ws.dev[4].ReserveAs<int2>(2 * scans_total);
  int2* d_scan_best = d_dims + scans_total;
  int* d_sorted_count = ws.dev[5].ReserveAs<int>(scans_total);
  float* d_cscore = ws.dev[6].ReserveAs<float>(coarse_total);
  int* d_csum = ws.dev[7].ReserveAs<int>(coarse_total);
  const size_t problems_bytes = (num * sizeof(Fast2DProblem) + 255) & ~size_t(255);
  const size_t states_bytes = (num * sizeof(ProblemState) + 255) & ~size_t(255);
  const size_t upload_bytes = problems_bytes + states_bytes + rotation_floats * sizeof(float);
  char* d_upload = static_cast<char*>(ws.dev[8].Reserve(upload_bytes));
  out->d_problems = reinterpret_cast<Fast2DProblem*>(d_upload);
  out->d_states = reinterpret_cast<ProblemState*>(d_upload + problems_bytes);
  const float* d_rotations = reinterpret_cast<const float*>(d_upload + problems_bytes + states_bytes);
  char* h_upload = static_cast<char*>(ws.pinned[1].Reserve(upload_bytes));
  Fast2DProblem* h_prob = reinterpret_cast<Fast2DProblem*>(h_upload);
  ProblemState* h_state = reinterpret_cast<ProblemState*>(h_upload + problems_bytes);
  float* h_rotations = reinterpret_cast<float*>(h_upload + problems_bytes + states_bytes);
  for (const Rotation& r : rotations)
    std::memcpy(h_rotations + r.offset, r.table->data(), r.table->size() * sizeof(float2));

  if (TimelineEnabled() && any_fused) {
    int max_scans = 0;
    for (const HostSearch& h : out->search) max_scans = std::max(max_scans, h.num_scans);
    out->timeline_blocks = (max_scans + 255) / 256 * 256 * num;
    const size_t bytes = static_cast<size_t>(out->timeline_blocks) * kTimelineStamps * 8;
    out->d_timeline = static_cast<unsigned long long*>(ws.dev[15].Reserve(bytes));
    CMX_HIP(hipMemsetAsync(out->d_timeline, 0, bytes, ws.stream));
  }
  // Batches and the work-queue search keep the cells of surviving scans (debug switch
  // fast2d_store_scans overrides).
  const int store_override = Debug().fast2d_store_scans;
  const int store_scans =
      store_override ? store_override - 1 : ((num >= 4 || QueueSearchWanted(n, num)) ? 1 : 0);
  size_t disc_off = 0, scan_off = 0, coarse_off = 0;
  for (int p = 0; p < num; ++p) {
    const Fast2DMatcher& m = *matchers[p];
    const cmx_grid2d_limits& lim = m.limits();
    const HostSearch& h = out->search[p];
    Fast2DProblem& P = out->h_problems[p];
    P.timeline = out->d_timeline;
    P.xyz = d_xyz;
    P.recompute_scans = (P.use_fused && !P.write_all_discrete) ? 1 : 0;
    P.store_scans = store_scans;
    for (int i = 0; i < m.depth(); ++i) P.level[i] = m.level(i);
    P.depth = m.depth();
    P.nx = lim.num_x_cells; P.ny = lim.num_y_cells;
    P.nl = h.nl;
    P.res = lim.resolution; P.max_x = lim.max_x; P.max_y = lim.max_y;
    P.tx = static_cast<float>(out->initial[p].x);
    P.ty = static_cast<float>(out->initial[p].y);
    {  // Quaternion(AngleAxisf(initial_rotation.cast<float>().angle(), Z))
      const float ha = 0.5f * static_cast<float>(out->initial[p].theta);
      P.init_qw = std::cos(ha);
      P.init_qz = std::sin(ha) * 1.f;
    }
    P.num_scans = h.num_scans;
    P.inv_res = 1.0 / P.res;
    P.scan_rot = reinterpret_cast<const float2*>(d_rotations + rotations[rotation_of[p]].offset);
    P.min_s = m.min_s();
    P.score_scale = m.score_scale();
    P.min_score = min_of(p);
    P.planes = m.planes();
    P.planes_group = m.planes_group();
    P.plane_i = m.plane_i();
    P.plane_j = m.plane_j();
    P.plane_stride = m.plane_stride();
    P.discrete = d_discrete + disc_off;
    P.sorted = d_sorted + disc_off;
    P.bounds = d_bounds + scan_off;
    P.coarse_dims = d_dims + scan_off;
    P.scan_best = d_scan_best + scan_off;
    P.sorted_count = d_sorted_count + scan_off;
    P.coarse_score = d_cscore + coarse_off;
    P.coarse_sum = d_csum + coarse_off;
    h_prob[p] = P;
    std::memset(&h_state[p], 0, sizeof(ProblemState));
    const float bound = std::max(min_of(p), 0.f);
    std::memcpy(&h_state[p].best_bits, &bound, sizeof(float));
    disc_off += static_cast<size_t>(h.num_scans) * n;
    scan_off += h.num_scans + 1;
    coarse_off += P.coarse_capacity;
    out->max_scans = std::max(out->max_scans, h.num_scans);
  }
  // (from here to the end of this function: the call's turn at the runtime's launch path)
  LaunchTurn turn;
  // One H2D for the problem descriptors, their initial states and the rotation tables.
  SmallCopyAsync(d_upload, h_upload, upload_bytes, /*to_device=*/true, ws.stream);

  const dim3 per_scan(out->max_scans, num);
  auto mark = [&](const char* name) { if (out->trace) out->trace->Mark(name); };
  mark("upload");
  // Whichever kernel runs first clears the search's list counters.
  int* clear_words = reinterpret_cast<int*>(out->d_misc);
  const int clear_count = out->d_misc ? static_cast<int>(sizeof(Counters) / sizeof(int)) : 0;
  RecordEvent(ws.ev_k0, ws.stream);
  if (any_fused) {
    // Threads per block: with 192 (three waves) ten blocks fit a CU, i.e. a single search's
    // ~2300 rotations are all resident at once and the launch takes one block's latency;
    // batches run several rounds anyway and use full 256-thread blocks.
    // (units of a launch: rotations, or groups of three; a batch that mixes both is sized for
    // single rotations -- surplus workgroups of a grouped problem return at once)
    bool all_group = true;
    for (const Fast2DProblem& P : out->h_problems) all_group = all_group && (!P.use_fused || P.group > 1);
    const int per_unit = all_group ? kFusedGroup : 1;
    const int units = (out->max_scans + per_unit - 1) / per_unit;
    const long long blocks = static_cast<long long>(units) * num;
    // pts | misc | candidate sums | 64 point words per wavefront (at most four)
    const size_t lds = 4 * static_cast<size_t>(n_pad) * (out->any_group ? kFusedGroup : 1) +
                       4 * static_cast<size_t>(kFusedMisc + fused_acc) + 4 * 256;
    out->fused_lds = lds;
    out->fused_acc = static_cast<int>(fused_acc);
    out->d_xyz = d_xyz;
    int threads = 256;
    for (int t : {256, 192, 128}) {
      if (blocks <= FusedResidentBlocks(ws.device, t, lds)) { threads = t; break; }
    }
    if (Debug().fast2d_fused_threads > 0) threads = Debug().fast2d_fused_threads;   // experiments
    if (out->trace && out->trace->enabled())
      fprintf(stderr, "[cmx trace] fused front end: %lld blocks x %d threads, %zu B LDS\n", blocks,
              threads, lds);
    // (grid.x rounded up to a multiple of 256 for the rotation -> block map of the kernel)
    const dim3 fused_grid((units + 255) / 256 * 256, num);
    out->fused_threads = threads;
    if (out->any_group && (Debug().fast2d_group_verify & 1)) {
      // Verification of the group bounds: first every rotation on the level itself (the same
      // descriptors with group = 1; the exact sums stay in coarse_sum), then the launch proper,
      // which compares every bound with them (error 3).
      std::vector<Fast2DProblem> exact(out->h_problems);
      for (Fast2DProblem& P : exact) { P.group = 1; P.store_scans = 0; }
      Fast2DProblem* d_exact = ws.dev[16].ReserveAs<Fast2DProblem>(num);
      CMX_HIP(hipMemcpyAsync(d_exact, exact.data(), num * sizeof(Fast2DProblem), hipMemcpyHostToDevice,
                             ws.stream));
      CMX_HIP(hipStreamSynchronize(ws.stream));        // (`exact` is a local)
      const dim3 exact_grid((out->max_scans + 255) / 256 * 256, num);
      PrepScoreFusedKernel<false><<<exact_grid, threads, lds, ws.stream>>>(
          d_exact, d_xyz, n, out->d_states, static_cast<int>(fused_acc), clear_words, clear_count);
      clear_words = nullptr;
    }
    (out->d_timeline ? PrepScoreFusedKernel<true> : PrepScoreFusedKernel<false>)
        <<<fused_grid, threads, lds, ws.stream>>>(out->d_problems, d_xyz, n, out->d_states,
                                                  static_cast<int>(fused_acc), clear_words,
                                                  clear_count);
    clear_words = nullptr;
    mark("fused");
  }
  if (any_unfused) {
    PrepScansKernel<<<per_scan, 256, 0, ws.stream>>>(out->d_problems, d_xyz, n, out->d_states,
                                                     clear_words, clear_count);
    mark("prep");
    bool any_generic = false;
    int chunk_mask = 0;
    for (const Fast2DProblem& P : out->h_problems) {
      if (P.use_fused) continue;
      if (P.use_planes) chunk_mask |= 1 << (P.plane_stride >> 6);
      else any_generic = true;
    }
    const size_t acc_bytes = static_cast<size_t>(out->plane_acc_cells) * sizeof(int);
    const int plane_threads = 256;
    if (chunk_mask & (1 << 1))
      ScoreCoarsePlanesDwordKernel<<<per_scan, plane_threads, acc_bytes, ws.stream>>>(
          out->d_problems, n, out->d_states);
    if (chunk_mask & (1 << 2))
      ScoreCoarsePlanesKernel<2><<<per_scan, plane_threads, acc_bytes, ws.stream>>>(
          out->d_problems, n, out->d_states);
    if (chunk_mask & (1 << 3))
      ScoreCoarsePlanesKernel<3><<<per_scan, plane_threads, acc_bytes, ws.stream>>>(
          out->d_problems, n, out->d_states);
    if (chunk_mask & (1 << 4))
      ScoreCoarsePlanesKernel<4><<<per_scan, plane_threads, acc_bytes, ws.stream>>>(
          out->d_problems, n, out->d_states);
    if (any_generic)
      ScoreCoarseGenericKernel<<<per_scan, 256, 0, ws.stream>>>(out->d_problems, n, out->d_states);
    mark("coarse");
  }
  RecordEvent(ws.ev_k1, ws.stream);
  CMX_HIP(hipGetLastError());
}

struct BatchResult {
  std::vector<BestLeaf> best;
  std::vector<ProblemState> states;
  double device_ms = 0., dominant_ms = 0.;
  double expansion_ms = 0.;          // wave-per-node stages of the first pass
  int expansion_launches = 0;
  long long expansion_nodes = 0, expansion_lookups = 0;
};

struct ScoreIndex;
void ResolveDepthOne(const PreparedBatch& batch, std::vector<BestLeaf>* best,
                     const std::vector<ProblemState>& states);
void ResolveTies(Workspace& ws, const PreparedBatch& batch, const NodeList& leaves_dev,
                 const CountersSummary& h_counters, std::vector<BestLeaf>* best,
                 const std::vector<ProblemState>& states);

// d_misc: Counters | CountersSummary | SelectState[num] | BestLeaf[num] | ProblemState[num];
// everything after the Counters travels back in one D2H.
size_t SearchTailBytes(int num) {
  return sizeof(CountersSummary) +
         num * (sizeof(SelectState) + sizeof(BestLeaf) + sizeof(ProblemState));
}
size_t SearchMiscBytes(int num) { return sizeof(Counters) + SearchTailBytes(num); }
void ReserveSearchScratch(Workspace& ws, int num, PreparedBatch* batch) {
  batch->d_misc = static_cast<char*>(ws.dev[14].Reserve(SearchMiscBytes(num)));
}

// Full search of a prepared batch.
// (debug switch host_trace: where a caller's wall clock goes -- tools only)
thread_local long long g_host_wait_ns = 0;
std::atomic<long long> g_host_calls{0}, g_host_total_ns{0}, g_host_waited_ns{0};

void RunBranchAndBound(Workspace& ws, const PreparedBatch& batch, BatchResult* result) {
  const int num = batch.num_problems, n = batch.n;
  const int depth = batch.h_problems[0].depth;
  for (const Fast2DProblem& P : batch.h_problems)
    CMX_REQUIRE(P.depth == depth, "all matchers of a batch must share branch_and_bound_depth");

  // Nodes per frontier / leaf buffer.  The debug switch frontier_capacity shrinks the frontiers
  // (tests only) so that the overflow -> strict, chunked retry below is exercised.
  const auto capacity = [](int v, int fallback) {
    return v >= kSubLists ? std::min(v, fallback) / kSubLists * kSubLists : fallback;
  };
  // 64 K nodes per problem (a weak match keeps ~30 k lowest-resolution nodes alive), at
  // least 2 M, at most 32 M (1 GB per buffer): HBM is not the scarce resource here, and
  // an overflow costs a whole second, chunked pass (64 submaps: 34 -> 20 ms per scan).
  const int frontier_default = static_cast<int>(
      std::min<long long>(1ll << 25, std::max<long long>(1ll << 21, 65536ll * num)));
  const int kFrontierCapacity = capacity(Debug().frontier_capacity, frontier_default);
  const int kLeafCapacity = capacity(0, 1 << 20);
  const int kFrontierSub = kFrontierCapacity / kSubLists, kLeafSub = kLeafCapacity / kSubLists;
  // (the two frontier buffers of the level-synchronous path: reserved when that path runs)
  Node2D* d_front[2] = {nullptr, nullptr};
  Node2D* d_leaves = ws.dev[12].ReserveAs<Node2D>(kLeafCapacity);
  // Carved by ReserveSearchScratch before the first kernel of the call, which clears the
  // counters.
  static_assert(sizeof(Counters) % 16 == 0 && sizeof(CountersSummary) % 8 == 0, "alignment");
  char* d_misc = batch.d_misc;
  Counters* d_counters = reinterpret_cast<Counters*>(d_misc);
  char* d_tail = d_misc + sizeof(Counters);
  CountersSummary* d_summary = reinterpret_cast<CountersSummary*>(d_tail);
  SelectState* d_sel = reinterpret_cast<SelectState*>(d_tail + sizeof(CountersSummary));
  BestLeaf* d_best = reinterpret_cast<BestLeaf*>(d_tail + sizeof(CountersSummary) +
                                                 num * sizeof(SelectState));
  ProblemState* d_states_out = reinterpret_cast<ProblemState*>(
      d_tail + sizeof(CountersSummary) + num * (sizeof(SelectState) + sizeof(BestLeaf)));
  auto mark = [&](const char* name) { if (batch.trace) batch.trace->Mark(name); };
  // Stage k reads list k and appends to list k+1 (buffers ping-pong, counters
  // do not: they are all zeroed by the one memset above).
  auto front = [&](int stage) {
    return NodeList{d_front[stage & 1], d_counters->frontier[stage], kFrontierSub};
  };
  const NodeList leaf_list = {d_leaves, d_counters->leaves, kLeafSub};

  // One D2H for the counters' summary + selection state + best leaves.
  const size_t misc_bytes = SearchTailBytes(num);
  char* h_misc = static_cast<char*>(ws.pinned[3].Reserve(misc_bytes));
  CountersSummary* h_counters = reinterpret_cast<CountersSummary*>(h_misc);
  BestLeaf* h_best = reinterpret_cast<BestLeaf*>(h_misc + sizeof(CountersSummary) +
                                                 num * sizeof(SelectState));
  ProblemState* h_states = reinterpret_cast<ProblemState*>(
      h_misc + sizeof(CountersSummary) + num * (sizeof(SelectState) + sizeof(BestLeaf)));
  // (depth > 1: SelectBestKernel stores the tail into h_misc itself)
  const bool direct = Debug().no_direct_results == 0 && misc_bytes % sizeof(unsigned) == 0;
  auto fetch_results = [&](bool published) {
    if (!published) SmallCopyAsync(h_misc, d_tail, misc_bytes, /*to_device=*/false, ws.stream);
    const auto t0 = std::chrono::steady_clock::now();
    CMX_HIP(hipStreamSynchronize(ws.stream));
    g_host_wait_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(
                          std::chrono::steady_clock::now() - t0).count();
  };

  if (depth == 1) {
    SelectDepthOneKernel<<<num, 1024, 0, ws.stream>>>(batch.d_problems, batch.d_states, n, d_best,
                                                      d_states_out);
    CMX_HIP(hipGetLastError());
    RecordEvent(ws.ev_end, ws.stream);
    fetch_results(false);
  } else {
    // ---- dive -------------------------------------------------------------
    // (the launches of the search proper: one turn from the dive to the tree launch)
    std::unique_ptr<LaunchTurn> turn(new LaunchTurn);
    DiveKernel<<<dim3(kSeedsPerProblem * (batch.any_group ? kFusedGroup : 1), num), 256, 0, ws.stream>>>(
        batch.d_problems, batch.d_states, n, leaf_list, d_counters);
    mark("seed+dive");

    // ---- search -------------------------------------------------------------
    // Top of the tree (two levels) per scan, then the subtrees of the
    // survivors down to the leaves on many blocks.
    // Stage shape (tunable for experiments through the environment).
    const int kLevelsPerStage = std::max(0, Debug().fast2d_levels_per_stage);
    const int kWaveLevels = Debug().fast2d_wave_levels > 0 ? Debug().fast2d_wave_levels - 1 : -1;
    // Single searches are latency-bound: one wave stage, then one depth-first
    // kernel down to the leaves.  Batches are throughput-bound: two wave stages,
    // then depth-first stages of two levels.
    const int levels_per_stage = kLevelsPerStage > 0 ? kLevelsPerStage : (num < 4 ? kMaxDepth : 2);
    const int wave_levels = kWaveLevels >= 0 ? kWaveLevels : (num < 4 ? 1 : 2);
    // Frontier sizes are only known on the device; grids are sized for the
    // typical case (a few thousand nodes at the top, tens below) and every
    // kernel grid-strides, so larger frontiers (big batches) still fill the chip.
    // Batches: one problem's nodes stay on one XCD (debug switch fast2d_xcd_affinity overrides).
    const int affinity_override = Debug().fast2d_xcd_affinity;
    const int affinity = affinity_override ? affinity_override - 1 : (num >= 16 ? 1 : 0);
    const int wide_blocks = std::min(4096, 1024 * std::max(1, (num + 3) / 4));
    const int narrow_blocks = std::min(4096, 512 * std::max(1, (num + 3) / 4));
    int num_chunks = 1;
    int strict = 0;
    // Something was dropped (a full frontier / queue / leaf list).  Bounds found so far are real
    // leaf scores and stay valid; the search is repeated in strict mode (prunes ties, records
    // only improving leaves) over more, smaller chunks of scans.  The best leaf found so far is
    // re-found by lowering the bound one ulp.
    const auto prepare_strict_retry = [&]() {
      CMX_REQUIRE(num_chunks < (1 << 12), "branch-and-bound overflow not resolvable");
      if (h_counters->frontier_overflow) num_chunks *= 4;
      strict = 1;
      for (int p = 0; p < num; ++p) {
        const float floor_score = std::max(batch.h_problems[p].min_score, 0.f);
        unsigned floor_bits;
        std::memcpy(&floor_bits, &floor_score, sizeof(float));
        if (h_states[p].best_bits > floor_bits) h_states[p].best_bits -= 1;
      }
      CMX_HIP(hipMemcpyAsync(batch.d_states, h_states, num * sizeof(ProblemState),
                             hipMemcpyHostToDevice, ws.stream));
      CMX_HIP(hipMemsetAsync(d_counters, 0, sizeof(Counters), ws.stream));
    };
    // ---- the work queue: filter + ONE launch for the whole tree and the selection ---------
    bool queue_ok = QueueSearchWanted(n, num);
    for (const Fast2DProblem& P : batch.h_problems) {
      queue_ok = queue_ok && (P.recompute_scans == 0 || P.store_scans != 0);
      for (int l = 0; l + 1 < depth; ++l)
        queue_ok = queue_ok && static_cast<unsigned long long>((P.level[l].qy + 3) >> 2) *
                                       static_cast<unsigned>(P.level[l].qtx) * 128ull < (1ull << 31);
    }
    bool searched = false;
    if (queue_ok) {
      // 16 K nodes per sub-queue (1 M nodes, 64 MB) for up to 16 problems, 64 K per problem
      // beyond; slots are not reused within a call.
      const long long wanted = std::max<long long>(1ll << 20, 65536ll * num);
      int sub_capacity = static_cast<int>(std::min<long long>(wanted, 1ll << 23) / kQueues);
      if (Debug().fast2d_queue_capacity > 0) sub_capacity = Debug().fast2d_queue_capacity;
      TreeQueue queue;
      queue.capacity = sub_capacity;
      queue.slots = static_cast<unsigned long long*>(ws.tagged[0].Acquire(
          static_cast<size_t>(kQueues) * sub_capacity * 8 * sizeof(unsigned long long), ws.stream,
          &queue.epoch));
      d_front[0] = ws.dev[10].ReserveAs<Node2D>(kFrontierCapacity);
      FilterCoarseKernel<<<dim3(DivUp(batch.max_scans, 4), num), 256, 0, ws.stream>>>(
          batch.d_problems, batch.d_states, n, 0, 1, /*strict=*/0, /*affinity=*/0, front(0),
          d_counters);
      mark("filter");
      // (events only under the debug switch `timing`: RecordEvent is a no-op otherwise)
      const bool timed = true;
      if (timed) RecordEvent(ws.ev_x0, ws.stream);
      // One workgroup per CU: 1024 wavefronts.  More of them shorten a single hard search (512
      // workgroups: 240 against 290 us on the hardest of the bench's eight scans) and cost the
      // eight-thread line more than that (17 200 against 19 300 matches/s): wavefronts that find
      // nothing to steal are pure overhead for the searches that share the chip.
      const int blocks = Debug().fast2d_queue_blocks > 0
                             ? Debug().fast2d_queue_blocks
                             : std::min(2048, 256 * std::max(1, (num + 3) / 4));
      TreeQueueKernel<<<blocks, 256, 0, ws.stream>>>(
          batch.d_problems, batch.d_states, n, front(0), queue, leaf_list, d_counters, d_sel,
          d_best, num, d_states_out, d_summary, reinterpret_cast<const unsigned*>(d_tail),
          direct ? reinterpret_cast<unsigned*>(h_misc) : nullptr,
          static_cast<int>(misc_bytes / sizeof(unsigned)),
          batch.trace && batch.trace->enabled() ? 1 : 0,
          Debug().fast2d_queue_lost > 0 ? Debug().fast2d_queue_lost : 3);
      mark("queue");
      CMX_HIP(hipGetLastError());
      if (timed) {
        RecordEvent(ws.ev_x1, ws.stream);
        result->expansion_launches = 1;
      }
      RecordEvent(ws.ev_end, ws.stream);
      turn.reset();
      fetch_results(direct);
      result->expansion_lookups = 64ll * h_counters->wave_gathers;
      for (int p = 0; p < num; ++p)
        for (int k = 0; k < kStatShards; ++k) result->expansion_nodes += h_states[p].expanded_shard[k];
      if (!h_counters->frontier_overflow && !h_counters->leaf_overflow) {
        searched = true;
      } else if (h_counters->leaf_overflow) {
        prepare_strict_retry();
      } else {
        // A sub-queue filled up (it holds 1 K nodes: one wavefront's siblings -- landscapes where
        // nearly everything ties fill it).  The level-synchronous path has room for millions of
        // nodes and reproduces the reference's order among ANY number of tied leaves, which the
        // strict retry cannot: it runs first, from the bounds found so far (real leaf scores),
        // with the lists cleared.
        CMX_HIP(hipMemsetAsync(d_counters, 0, sizeof(Counters), ws.stream));
      }
    }
    turn.reset();      // (the level-synchronous launches below issue as they come)
    if (!searched) {
      d_front[0] = ws.dev[10].ReserveAs<Node2D>(kFrontierCapacity);
      d_front[1] = ws.dev[11].ReserveAs<Node2D>(kFrontierCapacity);
    }
    while (!searched) {
      for (int chunk = 0; chunk < num_chunks; ++chunk) {
        if (chunk > 0 || strict)
          CMX_HIP(hipMemsetAsync(d_counters->frontier, 0, sizeof(d_counters->frontier),
                                 ws.stream));
        FilterCoarseKernel<<<dim3(DivUp(batch.max_scans, 4), num), 256, 0, ws.stream>>>(
            batch.d_problems, batch.d_states, n, chunk, num_chunks, strict, affinity, front(0),
            d_counters);
        mark("filter");
        int stage = 0;
        int top = depth - 1;
        // Wave-per-node level-synchronous expansion of the (wide, shallow-lived)
        // top levels.  (Timed for the statistics in the first pass of a batch: there the
        // expansion is the dominant kernel; a single search's chain of launches is not given
        // two more event packets to wait behind.)
        const bool timed = !strict && chunk == 0 && num >= 4;
        if (timed) RecordEvent(ws.ev_x0, ws.stream);
        for (int used = 0; used < wave_levels && top - 1 >= 1; ++used, --top, ++stage) {
          ExpandWaveKernel<<<used == 0 ? wide_blocks : narrow_blocks, 256, 0, ws.stream>>>(
              batch.d_problems, batch.d_states, n, front(stage), strict, affinity,
              front(stage + 1), d_counters);
          mark("wave");
        }
        if (timed) {
          RecordEvent(ws.ev_x1, ws.stream);
          result->expansion_launches = stage;
        }
        // Block-per-node depth-first stages of kLevelsPerStage levels: the bushy
        // part of the tree near the optimum spreads over many blocks instead of
        // being walked serially by one.
        for (; top > 0; top -= levels_per_stage, ++stage) {
          const int stop = std::max(0, top - levels_per_stage);
          SubtreeKernel<<<narrow_blocks, 256, 0, ws.stream>>>(
              batch.d_problems, batch.d_states, n, front(stage), stop, strict, front(stage + 1),
              leaf_list, d_counters);
          mark("subtree");
        }
      }
      SelectBestKernel<<<1, 1024, 0, ws.stream>>>(
          leaf_list, batch.d_states, d_sel, d_best, num, d_states_out, d_counters, d_summary,
          reinterpret_cast<const unsigned*>(d_tail),
          direct ? reinterpret_cast<unsigned*>(h_misc) : nullptr,
          static_cast<int>(misc_bytes / sizeof(unsigned)));
      mark("select");
      CMX_HIP(hipGetLastError());
      RecordEvent(ws.ev_end, ws.stream);
      fetch_results(direct);
      if (!strict) {
        for (int st = 0; st < result->expansion_launches; ++st)
          result->expansion_nodes += h_counters->frontier_total[st];
        result->expansion_lookups = 64ll * h_counters->wave_gathers;
      }
      if (!h_counters->frontier_overflow && !h_counters->leaf_overflow) break;
      prepare_strict_retry();
    }
  }

  result->best.assign(h_best, h_best + num);
  result->states.assign(h_states, h_states + num);
  if (batch.trace && batch.trace->enabled() && depth > 1) {
    fprintf(stderr, "[cmx trace] list sizes:");
    for (int st = 0; st < kMaxStages; ++st)
      if (h_counters->frontier_total[st])
        fprintf(stderr, " frontier[%d]=%d", st, h_counters->frontier_total[st]);
    long long leaves = 0;
    for (int k = 0; k < kSubLists; ++k) leaves += h_counters->leaves[k];
    fprintf(stderr, " leaves=%lld\n", leaves);
    const unsigned* q = h_counters->queue_stats;
    if (q[5])
      fprintf(stderr, "[cmx trace] work queue: %u chains (%u nodes from the list, %u popped), %u "
              "pushed, %u lost races, %u slot re-reads\n", q[5], q[4], q[0], q[3], q[1], q[2]);
  }
  if (depth > 1) {
    ResolveTies(ws, batch, leaf_list, *h_counters, &result->best, result->states);
  } else {
    ResolveDepthOne(batch, &result->best, result->states);
  }
  float ms = 0.f;
  ms = ElapsedMs(ws.ev_begin, ws.ev_end);
  result->device_ms = ms;
  ms = ElapsedMs(ws.ev_k0, ws.ev_k1);
  result->dominant_ms = ms;
  if (result->expansion_launches > 0) {
    ms = ElapsedMs(ws.ev_x0, ws.ev_x1);
    result->expansion_ms = ms;
  }
}


// Exact tie resolution.  When several leaves share the best score the
// reference returns the one its depth-first search meets first, and at the top
// level that order is whatever std::sort (libstdc++ introsort, unstable) makes
// of equal-score candidates (SM2/fast_...2d.cc:331-332).  The host repeats that
// very sort on the lowest-resolution scores (same initial order, same
// comparator) and ranks the tied leaves by (sorted position of their
// lowest-resolution ancestor, sibling ranks down the tree).  Only runs when the
// device reported a tie.
struct ScoreIndex {
  float score;
  int index;
  bool operator>(const ScoreIndex& other) const { return score > other.score; }
};

// The device keeps the lowest-resolution candidates of scan s at [s * coarse_stride, ...);
// the reference's generation order (scan, x, y) is the dense concatenation.  Host-side
// views for the rare paths that need that order (tie replay, depth 1, introspection).
struct CoarseLayout {
  std::vector<int2> dims;   // [S]
  std::vector<int> off;     // [S + 1] dense prefix
  int Dense(const Fast2DProblem& P, int strided) const {
    return off[strided / P.coarse_stride] + strided % P.coarse_stride;
  }
};
CoarseLayout DownloadLayout(const Fast2DProblem& P) {
  CoarseLayout L;
  const int S = P.num_scans;
  L.dims.resize(S);
  L.off.resize(S + 1);
  CMX_HIP(hipMemcpy(L.dims.data(), P.coarse_dims, S * sizeof(int2), hipMemcpyDeviceToHost));
  L.off[0] = 0;
  for (int s = 0; s < S; ++s) L.off[s + 1] = L.off[s] + L.dims[s].x * L.dims[s].y;
  return L;
}
template <typename T>
std::vector<T> DownloadDense(const Fast2DProblem& P, const CoarseLayout& L, const T* device) {
  const int S = P.num_scans;
  std::vector<T> strided(static_cast<size_t>(S) * P.coarse_stride);
  CMX_HIP(hipMemcpy(strided.data(), device, strided.size() * sizeof(T), hipMemcpyDeviceToHost));
  std::vector<T> dense(L.off[S]);
  for (int s = 0; s < S; ++s)
    std::copy_n(strided.begin() + static_cast<size_t>(s) * P.coarse_stride,
                L.off[s + 1] - L.off[s], dense.begin() + L.off[s]);
  return dense;
}

// Under group bounds the lowest-resolution scores of a problem are upper bounds shared by three
// rotations.  The replay of the reference's order needs the scores themselves: the fused front
// end once more for THIS problem, every rotation summed on the level itself (group = 1; same
// buffers, the search is over).  Rare: leaves that tie for the best score.
void RescoreExact(Workspace& ws, const PreparedBatch& batch, int p) {
  Fast2DProblem P = batch.h_problems[p];
  if (P.group <= 1) return;
  P.group = 1;
  P.group_verify = 0;
  P.store_scans = 0;
  P.timeline = nullptr;
  CMX_HIP(hipMemcpyAsync(batch.d_problems + p, &P, sizeof(P), hipMemcpyHostToDevice, ws.stream));
  CMX_HIP(hipStreamSynchronize(ws.stream));          // (`P` is a local)
  const dim3 grid((P.num_scans + 255) / 256 * 256, 1);
  PrepScoreFusedKernel<false><<<grid, batch.fused_threads, batch.fused_lds, ws.stream>>>(
      batch.d_problems + p, batch.d_xyz, batch.n, batch.d_states + p, batch.fused_acc, nullptr, 0);
  CMX_HIP(hipGetLastError());
  CMX_HIP(hipStreamSynchronize(ws.stream));
}

void ResolveTies(Workspace& ws, const PreparedBatch& batch, const NodeList& leaves_dev,
                 const CountersSummary& h_counters, std::vector<BestLeaf>* best,
                 const std::vector<ProblemState>& states) {
  bool any = false;
  for (const BestLeaf& b : *best) any |= (b.found && b.ties > 1);
  if (!any) return;
  // All recorded leaves.
  std::vector<Node2D> leaves;
  for (int sub = 0; sub < kSubLists; ++sub) {
    const int count = std::min(h_counters.leaves[sub], leaves_dev.sub_capacity);
    if (count <= 0) continue;
    const size_t old = leaves.size();
    leaves.resize(old + count);
    CMX_HIP(hipMemcpy(leaves.data() + old,
                      leaves_dev.nodes + static_cast<size_t>(sub) * leaves_dev.sub_capacity,
                      count * sizeof(Node2D), hipMemcpyDeviceToHost));
  }
  for (int p = 0; p < batch.num_problems; ++p) {
    BestLeaf& b = (*best)[p];
    if (!b.found || b.ties <= 1) continue;
    unsigned best_bits;
    std::memcpy(&best_bits, &b.score, sizeof(float));
    // The dive and the search record the same leaf twice; only distinct leaves tie.
    std::vector<const Node2D*> tied;
    for (const Node2D& nd : leaves) {
      unsigned bits;
      std::memcpy(&bits, &nd.score, sizeof(float));
      if ((nd.problem & 0xffffff) != p || bits != best_bits) continue;
      bool duplicate = false;
      for (const Node2D* t : tied)
        duplicate |= (t->scan == nd.scan && t->dx == nd.dx && t->dy == nd.dy);
      if (!duplicate) tied.push_back(&nd);
      if (tied.size() > 4096) break;   // degenerate input: plenty of ties, stop deduplicating
    }
    if (tied.size() <= 1) continue;
    const Fast2DProblem& P = batch.h_problems[p];
    RescoreExact(ws, batch, p);
    const CoarseLayout layout = DownloadLayout(P);
    const std::vector<float> scores = DownloadDense(P, layout, P.coarse_score);
    const int total = static_cast<int>(scores.size());
    CMX_REQUIRE(total == states[p].coarse_total, "internal error: candidate layout mismatch");
    std::vector<ScoreIndex> sorted(total);
    for (int c = 0; c < total; ++c) sorted[c] = {scores[c], c};
    std::sort(sorted.begin(), sorted.end(), std::greater<ScoreIndex>());
    std::vector<int> position(total);
    for (int i = 0; i < total; ++i) position[sorted[i].index] = i;
    bool have = false;
    unsigned long long best_key = 0;
    for (const Node2D& nd : leaves) {
      unsigned bits;
      std::memcpy(&bits, &nd.score, sizeof(float));
      if ((nd.problem & 0xffffff) != p || bits != best_bits) continue;
      const unsigned long long key =
          (static_cast<unsigned long long>(position[layout.Dense(P, nd.coarse_index)]) << 32) |
          nd.path;
      if (!have || key < best_key) {
        have = true;
        best_key = key;
        b.scan = nd.scan; b.dx = nd.dx; b.dy = nd.dy;
      }
    }
  }
}

// depth == 1: BranchAndBound returns candidates[0] of the std::sort-ed
// lowest-resolution candidates (SM2/fast_...2d.cc:340-343); replay that sort.
void ResolveDepthOne(const PreparedBatch& batch, std::vector<BestLeaf>* best,
                     const std::vector<ProblemState>& states) {
  for (int p = 0; p < batch.num_problems; ++p) {
    BestLeaf& b = (*best)[p];
    const Fast2DProblem& P = batch.h_problems[p];
    if (states[p].error || states[p].coarse_total <= 0) continue;
    const CoarseLayout layout = DownloadLayout(P);
    const std::vector<float> scores = DownloadDense(P, layout, P.coarse_score);
    const int total = static_cast<int>(scores.size());
    std::vector<ScoreIndex> sorted(total);
    for (int c = 0; c < total; ++c) sorted[c] = {scores[c], c};
    std::sort(sorted.begin(), sorted.end(), std::greater<ScoreIndex>());
    const int S = P.num_scans;
    const std::vector<int>& off = layout.off;
    const std::vector<int2>& dims = layout.dims;
    std::vector<int4> bounds(S);
    CMX_HIP(hipMemcpy(bounds.data(), P.bounds, S * sizeof(int4), hipMemcpyDeviceToHost));
    const int c = sorted[0].index;
    const int s = static_cast<int>(std::upper_bound(off.begin(), off.end(), c) - off.begin()) - 1;
    const int local = c - off[s];
    b = BestLeaf{};
    b.score = sorted[0].score;
    b.found = b.score > P.min_score;
    b.scan = s;
    b.dx = bounds[s].x + local / dims[s].y;    // depth 1: step 1, x outer / y inner
    b.dy = bounds[s].z + local % dims[s].y;
    b.ties = 1;
  }
}

void CheckProblemErrors(const BatchResult& r) {
  for (const ProblemState& st : r.states) {
    CMX_REQUIRE(st.error != 1, "scan cell indices exceed the int16 range supported on device");
    CMX_REQUIRE(st.error != 2, "internal error: lowest-resolution candidate capacity exceeded");
    CMX_REQUIRE(st.error != 3, "internal error: a group bound of the fused front end lies below one of its rotations' sums");
  }
}

void MatchBatch(const cmx_fast2d* const* handles, int num, const cmx_pose2d* initial,
                bool full_submap, const float* host_xyz, const cmx_cloud* cloud, int n,
                float min_score, int32_t* found, float* scores, cmx_pose2d* poses,
                cmx_match_stats* stats, const int32_t* full_flags = nullptr,
                const float* min_scores = nullptr) {
  CMX_REQUIRE(handles != nullptr && num >= 1, "no matchers given");
  CMX_REQUIRE(num < (1 << 24), "too many matchers in one batch");
  CMX_REQUIRE(found != nullptr && scores != nullptr && poses != nullptr,
              "score / pose_estimate outputs must not be null");   // CHECK at :232-233
  CMX_REQUIRE(n >= 1, "empty point cloud");
  CMX_REQUIRE(n <= (1 << 24), "point cloud too large");
  std::vector<const Fast2DMatcher*> matchers(num);
  for (int p = 0; p < num; ++p) {
    CMX_REQUIRE(handles[p] != nullptr && handles[p]->impl, "null matcher handle");
    matchers[p] = handles[p]->impl.get();
    CMX_REQUIRE(matchers[p]->device() == matchers[0]->device(),
                "all matchers of a batch must live on the same device");
  }
  const int device = matchers[0]->device();
  // Large batches (from 32 problems on; debug switch fast2d_fanout: 1 never, N > 1 from N on) as
  // INDEPENDENT searches over the host pool: every problem the single-search chain (front end, dive,
  // filter, work-queue tree) on a workspace and stream of its own, sixteen in flight on sixteen
  // hardware queues, instead of the level-synchronous launches over the whole batch -- 64 submaps
  // 6.6 against 9.8 ms, 128: 11.8 against 17.9, 16: the same (profiles/r06g_fanout.txt; with the
  // runtime's four queues of until round 6 it lost: 1.69 against 1.42 ms for 16).  Same results: a
  // problem's search does not depend on its neighbours in the batch.  A caller that finds the
  // pool busy (another batch of the process) runs its problems one after the other itself.
  const int fanout_from = Debug().fast2d_fanout == 0 ? 32 : Debug().fast2d_fanout == 1 ? (1 << 30)
                                                                                       : Debug().fast2d_fanout;
  // (full-submap searches only: a windowed search is a few launches' worth of work, and a batch of
  // them is cheaper in the batch's few launches than in five launches each)
  bool all_full = true;
  for (int p = 0; p < num && all_full; ++p) all_full = full_flags ? full_flags[p] != 0 : full_submap;
  if (num >= fanout_from && all_full && OverrideStream(device) == nullptr) {
    std::vector<cmx_match_stats> part(num);
    ParallelFor(num, 2, [&](int p) {
      MatchBatch(handles + p, 1, initial ? initial + p : nullptr, full_submap, host_xyz, cloud, n,
                 min_scores ? min_scores[p] : min_score, found + p, scores + p, poses + p, &part[p],
                 full_flags ? full_flags + p : nullptr, nullptr);
    });
    cmx_match_stats total{};
    for (const cmx_match_stats& st : part) {
      total.candidates_scored += st.candidates_scored;
      total.coarse_candidates += st.coarse_candidates;
      total.nodes_expanded += st.nodes_expanded;
      total.num_scans += st.num_scans;
      total.device_ms += st.device_ms;                      // (sums over concurrent searches)
      total.dominant_kernel_ms += st.dominant_kernel_ms;
      total.expansion_ms += st.expansion_ms;
      total.expansion_launches += st.expansion_launches;
      total.expansion_nodes += st.expansion_nodes;
      total.expansion_lookups += st.expansion_lookups;
    }
    if (stats) *stats = total;
    return;
  }
  const auto t_call = std::chrono::steady_clock::now();
  g_host_wait_ns = 0;
  WorkspaceLease ws(device);
  const float* d_xyz;
  float max_range;
  if (cloud) {
    CMX_REQUIRE(cloud->device == device, "cloud and matcher are on different devices");
    d_xyz = cloud->xyz;
    max_range = cloud->max_range_xy;
  } else {
    CMX_REQUIRE(host_xyz != nullptr, "point cloud is null");
    float* buf = ws->dev[0].ReserveAs<float>(3 * static_cast<size_t>(n));
    CMX_HIP(hipMemcpyAsync(buf, host_xyz, 3 * sizeof(float) * n, hipMemcpyHostToDevice,
                           ws->stream));
    d_xyz = buf;
    max_range = MaxRangeXY(host_xyz, n);
  }
  RecordEvent(ws->ev_begin, ws->stream);
  PreparedBatch batch;
  StageTrace trace(ws->stream);
  batch.trace = &trace;
  ReserveSearchScratch(*ws, num, &batch);
  PrepareAndScoreCoarse(*ws, matchers.data(), num, initial, full_submap, d_xyz, n, max_range,
                        min_score, &batch, full_flags, min_scores);
  BatchResult result;
  RunBranchAndBound(*ws, batch, &result);
  if (Debug().host_trace) {
    g_host_total_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(
                           std::chrono::steady_clock::now() - t_call).count();
    g_host_waited_ns += g_host_wait_ns;
    const long long calls = ++g_host_calls;
    if (calls % 2000 == 0)
      fprintf(stderr, "[cmx host] fast2d: %lld calls, mean %.1f us per call, of which %.1f us in the final synchronisation\n",
              calls, g_host_total_ns.load() * 1e-3 / calls, g_host_waited_ns.load() * 1e-3 / calls);
  }
  trace.Report();
  if (batch.d_timeline)
    ReportTimeline("PrepScoreFusedKernel", batch.d_timeline, batch.timeline_blocks, ws->stream);
  if (trace.enabled()) {
    for (int p = 0; p < std::min(num, 4); ++p) {
      unsigned long long ex = 0;
      for (int k = 0; k < kStatShards; ++k) ex += result.states[p].expanded_shard[k];
      fprintf(stderr, "[cmx trace] problem %d: coarse %d expanded %llu found %d ties %d\n", p,
              result.states[p].coarse_total, ex, result.best[p].found, result.best[p].ties);
    }
  }
  CheckProblemErrors(result);
  cmx_match_stats total{};
  for (int p = 0; p < num; ++p) {
    const BestLeaf& b = result.best[p];
    const HostSearch& h = batch.search[p];
    const bool ok = b.found && b.score > (min_scores ? min_scores[p] : min_score);
    found[p] = ok ? 1 : 0;
    if (ok) {
      // Candidate2D (SM2/correlative_scan_matcher_2d.h:74-84) and the pose
      // composition of :254-259.
      const double res = matchers[p]->limits().resolution;
      const double cx = -b.dy * res, cy = -b.dx * res;
      const double orientation = (b.scan - h.num_angular) * h.step;
      scores[p] = b.score;
      poses[p].x = batch.initial[p].x + cx;
      poses[p].y = batch.initial[p].y + cy;
      poses[p].theta = batch.initial[p].theta + orientation;
    }
    total.candidates_scored += result.states[p].coarse_total;
    total.coarse_candidates += result.states[p].coarse_total;
    for (int k = 0; k < kStatShards; ++k) {
      total.candidates_scored += result.states[p].scored_shard[k];
      total.nodes_expanded += result.states[p].expanded_shard[k];
    }
    total.num_scans += h.num_scans;
  }
  total.device_ms = result.device_ms;
  total.dominant_kernel_ms = result.dominant_ms;
  total.expansion_ms = result.expansion_ms;
  total.expansion_launches = result.expansion_launches;
  total.expansion_nodes = result.expansion_nodes;
  total.expansion_lookups = result.expansion_lookups;
  if (stats) *stats = total;
}

}  // namespace
}  // namespace cmx

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
using cmx::Guard;

extern "C" {

cmx_status cmx_fast2d_create(const cmx_fast2d_options* options, const cmx_grid2d_limits* limits,
                             const uint16_t* cells, int32_t device, cmx_fast2d** out) {
  return Guard([&] {
    CMX_REQUIRE(options && limits && out, "null argument");
    *out = nullptr;
    std::unique_ptr<cmx_fast2d> h(new cmx_fast2d);
    h->impl.reset(new cmx::Fast2DMatcher(*options, *limits, cells, device));
    *out = h.release();
  });
}

void cmx_fast2d_destroy(cmx_fast2d* matcher) { delete matcher; }

cmx_status cmx_fast2d_match(const cmx_fast2d* matcher, const cmx_pose2d* initial_pose_estimate,
                            const float* point_cloud_xyz, int32_t num_points, float min_score,
                            int32_t* found, float* score, cmx_pose2d* pose_estimate,
                            cmx_match_stats* stats) {
  return Guard([&] {
    CMX_REQUIRE(matcher && initial_pose_estimate, "null argument");
    cmx::MatchBatch(&matcher, 1, initial_pose_estimate, false, point_cloud_xyz, nullptr,
                    num_points, min_score, found, score, pose_estimate, stats);
  });
}

cmx_status cmx_fast2d_match_full_submap(const cmx_fast2d* matcher, const float* point_cloud_xyz,
                                        int32_t num_points, float min_score, int32_t* found,
                                        float* score, cmx_pose2d* pose_estimate,
                                        cmx_match_stats* stats) {
  return Guard([&] {
    CMX_REQUIRE(matcher, "null argument");
    cmx::MatchBatch(&matcher, 1, nullptr, true, point_cloud_xyz, nullptr, num_points, min_score,
                    found, score, pose_estimate, stats);
  });
}

cmx_status cmx_fast2d_match_full_submap_batch(const cmx_fast2d* const* matchers,
                                              int32_t num_matchers, const float* point_cloud_xyz,
                                              int32_t num_points, float min_score,
                                              int32_t* found, float* scores,
                                              cmx_pose2d* pose_estimates, cmx_match_stats* stats) {
  return Guard([&] {
    cmx::MatchBatch(matchers, num_matchers, nullptr, true, point_cloud_xyz, nullptr, num_points,
                    min_score, found, scores, pose_estimates, stats);
  });
}

cmx_status cmx_cloud_upload(const float* point_cloud_xyz, int32_t num_points, int32_t device,
                            cmx_cloud** out) {
  return Guard([&] {
    CMX_REQUIRE(point_cloud_xyz && out && num_points >= 1, "invalid point cloud");
    *out = nullptr;
    cmx::UseDevice(device);
    std::unique_ptr<cmx_cloud> c(new cmx_cloud);
    c->device = device;
    c->num_points = num_points;
    c->host_xyz.assign(point_cloud_xyz, point_cloud_xyz + 3 * static_cast<size_t>(num_points));
    c->max_range_xy = cmx::MaxRangeXY(point_cloud_xyz, num_points);
    float m = 0.f;
    for (int i = 0; i < num_points; ++i) {
      const float x = point_cloud_xyz[3 * i], y = point_cloud_xyz[3 * i + 1],
                  z = point_cloud_xyz[3 * i + 2];
      m = std::max(m, std::sqrt(x * x + y * y + z * z));
    }
    c->max_range_xyz = m;
    {
      double best = 0.;
      for (int i = 0; i < num_points; ++i) {
        const double x = point_cloud_xyz[3 * i], y = point_cloud_xyz[3 * i + 1];
        best = std::max(best, x * x + y * y);
      }
      for (int i = 0; i < num_points && c->far_points.size() <= 64; ++i) {
        const double x = point_cloud_xyz[3 * i], y = point_cloud_xyz[3 * i + 1];
        if (x * x + y * y >= best * (1. - 1e-4)) c->far_points.push_back(i);
      }
      if (c->far_points.size() > 64) c->far_points.clear();
    }
    CMX_HIP(hipMalloc(&c->xyz, 3 * sizeof(float) * num_points));
    hipError_t err = hipMemcpy(c->xyz, point_cloud_xyz, 3 * sizeof(float) * num_points,
                               hipMemcpyHostToDevice);
    if (err != hipSuccess) {
      (void)hipFree(c->xyz);
      c->xyz = nullptr;
      CMX_HIP(err);
    }
    *out = c.release();
  });
}

void cmx_cloud_destroy(cmx_cloud* cloud) {
  if (!cloud) return;
  if (cloud->xyz) {
    (void)hipSetDevice(cloud->device);
    (void)hipFree(cloud->xyz);
  }
  delete cloud;
}

cmx_status cmx_fast2d_match_full_submap_batch_resident(
    const cmx_fast2d* const* matchers, int32_t num_matchers, const cmx_cloud* cloud,
    float min_score, int32_t* found, float* scores, cmx_pose2d* pose_estimates,
    cmx_match_stats* stats) {
  return Guard([&] {
    CMX_REQUIRE(cloud != nullptr, "null cloud");
    cmx::MatchBatch(matchers, num_matchers, nullptr, true, nullptr, cloud, cloud->num_points,
                    min_score, found, scores, pose_estimates, stats);
  });
}

cmx_status cmx_fast2d_match_batch(const cmx_fast2d* const* matchers, int32_t num_matchers,
                                  const cmx_pose2d* initial_pose_estimates,
                                  const int32_t* match_full_submap, const float* min_scores,
                                  const float* point_cloud_xyz, int32_t num_points, int32_t* found,
                                  float* scores, cmx_pose2d* pose_estimates,
                                  cmx_match_stats* stats) {
  return Guard([&] {
    CMX_REQUIRE(match_full_submap != nullptr && min_scores != nullptr, "null argument");
    bool any_windowed = false;
    for (int p = 0; p < num_matchers; ++p) any_windowed |= match_full_submap[p] == 0;
    CMX_REQUIRE(!any_windowed || initial_pose_estimates != nullptr,
                "initial_pose_estimates required for windowed searches");
    cmx::MatchBatch(matchers, num_matchers, initial_pose_estimates, false, point_cloud_xyz, nullptr,
                    num_points, 0.f, found, scores, pose_estimates, stats, match_full_submap,
                    min_scores);
  });
}

cmx_status cmx_fast2d_level_dims(const cmx_fast2d* matcher, int32_t level, int32_t* wide_x,
                                 int32_t* wide_y) {
  return Guard([&] {
    CMX_REQUIRE(matcher && matcher->impl && wide_x && wide_y, "null argument");
    CMX_REQUIRE(level >= 0 && level < matcher->impl->depth(), "level out of range");
    *wide_x = matcher->impl->level(level).wx;
    *wide_y = matcher->impl->level(level).wy;
  });
}

cmx_status cmx_fast2d_level_cells(const cmx_fast2d* matcher, int32_t level, uint8_t* out) {
  return Guard([&] {
    CMX_REQUIRE(matcher && matcher->impl && out, "null argument");
    CMX_REQUIRE(level >= 0 && level < matcher->impl->depth(), "level out of range");
    cmx::UseDevice(matcher->impl->device());
    const cmx::LevelDesc& L = matcher->impl->level(level);
    CMX_HIP(hipMemcpy(out, L.cells, static_cast<size_t>(L.wx) * L.wy, hipMemcpyDeviceToHost));
  });
}

cmx_status cmx_fast2d_debug_prepare(const cmx_fast2d* matcher,
                                    const cmx_pose2d* initial_pose_estimate,
                                    const float* point_cloud_xyz, int32_t num_points,
                                    int32_t full_submap, int32_t* num_scans,
                                    double* angular_step, int32_t* discrete_xy,
                                    int64_t discrete_capacity, int32_t* bounds,
                                    int64_t bounds_capacity, int32_t* coarse_sums,
                                    int64_t sums_capacity, int64_t* num_coarse) {
  return Guard([&] {
    CMX_REQUIRE(matcher && matcher->impl && point_cloud_xyz && num_points >= 1, "bad argument");
    CMX_REQUIRE(full_submap || initial_pose_estimate, "initial pose required");
    const cmx::Fast2DMatcher* m = matcher->impl.get();
    cmx::WorkspaceLease ws(m->device());
    const int n = num_points;
    float* d_xyz = ws->dev[0].ReserveAs<float>(3 * static_cast<size_t>(n));
    CMX_HIP(hipMemcpyAsync(d_xyz, point_cloud_xyz, 3 * sizeof(float) * n, hipMemcpyHostToDevice,
                           ws->stream));
    cmx::RecordEvent(ws->ev_begin, ws->stream);
    cmx::PreparedBatch batch;
    batch.write_all_discrete = true;
    cmx::PrepareAndScoreCoarse(*ws, &m, 1, initial_pose_estimate, full_submap != 0, d_xyz, n,
                               cmx::MaxRangeXY(point_cloud_xyz, n), 0.f, &batch);
    CMX_HIP(hipStreamSynchronize(ws->stream));
    const cmx::Fast2DProblem& P = batch.h_problems[0];
    cmx::ProblemState st;
    CMX_HIP(hipMemcpy(&st, batch.d_states, sizeof(st), hipMemcpyDeviceToHost));
    CMX_REQUIRE(st.error == 0, "device preparation error %d", st.error);
    const int S = P.num_scans;
    const cmx::CoarseLayout layout = cmx::DownloadLayout(P);
    st.coarse_total = layout.off[S];
    if (num_scans) *num_scans = S;
    if (angular_step) *angular_step = batch.search[0].step;
    if (num_coarse) *num_coarse = st.coarse_total;
    if (discrete_xy) {
      CMX_REQUIRE(discrete_capacity >= 2ll * S * n, "discrete_xy capacity too small");
      std::vector<uint32_t> packed(static_cast<size_t>(S) * n);
      CMX_HIP(hipMemcpy(packed.data(), P.discrete, packed.size() * sizeof(uint32_t),
                        hipMemcpyDeviceToHost));
      for (size_t i = 0; i < packed.size(); ++i) {
        discrete_xy[2 * i] = static_cast<short>(packed[i] & 0xffffu);
        discrete_xy[2 * i + 1] = static_cast<short>(packed[i] >> 16);
      }
    }
    if (bounds) {
      CMX_REQUIRE(bounds_capacity >= 4ll * S, "bounds capacity too small");
      std::vector<int4> b(S);
      CMX_HIP(hipMemcpy(b.data(), P.bounds, S * sizeof(int4), hipMemcpyDeviceToHost));
      for (int s = 0; s < S; ++s) {
        bounds[4 * s] = b[s].x; bounds[4 * s + 1] = b[s].y;
        bounds[4 * s + 2] = b[s].z; bounds[4 * s + 3] = b[s].w;
      }
    }
    if (coarse_sums) {
      CMX_REQUIRE(sums_capacity >= st.coarse_total, "coarse_sums capacity too small");
      const std::vector<int> dense = cmx::DownloadDense(P, layout, P.coarse_sum);
      std::copy(dense.begin(), dense.end(), coarse_sums);
    }
  });
}

}  // extern "C"
