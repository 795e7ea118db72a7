// FastCorrelativeScanMatcher2D on gfx950: precomputation-grid stack
// construction, scan preparation, lowest-resolution scoring and a batched,
// level-synchronous branch and bound.
//
// Reference behaviour being replaced:
//   SM2/fast_correlative_scan_matcher_2d.cc:91-186   PrecomputationGrid2D / Stack
//   SM2/correlative_scan_matcher_2d.cc:73-127        ShrinkToFit / GenerateRotatedScans / DiscretizeScans
//   SM2/fast_correlative_scan_matcher_2d.cc:227-378  MatchWithSearchParameters, ScoreCandidates, BranchAndBound
// (SM2 = cartographer/mapping/internal/2d/scan_matching).
//
// Search schedule (any sound schedule returns the reference's best score):
//   1. score every lowest-resolution candidate (reference: :264-274);
//   2. "dive": greedily descend from the best few of them to obtain a real
//      leaf score b0 (a valid lower bound);
//   3. level-synchronous expansion of every node whose upper bound exceeds
//      b0, depth by depth, all problems of a batch together;
//   4. pick the best leaf; ties are resolved in the order the reference's
//      depth-first search would meet them (see SelectBest*).
#include <algorithm>
#include <cmath>

#include "scan_matching_2d.h"

namespace cmx {
namespace {

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

// ---------------------------------------------------------------------------
// Scan preparation: rotate, translate, discretise, ShrinkToFit
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
PrepScansKernel(const Fast2DProblem* __restrict__ problems, const float* __restrict__ xyz, int n,
                ProblemState* __restrict__ states) {
  const Fast2DProblem& P = problems[blockIdx.y];
  const int s = blockIdx.x;
  if (s >= P.num_scans) return;
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
    const int ix = LRoundF64((P.max_y - static_cast<double>(y)) / P.res - 0.5);
    const int iy = LRoundF64((P.max_x - static_cast<double>(x)) / P.res - 0.5);
    if (ix < -32768 || ix > 32767 || iy < -32768 || iy > 32767) bad = 1;
    out[i] = (static_cast<uint32_t>(ix) & 0xffffu) | (static_cast<uint32_t>(iy) << 16);
    lo_x = min(lo_x, -ix);
    lo_y = min(lo_y, -iy);
    hi_x = max(hi_x, P.nx - 1 - ix);
    hi_y = max(hi_y, P.ny - 1 - iy);
  }
  __shared__ int red[4][5];
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
    P.coarse_dims[s] = make_int2((bd.y - bd.x + step) / step, (bd.w - bd.z + step) / step);
    if (bad) atomicMax(&states[blockIdx.y].error, 1);
  }
}

// Exclusive prefix sum of per-scan candidate counts.
__global__ void __launch_bounds__(1024)
CoarseLayoutKernel(const Fast2DProblem* __restrict__ problems, ProblemState* __restrict__ states) {
  const Fast2DProblem& P = problems[blockIdx.x];
  __shared__ int partial[1024];
  const int S = P.num_scans;
  const int chunk = (S + 1023) / 1024;
  const int begin = min(threadIdx.x * chunk, S), end = min(begin + chunk, S);
  int sum = 0;
  for (int s = begin; s < end; ++s) sum += P.coarse_dims[s].x * P.coarse_dims[s].y;
  partial[threadIdx.x] = sum;
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int t = 0; t < 1024; ++t) { const int v = partial[t]; partial[t] = run; run += v; }
    P.coarse_off[S] = run;
    states[blockIdx.x].coarse_total = run;
    if (run > P.coarse_capacity) atomicMax(&states[blockIdx.x].error, 2);
  }
  __syncthreads();
  int run = partial[threadIdx.x];
  for (int s = begin; s < end; ++s) {
    P.coarse_off[s] = run;
    run += P.coarse_dims[s].x * P.coarse_dims[s].y;
  }
}

// ---------------------------------------------------------------------------
// Scoring
// ---------------------------------------------------------------------------

// Integer sum of one candidate over all points, one wave per candidate
// (SM2/fast_...2d.cc:320-329 with GetValue of .h:56-71).
__device__ __forceinline__ int ScoreCandidateWave(const LevelDesc& L, int level,
                                                  const uint32_t* __restrict__ scan, int n, int dx,
                                                  int dy, int lane) {
  const int off = (1 << level) - 1;   // -offset_
  const int ax = dx + off, ay = dy + off;
  int sum = 0;
  for (int i = lane; i < n; i += kWave) {
    const uint32_t p = scan[i];
    const int x = static_cast<short>(p & 0xffffu) + ax;
    const int y = static_cast<short>(p >> 16) + ay;
    if (static_cast<unsigned>(x) < static_cast<unsigned>(L.wx) &&
        static_cast<unsigned>(y) < static_cast<unsigned>(L.wy)) {
      sum += L.cells[x + y * L.wx];
    }
  }
  return WaveSum(sum);
}

__device__ __forceinline__ float ToScore(const Fast2DProblem& P, int sum, int n) {
  // ToScore(sum / float(N))  (SM2/fast_...2d.cc:330-331, .h:74-76)
  return P.min_s + (static_cast<float>(sum) / static_cast<float>(n)) * P.score_scale;
}

__global__ void __launch_bounds__(256)
ScoreCoarseKernel(const Fast2DProblem* __restrict__ problems, int n,
                  const ProblemState* __restrict__ states) {
  const Fast2DProblem& P = problems[blockIdx.y];
  const int s = blockIdx.x;
  if (s >= P.num_scans || states[blockIdx.y].error) return;
  const int level = P.depth - 1;
  const int step = 1 << level;
  const int2 dims = P.coarse_dims[s];
  const int4 bd = P.bounds[s];
  const int base = P.coarse_off[s];
  const uint32_t* scan = P.discrete + static_cast<size_t>(s) * n;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int count = dims.x * dims.y;
  for (int c = wave; c < count; c += 4) {
    const int ix = c / dims.y, iy = c - ix * dims.y;   // x outer, y inner (:295-307)
    const int sum = ScoreCandidateWave(P.level[level], level, scan, n, bd.x + ix * step,
                                       bd.z + iy * step, lane);
    if (lane == 0) {
      P.coarse_sum[base + c] = sum;
      P.coarse_score[base + c] = ToScore(P, sum, n);
    }
  }
}

// ---------------------------------------------------------------------------
// Branch and bound
// ---------------------------------------------------------------------------
struct Counters {           // device, zeroed per call
  int frontier[2];          // ping-pong frontier sizes
  int leaves;
  int overflow;
  int dive[2];
  int pad[2];
};

__device__ __forceinline__ int FindScan(const int* __restrict__ off, int num_scans, int c) {
  int lo = 0, hi = num_scans;   // off[lo] <= c < off[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (off[mid] <= c) lo = mid; else hi = mid;
  }
  return lo;
}

__device__ __forceinline__ Node2D CoarseNode(const Fast2DProblem& P, int problem, int c) {
  const int s = FindScan(P.coarse_off, P.num_scans, c);
  const int local = c - P.coarse_off[s];
  const int2 dims = P.coarse_dims[s];
  const int4 bd = P.bounds[s];
  const int step = 1 << (P.depth - 1);
  const int ix = local / dims.y, iy = local - ix * dims.y;
  Node2D nd;
  nd.problem = problem;
  nd.scan = s;
  nd.dx = bd.x + ix * step;
  nd.dy = bd.z + iy * step;
  nd.score = P.coarse_score[c];
  nd.coarse_index = c;
  nd.path = 0;
  nd.coarse_score = nd.score;
  return nd;
}

// Seeds of the dive: the ~kSeedTarget best lowest-resolution candidates of
// each problem, chosen with a histogram threshold on the integer sums.
constexpr int kSeedTarget = 64;
constexpr int kSeedCap = 256;      // per problem

__global__ void __launch_bounds__(1024)
SeedKernel(const Fast2DProblem* __restrict__ problems, const ProblemState* __restrict__ states,
           int n, Node2D* __restrict__ out, Counters* __restrict__ counters, int out_slot) {
  const int problem = blockIdx.x;
  const Fast2DProblem& P = problems[problem];
  if (states[problem].error) return;
  const int total = states[problem].coarse_total;
  __shared__ int hist[1024];
  __shared__ int threshold_bin;
  __shared__ int taken;
  hist[threadIdx.x] = 0;
  if (threadIdx.x == 0) taken = 0;
  __syncthreads();
  const long long range = 255ll * n + 1;
  for (int c = threadIdx.x; c < total; c += blockDim.x) {
    const int bin = static_cast<int>(P.coarse_sum[c] * 1024ll / range);
    atomicAdd(&hist[bin], 1);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int acc = 0, b = 1023;
    for (; b > 0; --b) {
      acc += hist[b];
      if (acc >= kSeedTarget) break;
    }
    threshold_bin = b;
  }
  __syncthreads();
  const int tb = threshold_bin;
  const float min_score = P.min_score;
  for (int c = threadIdx.x; c < total; c += blockDim.x) {
    const int bin = static_cast<int>(P.coarse_sum[c] * 1024ll / range);
    if (bin >= tb && P.coarse_score[c] > min_score) {
      if (atomicAdd(&taken, 1) < kSeedCap) {
        const int slot = atomicAdd(&counters->dive[out_slot], 1);
        out[slot] = CoarseNode(P, problem, c);
      }
    }
  }
}

// Appends every lowest-resolution candidate in [chunk_begin, chunk_end) whose
// score beats the current best of its problem (reference: :346-350).
__global__ void __launch_bounds__(256)
FilterCoarseKernel(const Fast2DProblem* __restrict__ problems,
                   const ProblemState* __restrict__ states, int chunk, int num_chunks,
                   Node2D* __restrict__ out, int capacity, Counters* __restrict__ counters,
                   int out_slot) {
  const int problem = blockIdx.y;
  const Fast2DProblem& P = problems[problem];
  if (states[problem].error) return;
  const int total = states[problem].coarse_total;
  const int begin = static_cast<int>(static_cast<long long>(total) * chunk / num_chunks);
  const int end = static_cast<int>(static_cast<long long>(total) * (chunk + 1) / num_chunks);
  const float best = __uint_as_float(states[problem].best_bits);
  for (int c = begin + blockIdx.x * blockDim.x + threadIdx.x; c < end;
       c += gridDim.x * blockDim.x) {
    if (P.coarse_score[c] > best) {
      const int slot = atomicAdd(&counters->frontier[out_slot], 1);
      if (slot < capacity) {
        out[slot] = CoarseNode(P, problem, c);
      } else {
        counters->overflow = 1;
      }
    }
  }
}

enum ExpandMode { kExpandFull = 0, kExpandDive = 1 };

// Expands frontier nodes of depth child_level+1 into their <=4 children
// (SM2/fast_...2d.cc:351-368), one block per node, one wave per child.
//   Full mode, child_level > 0: children beating the problem's bound go to `out`.
//   Dive mode, child_level > 0: only the best child is kept.
//   child_level == 0: the best child (first maximum in generation order, as
//   the stable sort of <=4 leaves at :331-332 yields) is recorded as a leaf.
__global__ void __launch_bounds__(256)
ExpandKernel(const Fast2DProblem* __restrict__ problems, ProblemState* __restrict__ states, int n,
             const Node2D* __restrict__ in, const int* __restrict__ in_count, int in_capacity,
             int child_level, int mode, Node2D* __restrict__ out, int* __restrict__ out_count, int out_capacity,
             Node2D* __restrict__ leaves, int* __restrict__ leaf_count, int leaf_capacity,
             int* __restrict__ overflow) {
  __shared__ float child_score[4];
  const int lane = threadIdx.x & 63, k = threadIdx.x >> 6;
  const int count = min(*in_count, in_capacity);
  const int half = 1 << child_level;
  for (int i = blockIdx.x; i < count; i += gridDim.x) {
    const Node2D nd = in[i];
    const Fast2DProblem& P = problems[nd.problem];
    ProblemState& st = states[nd.problem];
    if (mode == kExpandFull && child_level > 0 &&
        !(nd.score > __uint_as_float(st.best_bits))) {
      continue;  // uniform across the block
    }
    const int4 bd = P.bounds[nd.scan];
    const int xo = (k >> 1) * half, yo = (k & 1) * half;
    const bool valid = (nd.dx + xo <= bd.y) && (nd.dy + yo <= bd.w);
    float score = -1.f;
    if (valid) {
      const int sum = ScoreCandidateWave(P.level[child_level], child_level,
                                         P.discrete + static_cast<size_t>(nd.scan) * n, n,
                                         nd.dx + xo, nd.dy + yo, lane);
      score = ToScore(P, sum, n);
    }
    if (lane == 0) child_score[k] = score;
    __syncthreads();
    if (threadIdx.x < 4) {
      const int me = threadIdx.x;
      const float mine = child_score[me];
      int rank = 0, nvalid = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float other = child_score[j];
        if (other >= 0.f) ++nvalid;
        if (j != me && other >= 0.f && (other > mine || (other == mine && j < me))) ++rank;
      }
      if (me == 0) {
        atomicAdd(&st.candidates_scored, static_cast<unsigned long long>(nvalid));
        atomicAdd(&st.nodes_expanded, 1ull);
      }
      if (mine >= 0.f) {
        Node2D child = nd;
        child.dx = nd.dx + (me >> 1) * half;
        child.dy = nd.dy + (me & 1) * half;
        child.score = mine;
        child.path = nd.path | (static_cast<unsigned>(rank) << (2 * child_level));
        if (child_level == 0) {
          if (rank == 0 && mine > P.min_score &&
              mine >= __uint_as_float(st.best_bits)) {
            const int slot = atomicAdd(leaf_count, 1);
            if (slot < leaf_capacity) leaves[slot] = child; else *overflow = 1;
            atomicMax(&st.best_bits, __float_as_uint(mine));
          }
        } else if (mode == kExpandDive) {
          if (rank == 0) {
            const int slot = atomicAdd(out_count, 1);
            if (slot < out_capacity) out[slot] = child; else *overflow = 1;
          }
        } else if (mine > __uint_as_float(st.best_bits)) {
          const int slot = atomicAdd(out_count, 1);
          if (slot < out_capacity) out[slot] = child; else *overflow = 1;
        }
      }
    }
    __syncthreads();
  }
}

// Best-leaf selection in the reference's depth-first visiting order among
// equal scores: higher-scoring lowest-resolution ancestor first (the sorted
// order of :331-332; equal ancestors fall back to generation order), then the
// sibling ranks down the tree.
struct SelectState {         // per problem, device, zeroed per call
  unsigned best_coarse_bits;
  int ties;
  unsigned long long key;    // (coarse_index << 32) | path, minimised
};

__global__ void RelaxBoundsKernel(const Fast2DProblem* __restrict__ problems,
                                  ProblemState* __restrict__ states, int num) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < num) {
    const unsigned floor_bits = __float_as_uint(fmaxf(problems[i].min_score, 0.f));
    if (states[i].best_bits > floor_bits) states[i].best_bits -= 1;
  }
}

__global__ void InitSelectKernel(SelectState* __restrict__ sel, int num) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < num) {
    sel[i].best_coarse_bits = 0;
    sel[i].ties = 0;
    sel[i].key = ~0ull;
  }
}

__global__ void SelectBestPass1(const Node2D* __restrict__ leaves, const int* __restrict__ count,
                                int capacity, const ProblemState* __restrict__ states,
                                SelectState* __restrict__ sel) {
  const int total = min(*count, capacity);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const Node2D nd = leaves[i];
    if (__float_as_uint(nd.score) == states[nd.problem].best_bits) {
      atomicMax(&sel[nd.problem].best_coarse_bits, __float_as_uint(nd.coarse_score));
      atomicAdd(&sel[nd.problem].ties, 1);
    }
  }
}
__global__ void SelectBestPass2(const Node2D* __restrict__ leaves, const int* __restrict__ count,
                                int capacity, const ProblemState* __restrict__ states,
                                SelectState* __restrict__ sel) {
  const int total = min(*count, capacity);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const Node2D nd = leaves[i];
    if (__float_as_uint(nd.score) == states[nd.problem].best_bits &&
        __float_as_uint(nd.coarse_score) == sel[nd.problem].best_coarse_bits) {
      const unsigned long long key =
          (static_cast<unsigned long long>(static_cast<unsigned>(nd.coarse_index)) << 32) | nd.path;
      atomicMin(&sel[nd.problem].key, key);
    }
  }
}
__global__ void SelectBestPass3(const Node2D* __restrict__ leaves, const int* __restrict__ count,
                                int capacity, const ProblemState* __restrict__ states,
                                const SelectState* __restrict__ sel, BestLeaf* __restrict__ best) {
  const int total = min(*count, capacity);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const Node2D nd = leaves[i];
    const unsigned long long key =
        (static_cast<unsigned long long>(static_cast<unsigned>(nd.coarse_index)) << 32) | nd.path;
    if (__float_as_uint(nd.score) == states[nd.problem].best_bits &&
        __float_as_uint(nd.coarse_score) == sel[nd.problem].best_coarse_bits &&
        key == sel[nd.problem].key) {
      BestLeaf b;
      b.score = nd.score; b.scan = nd.scan; b.dx = nd.dx; b.dy = nd.dy;
      b.found = 1; b.ties = sel[nd.problem].ties; b.pad0 = b.pad1 = 0;
      best[nd.problem] = b;   // duplicates (dive + full) carry identical content
    }
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
  uint16_t* d_cells = ws->dev[0].ReserveAs<uint16_t>(count);
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
  CMX_HIP(hipGetLastError());
  CMX_HIP(hipStreamSynchronize(ws->stream));
}

Fast2DMatcher::~Fast2DMatcher() {
  if (stack_mem_) {
    (void)hipSetDevice(device_);
    (void)hipFree(stack_mem_);
  }
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
  int num_problems = 0;
  int n = 0;
  std::vector<HostSearch> search;
  std::vector<cmx_pose2d> initial;
  Fast2DProblem* d_problems = nullptr;
  ProblemState* d_states = nullptr;
  std::vector<Fast2DProblem> h_problems;
};

// Uploads problem descriptors, carves scratch and runs the preparation +
// lowest-resolution scoring kernels.  `d_xyz` is the device point cloud.
void PrepareAndScoreCoarse(Workspace& ws, const Fast2DMatcher* const* matchers, int num,
                           const cmx_pose2d* initial_or_null, bool full_submap,
                           const float* d_xyz, int n, float max_range_xy, float min_score,
                           PreparedBatch* out) {
  out->num_problems = num;
  out->n = n;
  out->search.resize(num);
  out->initial.resize(num);
  out->h_problems.resize(num);

  // Per-problem search parameters and scratch sizes.
  size_t rot_total = 0, discrete_total = 0, scans_total = 0, coarse_total = 0;
  for (int p = 0; p < num; ++p) {
    const Fast2DMatcher& m = *matchers[p];
    const cmx_grid2d_limits& lim = m.limits();
    HostSearch h;
    cmx_pose2d init;
    if (full_submap) {
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
    rot_total += h.num_scans;
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
    const long long per_scan = per_axis(lim.num_x_cells) * per_axis(lim.num_y_cells);
    const long long cap = per_scan * h.num_scans;
    CMX_REQUIRE(cap < (1ll << 30), "search too large: %lld lowest-resolution candidates", cap);
    out->h_problems[p].coarse_capacity = static_cast<int>(cap);
    coarse_total += cap;
  }

  // Scratch carving.
  float2* d_rot = ws.dev[1].ReserveAs<float2>(rot_total);
  uint32_t* d_discrete = ws.dev[2].ReserveAs<uint32_t>(discrete_total);
  int4* d_bounds = ws.dev[3].ReserveAs<int4>(scans_total);
  int2* d_dims = ws.dev[4].ReserveAs<int2>(scans_total);
  int* d_off = ws.dev[5].ReserveAs<int>(scans_total);
  float* d_cscore = ws.dev[6].ReserveAs<float>(coarse_total);
  int* d_csum = ws.dev[7].ReserveAs<int>(coarse_total);
  out->d_problems = ws.dev[8].ReserveAs<Fast2DProblem>(num);
  out->d_states = ws.dev[9].ReserveAs<ProblemState>(num);

  float2* h_rot = ws.pinned[0].ReserveAs<float2>(rot_total);
  Fast2DProblem* h_prob = ws.pinned[1].ReserveAs<Fast2DProblem>(num);
  ProblemState* h_state = ws.pinned[2].ReserveAs<ProblemState>(num);

  size_t rot_off = 0, disc_off = 0, scan_off = 0, coarse_off = 0;
  for (int p = 0; p < num; ++p) {
    const Fast2DMatcher& m = *matchers[p];
    const cmx_grid2d_limits& lim = m.limits();
    const HostSearch& h = out->search[p];
    Fast2DProblem& P = out->h_problems[p];
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
    // GenerateRotatedScans (SM2/correlative_scan_matcher_2d.cc:99-107):
    // delta_theta accumulates in f64, each angle is narrowed to f32.
    double delta_theta = -h.num_angular * h.step;
    for (int s = 0; s < h.num_scans; ++s, delta_theta += h.step) {
      const float ha = 0.5f * static_cast<float>(delta_theta);
      h_rot[rot_off + s] = make_float2(std::cos(ha), std::sin(ha) * 1.f);
    }
    P.scan_rot = d_rot + rot_off;
    P.min_s = m.min_s();
    P.score_scale = m.score_scale();
    P.min_score = min_score;
    P.discrete = d_discrete + disc_off;
    P.bounds = d_bounds + scan_off;
    P.coarse_dims = d_dims + scan_off;
    P.coarse_off = d_off + scan_off;
    P.coarse_score = d_cscore + coarse_off;
    P.coarse_sum = d_csum + coarse_off;
    h_prob[p] = P;
    std::memset(&h_state[p], 0, sizeof(ProblemState));
    const float bound = std::max(min_score, 0.f);
    std::memcpy(&h_state[p].best_bits, &bound, sizeof(float));
    rot_off += h.num_scans;
    disc_off += static_cast<size_t>(h.num_scans) * n;
    scan_off += h.num_scans + 1;
    coarse_off += P.coarse_capacity;
  }
  CMX_HIP(hipMemcpyAsync(d_rot, h_rot, rot_total * sizeof(float2), hipMemcpyHostToDevice,
                         ws.stream));
  CMX_HIP(hipMemcpyAsync(out->d_problems, h_prob, num * sizeof(Fast2DProblem),
                         hipMemcpyHostToDevice, ws.stream));
  CMX_HIP(hipMemcpyAsync(out->d_states, h_state, num * sizeof(ProblemState),
                         hipMemcpyHostToDevice, ws.stream));

  int max_scans = 0;
  for (const HostSearch& h : out->search) max_scans = std::max(max_scans, h.num_scans);
  PrepScansKernel<<<dim3(max_scans, num), 256, 0, ws.stream>>>(out->d_problems, d_xyz, n,
                                                               out->d_states);
  CoarseLayoutKernel<<<num, 1024, 0, ws.stream>>>(out->d_problems, out->d_states);
  CMX_HIP(hipEventRecord(ws.ev_k0, ws.stream));
  ScoreCoarseKernel<<<dim3(max_scans, num), 256, 0, ws.stream>>>(out->d_problems, n,
                                                                 out->d_states);
  CMX_HIP(hipEventRecord(ws.ev_k1, ws.stream));
  CMX_HIP(hipGetLastError());
}

struct BatchResult {
  std::vector<BestLeaf> best;
  std::vector<ProblemState> states;
  double device_ms = 0., dominant_ms = 0.;
};

// Full search of a prepared batch.
void RunBranchAndBound(Workspace& ws, const PreparedBatch& batch, BatchResult* result) {
  const int num = batch.num_problems, n = batch.n;
  int max_depth = 0;
  for (const Fast2DProblem& P : batch.h_problems) max_depth = std::max(max_depth, P.depth);

  const int kFrontierCapacity = 1 << 22;          // nodes per ping-pong buffer
  const int kLeafCapacity = 1 << 22;
  const int dive_capacity = kSeedCap * num;
  Node2D* d_front[2] = {ws.dev[10].ReserveAs<Node2D>(kFrontierCapacity),
                        ws.dev[11].ReserveAs<Node2D>(kFrontierCapacity)};
  Node2D* d_leaves = ws.dev[12].ReserveAs<Node2D>(kLeafCapacity);
  Node2D* d_dive[2] = {ws.dev[13].ReserveAs<Node2D>(2 * dive_capacity), nullptr};
  d_dive[1] = d_dive[0] + dive_capacity;
  char* d_misc = static_cast<char*>(
      ws.dev[14].Reserve(sizeof(Counters) + num * (sizeof(SelectState) + sizeof(BestLeaf))));
  Counters* d_counters = reinterpret_cast<Counters*>(d_misc);
  SelectState* d_sel = reinterpret_cast<SelectState*>(d_misc + sizeof(Counters));
  BestLeaf* d_best = reinterpret_cast<BestLeaf*>(d_misc + sizeof(Counters) +
                                                 num * sizeof(SelectState));
  const int expand_blocks = 2048;

  auto zero_counters = [&] {
    CMX_HIP(hipMemsetAsync(d_counters, 0, sizeof(Counters), ws.stream));
  };
  zero_counters();
  CMX_HIP(hipMemsetAsync(d_best, 0, num * sizeof(BestLeaf), ws.stream));

  // Problems whose depth differs from max_depth simply start lower: nodes of
  // a depth-d problem enter the frontier when the loop reaches level d-1.
  // (All problems of a batch normally share the depth; mixed depths are
  // handled by seeding/filtering per depth below.)
  for (const Fast2DProblem& P : batch.h_problems)
    CMX_REQUIRE(P.depth == max_depth, "all matchers of a batch must share branch_and_bound_depth");

  if (max_depth == 1) {
    // Lowest resolution is already full resolution: every candidate is a leaf.
    // Treat level 0 candidates as children of virtual parents: reuse the
    // filter + selection path by recording them as leaves directly.
  }

  // ---- dive -------------------------------------------------------------
  if (max_depth > 1) {
    SeedKernel<<<num, 1024, 0, ws.stream>>>(batch.d_problems, batch.d_states, n, d_dive[0],
                                            d_counters, 0);
    int cur = 0;
    for (int child_level = max_depth - 2; child_level >= 0; --child_level) {
      ExpandKernel<<<std::min(expand_blocks, dive_capacity), 256, 0, ws.stream>>>(
          batch.d_problems, batch.d_states, n, d_dive[cur], &d_counters->dive[cur], dive_capacity,
          child_level, kExpandDive, d_dive[cur ^ 1], &d_counters->dive[cur ^ 1], dive_capacity, d_leaves,
          &d_counters->leaves, kLeafCapacity, &d_counters->overflow);
      CMX_HIP(hipMemsetAsync(&d_counters->dive[cur], 0, sizeof(int), ws.stream));
      cur ^= 1;
    }
  }

  // ---- full expansion, chunked over the lowest-resolution candidates on
  //      frontier overflow ---------------------------------------------------
  Counters* h_counters = ws.pinned[3].ReserveAs<Counters>(1);
  int num_chunks = 1;
  for (;;) {
    for (int chunk = 0; chunk < num_chunks; ++chunk) {
      CMX_HIP(hipMemsetAsync(&d_counters->frontier[0], 0, 2 * sizeof(int), ws.stream));
      if (max_depth > 1) {
        FilterCoarseKernel<<<dim3(128, num), 256, 0, ws.stream>>>(
            batch.d_problems, batch.d_states, chunk, num_chunks, d_front[0], kFrontierCapacity,
            d_counters, 0);
        int cur = 0;
        for (int child_level = max_depth - 2; child_level >= 0; --child_level) {
          ExpandKernel<<<expand_blocks, 256, 0, ws.stream>>>(
              batch.d_problems, batch.d_states, n, d_front[cur], &d_counters->frontier[cur],
              kFrontierCapacity, child_level, kExpandFull, d_front[cur ^ 1], &d_counters->frontier[cur ^ 1],
              kFrontierCapacity, d_leaves, &d_counters->leaves, kLeafCapacity,
              &d_counters->overflow);
          CMX_HIP(hipMemsetAsync(&d_counters->frontier[cur], 0, sizeof(int), ws.stream));
          cur ^= 1;
        }
      }
    }
    CMX_HIP(hipMemcpyAsync(h_counters, d_counters, sizeof(Counters), hipMemcpyDeviceToHost,
                           ws.stream));
    CMX_HIP(hipStreamSynchronize(ws.stream));
    if (!h_counters->overflow) break;
    // A frontier overflowed: children were dropped.  Redo the expansion with
    // the lowest-resolution candidates split into more chunks; bounds found so
    // far stay valid (they are real leaf scores), recorded leaves are kept.
    CMX_REQUIRE(num_chunks < (1 << 16), "branch-and-bound frontier overflow not resolvable");
    num_chunks *= 4;
    // Restart the leaf record; lowering every bound by one ulp makes the
    // expansion re-find the leaves that achieved it (bounds stay valid: they
    // are below real leaf scores).
    RelaxBoundsKernel<<<DivUp(num, 256), 256, 0, ws.stream>>>(batch.d_problems, batch.d_states,
                                                              num);
    CMX_HIP(hipMemsetAsync(&d_counters->overflow, 0, sizeof(int), ws.stream));
    CMX_HIP(hipMemsetAsync(&d_counters->leaves, 0, sizeof(int), ws.stream));
  }

  // ---- selection ----------------------------------------------------------
  InitSelectKernel<<<DivUp(num, 256), 256, 0, ws.stream>>>(d_sel, num);
  SelectBestPass1<<<64, 256, 0, ws.stream>>>(d_leaves, &d_counters->leaves, kLeafCapacity,
                                             batch.d_states, d_sel);
  SelectBestPass2<<<64, 256, 0, ws.stream>>>(d_leaves, &d_counters->leaves, kLeafCapacity,
                                             batch.d_states, d_sel);
  SelectBestPass3<<<64, 256, 0, ws.stream>>>(d_leaves, &d_counters->leaves, kLeafCapacity,
                                             batch.d_states, d_sel, d_best);
  CMX_HIP(hipGetLastError());
  CMX_HIP(hipEventRecord(ws.ev_end, ws.stream));

  result->best.resize(num);
  result->states.resize(num);
  CMX_HIP(hipMemcpyAsync(result->best.data(), d_best, num * sizeof(BestLeaf),
                         hipMemcpyDeviceToHost, ws.stream));
  CMX_HIP(hipMemcpyAsync(result->states.data(), batch.d_states, num * sizeof(ProblemState),
                         hipMemcpyDeviceToHost, ws.stream));
  CMX_HIP(hipStreamSynchronize(ws.stream));
  float ms = 0.f;
  CMX_HIP(hipEventElapsedTime(&ms, ws.ev_begin, ws.ev_end));
  result->device_ms = ms;
  CMX_HIP(hipEventElapsedTime(&ms, ws.ev_k0, ws.ev_k1));
  result->dominant_ms = ms;
}

void CheckProblemErrors(const BatchResult& r) {
  for (const ProblemState& st : r.states) {
    CMX_REQUIRE(st.error != 1, "scan cell indices exceed the int16 range supported on device");
    CMX_REQUIRE(st.error != 2, "internal error: lowest-resolution candidate capacity exceeded");
  }
}

// depth == 1: the lowest-resolution candidates are the leaves
// (BranchAndBound returns candidates[0], SM2/fast_...2d.cc:340-343).
void SelectDepthOne(Workspace& ws, const PreparedBatch& batch, BatchResult* result) {
  const int num = batch.num_problems;
  result->best.assign(num, BestLeaf{});
  result->states.resize(num);
  CMX_HIP(hipEventRecord(ws.ev_end, ws.stream));
  CMX_HIP(hipMemcpyAsync(result->states.data(), batch.d_states, num * sizeof(ProblemState),
                         hipMemcpyDeviceToHost, ws.stream));
  CMX_HIP(hipStreamSynchronize(ws.stream));
  CheckProblemErrors(*result);
  for (int p = 0; p < num; ++p) {
    const Fast2DProblem& P = batch.h_problems[p];
    const int total = result->states[p].coarse_total;
    const int S = P.num_scans;
    std::vector<float> scores(total);
    std::vector<int> off(S + 1);
    std::vector<int2> dims(S);
    std::vector<int4> bounds(S);
    CMX_HIP(hipMemcpy(scores.data(), P.coarse_score, total * sizeof(float), hipMemcpyDeviceToHost));
    CMX_HIP(hipMemcpy(off.data(), P.coarse_off, (S + 1) * sizeof(int), hipMemcpyDeviceToHost));
    CMX_HIP(hipMemcpy(dims.data(), P.coarse_dims, S * sizeof(int2), hipMemcpyDeviceToHost));
    CMX_HIP(hipMemcpy(bounds.data(), P.bounds, S * sizeof(int4), hipMemcpyDeviceToHost));
    int best = -1;
    for (int c = 0; c < total; ++c)
      if (best < 0 || scores[c] > scores[best]) best = c;
    BestLeaf& b = result->best[p];
    if (best >= 0 && scores[best] > P.min_score) {
      const int s = static_cast<int>(std::upper_bound(off.begin(), off.end(), best) -
                                     off.begin()) - 1;
      const int local = best - off[s];
      b.found = 1;
      b.score = scores[best];
      b.scan = s;
      b.dx = bounds[s].x + local / dims[s].y;
      b.dy = bounds[s].z + local % dims[s].y;
      b.ties = 1;
    }
  }
  float ms = 0.f;
  CMX_HIP(hipEventElapsedTime(&ms, ws.ev_begin, ws.ev_end));
  result->device_ms = ms;
  CMX_HIP(hipEventElapsedTime(&ms, ws.ev_k0, ws.ev_k1));
  result->dominant_ms = ms;
}

void MatchBatch(const cmx_fast2d* const* handles, int num, const cmx_pose2d* initial,
                bool full_submap, const float* host_xyz, const cmx_cloud* cloud, int n,
                float min_score, int32_t* found, float* scores, cmx_pose2d* poses,
                cmx_match_stats* stats) {
  CMX_REQUIRE(handles != nullptr && num >= 1, "no matchers given");
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
  CMX_HIP(hipEventRecord(ws->ev_begin, ws->stream));
  PreparedBatch batch;
  PrepareAndScoreCoarse(*ws, matchers.data(), num, initial, full_submap, d_xyz, n, max_range,
                        min_score, &batch);
  BatchResult result;
  if (matchers[0]->depth() == 1) {
    for (const Fast2DMatcher* m : matchers)
      CMX_REQUIRE(m->depth() == 1, "all matchers of a batch must share branch_and_bound_depth");
    SelectDepthOne(*ws, batch, &result);
  } else {
    RunBranchAndBound(*ws, batch, &result);
    CheckProblemErrors(result);
  }
  cmx_match_stats total{};
  for (int p = 0; p < num; ++p) {
    const BestLeaf& b = result.best[p];
    const HostSearch& h = batch.search[p];
    const bool ok = b.found && b.score > min_score;
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
    total.candidates_scored += result.states[p].coarse_total + result.states[p].candidates_scored;
    total.coarse_candidates += result.states[p].coarse_total;
    total.nodes_expanded += result.states[p].nodes_expanded;
    total.num_scans += h.num_scans;
  }
  total.device_ms = result.device_ms;
  total.dominant_kernel_ms = result.dominant_ms;
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
    CMX_HIP(hipEventRecord(ws->ev_begin, ws->stream));
    cmx::PreparedBatch batch;
    cmx::PrepareAndScoreCoarse(*ws, &m, 1, initial_pose_estimate, full_submap != 0, d_xyz, n,
                               cmx::MaxRangeXY(point_cloud_xyz, n), 0.f, &batch);
    CMX_HIP(hipStreamSynchronize(ws->stream));
    const cmx::Fast2DProblem& P = batch.h_problems[0];
    cmx::ProblemState st;
    CMX_HIP(hipMemcpy(&st, batch.d_states, sizeof(st), hipMemcpyDeviceToHost));
    CMX_REQUIRE(st.error == 0, "device preparation error %d", st.error);
    const int S = P.num_scans;
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
      CMX_HIP(hipMemcpy(coarse_sums, P.coarse_sum, st.coarse_total * sizeof(int),
                        hipMemcpyDeviceToHost));
    }
  });
}

}  // extern "C"
