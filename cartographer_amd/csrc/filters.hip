// Upstream point preparation on gfx950: the voxel filters that define the N of
// every matcher call and the rotational histogram of the 3D loop-closure matcher.
//   sensor/internal/voxel_filter.cc:30-36,38-75,79-115,193-198
//   mapping/internal/3d/scan_matching/rotational_scan_matcher.cc:30-120,164-177
//
// VoxelFilter keeps ONE RANDOM point per voxel: a reservoir sample driven by a
// std::minstd_rand0 that is default-seeded per call and consumed in point order, one
// std::uniform_int_distribution(1, k) draw for the k-th point of a voxel (k >= 2).  That reads
// as inherently sequential, but the only sequential quantity is the position of each draw in
// the generator's stream, and that is a prefix sum:
//   1. voxel key per point (per-axis lround(p / resolution) in f32, packed like the reference);
//   2. stable radix sort of (key, index): a voxel's points become a segment in index order, a
//      point's position in its segment is its k;
//   3. exclusive prefix sum over the points (original order) of [k >= 2] = stream position t;
//   4. minstd_rand0 is x -> 16807 x mod (2^31 - 1): the t-th output is 16807^t mod (2^31 - 1),
//      a 31-step modular power per point; libstdc++'s distribution (downscaling with rejection,
//      bits/uniform_int_dist.h) maps it to [1, k]; the point replaces the reservoir iff the
//      result is k; the LAST replacing point of a segment is the voxel's sample;
//   5. a draw is rejected with probability < k / 2^31 and shifts every later stream position:
//      the kernels flag it, and the (practically never taken) repair pass replays the draws
//      sequentially on one lane from the ranks already computed.
// The result is the reference's point set, bit for bit (against a reference built with this
// libstdc++; the standard leaves the distribution's algorithm to the implementation).
#include <hipcub/hipcub.hpp>

#include <cmath>
#include <cstring>
#include <mutex>

#include "cmx_atan2f.h"
#include "cmx_common.h"
#include "cmx_device.h"

namespace cmx {
namespace {

constexpr unsigned kMinstdA = 16807u, kMinstdM = 2147483647u;      // std::minstd_rand0
constexpr unsigned long long kUrngRange = 2147483645ull;           // max() - min()

__device__ __forceinline__ unsigned MulMod(unsigned a, unsigned b) {
  return static_cast<unsigned>((static_cast<unsigned long long>(a) * b) % kMinstdM);
}
// t-th output (t >= 1) of a default-seeded (state 1) minstd_rand0.
__device__ __forceinline__ unsigned MinstdOutput(unsigned long long t) {
  unsigned result = 1u, base = kMinstdA;
  while (t) {
    if (t & 1ull) result = MulMod(result, base);
    base = MulMod(base, base);
    t >>= 1;
  }
  return result;
}

// GetVoxelCellIndex (voxel_filter.cc:79-86).
__global__ void VoxelKeyKernel(const float* __restrict__ xyz, int n, float resolution,
                               unsigned long long* __restrict__ keys,
                               unsigned* __restrict__ index) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned long long x =
      static_cast<unsigned long long>(static_cast<long long>(LRoundF32(xyz[3 * i] / resolution)));
  const unsigned long long y = static_cast<unsigned long long>(
      static_cast<long long>(LRoundF32(xyz[3 * i + 1] / resolution)));
  const unsigned long long z = static_cast<unsigned long long>(
      static_cast<long long>(LRoundF32(xyz[3 * i + 2] / resolution)));
  keys[i] = (x << 42) + (y << 21) + z;
  index[i] = static_cast<unsigned>(i);
}

// Sorted position p: start of its segment if it is a head, else 0 (max-scanned afterwards).
__global__ void SegmentHeadKernel(const unsigned long long* __restrict__ sorted_keys, int n,
                                  int* __restrict__ head_or_zero) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  head_or_zero[p] = (p > 0 && sorted_keys[p] != sorted_keys[p - 1]) ? p : 0;
}

// Back to point order: k (1-based position in the voxel), the voxel's id (= segment start), the
// draw flag, and the reservoir of the voxel cleared.
__global__ void RankScatterKernel(const unsigned* __restrict__ sorted_index,
                                  const int* __restrict__ seg_start, int n,
                                  int* __restrict__ rank, int* __restrict__ voxel,
                                  int* __restrict__ draws, int* __restrict__ selected) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const int i = static_cast<int>(sorted_index[p]);
  const int k = p - seg_start[p] + 1;
  rank[i] = k;
  voxel[i] = seg_start[p];
  draws[i] = k >= 2 ? 1 : 0;
  if (k == 1) selected[p] = -1;
}

// libstdc++ uniform_int_distribution<int>(1, k) on a minstd_rand0 output g:
// ret = g - 1; scaling = urngrange / k; rejected if ret >= k * scaling; value = ret / scaling + 1.
__global__ void DrawKernel(const int* __restrict__ rank, const int* __restrict__ voxel,
                           const int* __restrict__ stream_pos, int n, int* __restrict__ selected,
                           int* __restrict__ rejected) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int k = rank[i];
  bool replace = true;                     // k == 1: the first point of a voxel is its sample
  if (k >= 2) {
    const unsigned long long ret = MinstdOutput(static_cast<unsigned long long>(stream_pos[i]) + 1) - 1;
    const unsigned long long scaling = kUrngRange / static_cast<unsigned long long>(k);
    if (ret >= static_cast<unsigned long long>(k) * scaling) {
      *rejected = 1;
      return;
    }
    replace = ret / scaling == static_cast<unsigned long long>(k - 1);
  }
  if (replace) atomicMax(&selected[voxel[i]], i);
}

// Repair pass (a draw was rejected somewhere): replays the generator sequentially on one lane.
__global__ void SequentialDrawKernel(const int* __restrict__ rank, const int* __restrict__ voxel,
                                     int n, int* __restrict__ selected) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  unsigned state = 1u;
  for (int i = 0; i < n; ++i) {
    const int k = rank[i];
    bool replace = true;
    if (k >= 2) {
      const unsigned long long scaling = kUrngRange / static_cast<unsigned long long>(k);
      const unsigned long long past = static_cast<unsigned long long>(k) * scaling;
      unsigned long long ret;
      do {
        state = MulMod(state, kMinstdA);
        ret = state - 1u;
      } while (ret >= past);
      replace = ret / scaling == static_cast<unsigned long long>(k - 1);
    }
    if (replace) selected[voxel[i]] = i;       // index order: the last one stays
    else if (k == 1) selected[voxel[i]] = i;
  }
}

__global__ void UsedFlagKernel(const int* __restrict__ voxel, const int* __restrict__ selected,
                               int n, int* __restrict__ used) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  used[i] = selected[voxel[i]] == i ? 1 : 0;
}

__global__ void RangeFlagKernel(const float* __restrict__ xyz, int n, float max_range,
                                int* __restrict__ used) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
  used[i] = sqrtf((x * x + y * y) + z * z) <= max_range ? 1 : 0;   // position.norm()
}

__global__ void CompactKernel(const float* __restrict__ xyz, const int* __restrict__ used,
                              const int* __restrict__ offset, int n, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !used[i]) return;
  const int o = offset[i];
  out[3 * o] = xyz[3 * i];
  out[3 * o + 1] = xyz[3 * i + 1];
  out[3 * o + 2] = xyz[3 * i + 2];
}

__global__ void CompactIndexKernel(const int* __restrict__ used, const int* __restrict__ offset,
                                   int n, int* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && used[i]) out[offset[i]] = i;
}

// out[offset[i]] = map[i] for the kept i: the kept points' indices in the cloud `map` refers to.
__global__ void CompactMappedIndexKernel(const int* __restrict__ used, const int* __restrict__ offset,
                                         int n, const int* __restrict__ map, int* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && used[i]) out[offset[i]] = map[i];
}

// Scratch of one filter call, carved from a workspace.
struct FilterScratch {
  unsigned long long *keys = nullptr, *keys_sorted = nullptr;
  unsigned *index = nullptr, *index_sorted = nullptr;
  int *head = nullptr, *seg_start = nullptr, *rank = nullptr, *voxel = nullptr, *draws = nullptr,
      *stream_pos = nullptr, *selected = nullptr, *used = nullptr, *offset = nullptr,
      *flags = nullptr;          // [0] rejected, [1] count
  void* temp = nullptr;
  size_t temp_bytes = 0;
};

FilterScratch Carve(Workspace& ws, int n) {
  FilterScratch s;
  const size_t N = static_cast<size_t>(std::max(n, 1));
  size_t sort_bytes = 0, scan_bytes = 0, max_bytes = 0;
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, s.keys, s.keys_sorted, s.index,
                                           s.index_sorted, n, 0, 64, ws.stream);
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, s.draws, s.stream_pos, n, ws.stream);
  (void)hipcub::DeviceScan::InclusiveScan(nullptr, max_bytes, s.head, s.seg_start, hipcub::Max(), n,
                                          ws.stream);
  s.temp_bytes = std::max(sort_bytes, std::max(scan_bytes, max_bytes)) + 256;
  char* base = static_cast<char*>(ws.dev[9].Reserve(N * (2 * 8 + 2 * 4 + 9 * 4) + 1024 + s.temp_bytes));
  s.keys = reinterpret_cast<unsigned long long*>(base); base += N * 8;
  s.keys_sorted = reinterpret_cast<unsigned long long*>(base); base += N * 8;
  s.index = reinterpret_cast<unsigned*>(base); base += N * 4;
  s.index_sorted = reinterpret_cast<unsigned*>(base); base += N * 4;
  int** ints[] = {&s.head, &s.seg_start, &s.rank, &s.voxel, &s.draws, &s.stream_pos, &s.selected,
                  &s.used, &s.offset};
  for (int** p : ints) { *p = reinterpret_cast<int*>(base); base += N * 4; }
  s.flags = reinterpret_cast<int*>(base); base += 256;
  base = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(base) + 255) & ~uintptr_t(255));
  s.temp = base;
  return s;
}

// points_used flags of RandomizedVoxelFilterIndices for the device cloud `d_xyz` into s.used,
// their exclusive prefix sum into s.offset; returns the number of points kept.
int VoxelFilterFlags(Workspace& ws, const FilterScratch& s, const float* d_xyz, int n,
                     float resolution) {
  if (n == 0) return 0;
  const int blocks = DivUp(n, 256);
  hipStream_t st = ws.stream;
  CMX_HIP(hipMemsetAsync(s.flags, 0, 8, st));
  VoxelKeyKernel<<<blocks, 256, 0, st>>>(d_xyz, n, resolution, s.keys, s.index);
  size_t bytes = s.temp_bytes;
  CMX_HIP(hipcub::DeviceRadixSort::SortPairs(s.temp, bytes, s.keys, s.keys_sorted, s.index,
                                             s.index_sorted, n, 0, 64, st));
  SegmentHeadKernel<<<blocks, 256, 0, st>>>(s.keys_sorted, n, s.head);
  bytes = s.temp_bytes;
  CMX_HIP(hipcub::DeviceScan::InclusiveScan(s.temp, bytes, s.head, s.seg_start, hipcub::Max(), n, st));
  RankScatterKernel<<<blocks, 256, 0, st>>>(s.index_sorted, s.seg_start, n, s.rank, s.voxel, s.draws,
                                            s.selected);
  bytes = s.temp_bytes;
  CMX_HIP(hipcub::DeviceScan::ExclusiveSum(s.temp, bytes, s.draws, s.stream_pos, n, st));
  DrawKernel<<<blocks, 256, 0, st>>>(s.rank, s.voxel, s.stream_pos, n, s.selected, s.flags);
  int rejected = 0;
  CMX_HIP(hipMemcpyAsync(&rejected, s.flags, sizeof(int), hipMemcpyDeviceToHost, st));
  CMX_HIP(hipStreamSynchronize(st));
  if (rejected) SequentialDrawKernel<<<1, 64, 0, st>>>(s.rank, s.voxel, n, s.selected);
  UsedFlagKernel<<<blocks, 256, 0, st>>>(s.voxel, s.selected, n, s.used);
  bytes = s.temp_bytes;
  CMX_HIP(hipcub::DeviceScan::ExclusiveSum(s.temp, bytes, s.used, s.offset, n, st));
  int last_used = 0, last_offset = 0;
  CMX_HIP(hipMemcpyAsync(&last_used, s.used + (n - 1), sizeof(int), hipMemcpyDeviceToHost, st));
  CMX_HIP(hipMemcpyAsync(&last_offset, s.offset + (n - 1), sizeof(int), hipMemcpyDeviceToHost, st));
  CMX_HIP(hipStreamSynchronize(st));
  CMX_HIP(hipGetLastError());
  return last_used + last_offset;
}

void Compact(Workspace& ws, const FilterScratch& s, const float* d_xyz, int n, float* d_out) {
  if (n == 0) return;
  CompactKernel<<<DivUp(n, 256), 256, 0, ws.stream>>>(d_xyz, s.used, s.offset, n, d_out);
}

// ---------------------------------------------------------------------------
// Small clouds (round 6): the whole filter in ONE workgroup.
// The local trajectory builders filter clouds of 10^2 - 10^3 points (adaptive filter:
// min_num_points 200 / 150, configuration_files/trajectory_builder_2d.lua:26-28), and a call of
// the generic path above is ~13 launches, three small read-backs and two synchronisations per
// VoxelFilter -- 0.14 ms of host time for 0.04 ms of kernels, five to ten times over for an
// AdaptiveVoxelFilter, whose next length depends on the previous count.  Up to kSmallCloud
// points the same steps run out of LDS in one workgroup: bitonic sort of (key, index), segment
// heads, ranks, the prefix sum of the draws, the draws, the reservoir's last writer, the prefix
// sum of the kept flags -- and the adaptive filter's whole search (voxel_filter.cc:38-75), the
// lengths decided by the workgroup itself.  One launch, one synchronisation, the result stored
// straight into pinned host memory.  Same arithmetic, same point sets (tests: both paths against
// the oracle and each other).
// ---------------------------------------------------------------------------
constexpr int kSmallCloud = 4096;
constexpr int kSmallThreads = 1024;

struct SmallFilterLds {       // carved from dynamic LDS; N = the cloud's size rounded up to a power of two
  unsigned long long* key;    // [N]
  unsigned* idx;              // [N]
  int *seg, *rank, *voxel, *pos, *selected;   // [N] each
  int* used;                  // = seg: the segment starts are dead when the flags are written
  int* wave_totals;           // [kSmallThreads / 64]
};
// (N = 4096: 128 KB of arrays + the kernel's 20 KB active list and result flags: under the 160 KB)
__host__ __device__ inline size_t SmallFilterLdsBytes(int N) {
  return static_cast<size_t>(N) * (8 + 4 + 5 * 4) + 64 * 4 + 64;
}
__device__ __forceinline__ SmallFilterLds CarveSmall(unsigned char* base, int N) {
  SmallFilterLds L;
  L.key = reinterpret_cast<unsigned long long*>(base); base += static_cast<size_t>(N) * 8;
  L.idx = reinterpret_cast<unsigned*>(base); base += static_cast<size_t>(N) * 4;
  int** ints[] = {&L.seg, &L.rank, &L.voxel, &L.pos, &L.selected};
  for (int** p : ints) { *p = reinterpret_cast<int*>(base); base += static_cast<size_t>(N) * 4; }
  L.used = L.seg;
  L.wave_totals = reinterpret_cast<int*>(base);
  return L;
}

// In place: data[i] <- sum of data[0 .. i - 1] (i < n <= 4 x blockDim); returns the total.
// Thread t owns elements 4 t .. 4 t + 3.
__device__ __forceinline__ int BlockExclusiveScan4(int* data, int n, int* wave_totals) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  int v[4], sum = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = 4 * t + k;
    v[k] = i < n ? data[i] : 0;
    sum += v[k];
  }
  int incl = sum;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int o = __shfl_up(incl, off, 64);
    if (lane >= off) incl += o;
  }
  if (lane == 63) wave_totals[wave] = incl;
  __syncthreads();
  int before = incl - sum;
  int total = 0;
  for (int w = 0; w < static_cast<int>(blockDim.x >> 6); ++w) {
    const int wt = wave_totals[w];
    if (w < wave) before += wt;
    total += wt;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = 4 * t + k;
    if (i < n) data[i] = before;
    before += v[k];
  }
  __syncthreads();
  return total;
}

// In place inclusive MAX scan (segment starts from head-or-zero flags), same ownership.
__device__ __forceinline__ void BlockInclusiveMaxScan4(int* data, int n, int* wave_totals) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  int v[4], top = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = 4 * t + k;
    v[k] = i < n ? data[i] : 0;
    top = max(top, v[k]);
    v[k] = top;                          // running maximum inside the thread
  }
  int incl = top;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int o = __shfl_up(incl, off, 64);
    if (lane >= off) incl = max(incl, o);
  }
  if (lane == 63) wave_totals[wave] = incl;
  __syncthreads();
  int before = __shfl_up(incl, 1, 64);
  if (lane == 0) before = 0;
  for (int w = 0; w < wave; ++w) before = max(before, wave_totals[w]);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = 4 * t + k;
    if (i < n) data[i] = max(v[k], before);
  }
  __syncthreads();
}

// RandomizedVoxelFilterIndices (voxel_filter.cc:88-131) of the n <= N points xyz[src[i]] (src:
// LDS or null = identity): L.used[i] = 1 for the kept ones, L.pos[i] = their exclusive prefix
// sum; returns the number kept.  Every thread of the workgroup calls it.
__device__ int SmallVoxelFilter(const float* __restrict__ xyz, const int* src, int n, int N,
                                float resolution, const SmallFilterLds& L, int* rejected_flag) {
  const int t = threadIdx.x, T = blockDim.x;
  // 1. keys, padded with the largest key (sorted behind every point)
  for (int i = t; i < N; i += T) {
    unsigned long long key = ~0ull;
    if (i < n) {
      const int at = src ? src[i] : i;
      const unsigned long long x = static_cast<unsigned long long>(
          static_cast<long long>(LRoundF32(xyz[3 * at] / resolution)));
      const unsigned long long y = static_cast<unsigned long long>(
          static_cast<long long>(LRoundF32(xyz[3 * at + 1] / resolution)));
      const unsigned long long z = static_cast<unsigned long long>(
          static_cast<long long>(LRoundF32(xyz[3 * at + 2] / resolution)));
      key = (x << 42) + (y << 21) + z;
    }
    L.key[i] = key;
    L.idx[i] = static_cast<unsigned>(i);
  }
  if (t == 0) *rejected_flag = 0;
  __syncthreads();
  // 2. bitonic sort by (key, index): indices are distinct, so this is the stable sort by key
  for (int k = 2; k <= N; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = t; i < N; i += T) {
        const int partner = i ^ j;
        if (partner > i) {
          const unsigned long long ka = L.key[i], kb = L.key[partner];
          const unsigned ia = L.idx[i], ib = L.idx[partner];
          const bool greater = ka > kb || (ka == kb && ia > ib);
          const bool ascending = (i & k) == 0;
          if (greater == ascending) {
            L.key[i] = kb; L.key[partner] = ka;
            L.idx[i] = ib; L.idx[partner] = ia;
          }
        }
      }
      __syncthreads();
    }
  }
  // 3. segment starts (sorted position p: its own position if it opens a voxel, else 0; max-scan)
  for (int p = t; p < n; p += T) L.seg[p] = (p > 0 && L.key[p] != L.key[p - 1]) ? p : 0;
  __syncthreads();
  BlockInclusiveMaxScan4(L.seg, n, L.wave_totals);
  // 4. back to point order: k (1-based position in the voxel), the voxel (= segment start), the
  //    draw flag; the voxel's reservoir cleared
  for (int p = t; p < n; p += T) {
    const int i = static_cast<int>(L.idx[p]);
    const int k = p - L.seg[p] + 1;
    L.rank[i] = k;
    L.voxel[i] = L.seg[p];
    L.pos[i] = k >= 2 ? 1 : 0;
    if (k == 1) L.selected[p] = -1;
  }
  __syncthreads();
  // 5. stream position of every draw, the draws, the reservoir's last writer
  BlockExclusiveScan4(L.pos, n, L.wave_totals);
  for (int i = t; i < n; i += T) {
    const int k = L.rank[i];
    bool replace = true;
    if (k >= 2) {
      const unsigned long long ret = MinstdOutput(static_cast<unsigned long long>(L.pos[i]) + 1) - 1;
      const unsigned long long scaling = kUrngRange / static_cast<unsigned long long>(k);
      if (ret >= static_cast<unsigned long long>(k) * scaling) {
        *rejected_flag = 1;
        replace = false;
      } else {
        replace = ret / scaling == static_cast<unsigned long long>(k - 1);
      }
    }
    if (replace) atomicMax(&L.selected[L.voxel[i]], i);
  }
  __syncthreads();
  if (*rejected_flag) {
    // a draw was rejected (probability < k / 2^31 each) and shifts every later stream position:
    // one lane replays the generator in point order (SequentialDrawKernel)
    if (t == 0) {
      unsigned state = 1u;
      for (int i = 0; i < n; ++i) {
        const int k = L.rank[i];
        if (k >= 2) {
          const unsigned long long scaling = kUrngRange / static_cast<unsigned long long>(k);
          const unsigned long long past = static_cast<unsigned long long>(k) * scaling;
          unsigned long long ret;
          do {
            state = MulMod(state, kMinstdA);
            ret = state - 1u;
          } while (ret >= past);
          if (ret / scaling == static_cast<unsigned long long>(k - 1)) L.selected[L.voxel[i]] = i;
        } else {
          L.selected[L.voxel[i]] = i;
        }
      }
    }
    __syncthreads();
  }
  // 6. points_used and their prefix sum
  for (int i = t; i < n; i += T) {
    const int u = L.selected[L.voxel[i]] == i ? 1 : 0;
    L.used[i] = u;
    L.pos[i] = u;
  }
  __syncthreads();
  return BlockExclusiveScan4(L.pos, n, L.wave_totals);
}

// mode 0: VoxelFilter(resolution = length);  mode 1: AdaptiveVoxelFilter(max_length = length,
// min_num_points, max_range).  out[0] = number kept, out[1 ..] = their indices in the input
// (ascending), then (out_xyz) their coordinates: pinned host memory.
__global__ void __launch_bounds__(kSmallThreads)
SmallFilterKernel(const float* __restrict__ xyz, int n0, int N, int mode, float length,
                  float min_num_points, float max_range, int* __restrict__ out,
                  float* __restrict__ out_xyz) {
  extern __shared__ __attribute__((aligned(16))) unsigned char small_smem[];
  const SmallFilterLds L = CarveSmall(small_smem, N);
  // behind the filter's arrays: the active list (adaptive: the points in range), the best
  // result so far as flags, three control words
  int* src = L.wave_totals + 64;
  unsigned char* best_used = reinterpret_cast<unsigned char*>(src + N);
  __shared__ int rejected, ctl_n, ctl_state;
  __shared__ float ctl_length;
  const int t = threadIdx.x, T = blockDim.x;
  int n = n0;
  const int* active = nullptr;
  if (mode == 1) {
    // FilterByMaxRange (voxel_filter.cc:30-36)
    for (int i = t; i < n0; i += T) {
      const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
      const int u = sqrtf((x * x + y * y) + z * z) <= max_range ? 1 : 0;   // position.norm()
      L.used[i] = u;
      L.pos[i] = u;
    }
    __syncthreads();
    n = BlockExclusiveScan4(L.pos, n0, L.wave_totals);
    for (int i = t; i < n0; i += T)
      if (L.used[i]) src[L.pos[i]] = i;
    __syncthreads();
    active = src;
  }
  int kept = n;
  bool have_flags = false;             // false: every active point is kept
  if (mode == 0) {
    kept = SmallVoxelFilter(xyz, active, n, N, length, L, &rejected);
    for (int i = t; i < n; i += T) best_used[i] = static_cast<unsigned char>(L.used[i]);
    have_flags = true;
    __syncthreads();
  } else if (!(static_cast<float>(n) <= min_num_points)) {
    // AdaptivelyVoxelFiltered (voxel_filter.cc:38-75): the same sequence of VoxelFilter calls;
    // every thread walks the same control flow (counts come back uniform).
    const auto evaluate = [&](float len, bool accept_always, int* count) {
      const int m = SmallVoxelFilter(xyz, active, n, N, len, L, &rejected);
      *count = m;
      const bool accept = accept_always || static_cast<float>(m) >= min_num_points;
      if (accept) {
        for (int i = t; i < n; i += T) best_used[i] = static_cast<unsigned char>(L.used[i]);
        __syncthreads();
      }
      return accept;
    };
    int m = 0;
    bool done = false;
    evaluate(length, true, &m);                      // result = VoxelFilter(cloud, max_length)
    kept = m;
    have_flags = true;
    if (static_cast<float>(m) >= min_num_points) done = true;
    for (float high_length = length; !done && high_length > 1e-2f * length; high_length /= 2.f) {
      float low_length = high_length / 2.f;
      evaluate(low_length, true, &m);                // result = VoxelFilter(cloud, low_length)
      kept = m;
      if (static_cast<float>(m) >= min_num_points) {
        while ((high_length - low_length) / low_length > 1e-1f) {
          const float mid_length = (low_length + high_length) / 2.f;
          if (evaluate(mid_length, false, &m)) {
            low_length = mid_length;
            kept = m;
          } else {
            high_length = mid_length;
          }
        }
        done = true;
      }
    }
  }
  (void)ctl_n; (void)ctl_state; (void)ctl_length;
  // ---- the result: indices into the input, ascending, and the points -------------------------
  for (int i = t; i < n; i += T) {
    const int u = have_flags ? best_used[i] : 1;
    L.used[i] = u;
    L.pos[i] = u;
  }
  __syncthreads();
  const int total = BlockExclusiveScan4(L.pos, n, L.wave_totals);
  for (int i = t; i < n; i += T) {
    if (!L.used[i]) continue;
    const int at = active ? active[i] : i;
    const int o = L.pos[i];
    out[1 + o] = at;
    if (out_xyz) {
      out_xyz[3 * o] = xyz[3 * at];
      out_xyz[3 * o + 1] = xyz[3 * at + 1];
      out_xyz[3 * o + 2] = xyz[3 * at + 2];
    }
  }
  if (t == 0) out[0] = total;
  (void)kept;
}

// Host side of the small-cloud path: one upload, one launch, one synchronisation.  `filtered_xyz`
// and `kept_indices` may each be null.
void SmallFilter(int mode, const float* point_cloud_xyz, int n, float length, float min_num_points,
                 float max_range, int device, float* filtered_xyz, int32_t* kept_indices,
                 int32_t* num_filtered) {
  WorkspaceLease ws(device);
  int N = 64;
  while (N < n) N <<= 1;
  const size_t lds = SmallFilterLdsBytes(N) + static_cast<size_t>(N) * 5 + 64;
  OptInLds(reinterpret_cast<const void*>(SmallFilterKernel), device, 160 * 1024 - 256);
  // pinned: the cloud up | count + indices | points down
  const size_t in_bytes = (12 * static_cast<size_t>(n) + 255) & ~size_t{255};
  const size_t idx_bytes = (4 * static_cast<size_t>(n + 1) + 255) & ~size_t{255};
  char* h = ws->pinned[0].ReserveAs<char>(in_bytes + idx_bytes + in_bytes);
  std::memcpy(h, point_cloud_xyz, 12 * static_cast<size_t>(n));
  float* d_xyz = ws->dev[0].ReserveAs<float>(3 * static_cast<size_t>(n) + 64);
  SmallCopyAsync(d_xyz, h, in_bytes, /*to_device=*/true, ws->stream);
  int* h_out = reinterpret_cast<int*>(h + in_bytes);
  float* h_xyz = reinterpret_cast<float*>(h + in_bytes + idx_bytes);
  SmallFilterKernel<<<1, kSmallThreads, lds, ws->stream>>>(d_xyz, n, N, mode, length, min_num_points,
                                                           max_range, h_out, filtered_xyz ? h_xyz : nullptr);
  CMX_HIP(hipGetLastError());
  CMX_HIP(hipStreamSynchronize(ws->stream));
  const int kept = h_out[0];
  CMX_REQUIRE(kept >= 0 && kept <= n, "internal error: small filter count %d of %d", kept, n);
  if (kept_indices) std::memcpy(kept_indices, h_out + 1, 4 * static_cast<size_t>(kept));
  if (filtered_xyz) std::memcpy(filtered_xyz, h_xyz, 12 * static_cast<size_t>(kept));
  *num_filtered = kept;
}

// (debug switch filters_generic = 1: every cloud through the multi-launch path -- the parity
// partner of the one-workgroup path)
bool UseSmallFilter(int n) { return n >= 1 && n <= kSmallCloud && Debug().filters_generic == 0; }

// ---------------------------------------------------------------------------
// RotationalScanMatcher::ComputeHistogram
// ---------------------------------------------------------------------------
constexpr float kMinDistance = 0.2f, kMaxDistance = 0.9f, kSliceHeight = 0.2f;

__global__ void SliceKeyKernel(const float* __restrict__ xyz, int n, unsigned* __restrict__ keys,
                               unsigned* __restrict__ index) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // std::map<int, ...> order: ascending RoundToInt(z / kSliceHeight); biased to unsigned.
  keys[i] = static_cast<unsigned>(LRoundF32(xyz[3 * i + 2] / kSliceHeight)) ^ 0x80000000u;
  index[i] = static_cast<unsigned>(i);
}

// One block per slice (slices = segments of equal key in the sorted order).  ComputeCentroid
// (:47-53) is a SEQUENTIAL f32 sum in slice order: the block stages the coordinates in LDS as
// three zero-padded rows, one lane per coordinate adds its row in index order (ChainSumLds,
// cmx_device.h: ~7 cycles per dependent addition; x + 0 = x exactly).  Every thread then computes
// its points' angle around the centroid with libm's own atan2f arithmetic (cmx_atan2f.h); points
// closer than kMinDistance are dropped (key = all ones sorts them behind the slice).
// Sort key: slice rank << 32 | orderable angle.
constexpr int kSliceChunk = 2048;     // points staged in LDS per pass

__global__ void __launch_bounds__(256)
SliceAngleKernel(const float* __restrict__ xyz, const unsigned* __restrict__ sorted_index,
                 const int* __restrict__ slice_begin, int num_slices, int n,
                 unsigned long long* __restrict__ keys2, unsigned* __restrict__ index2) {
  const int s = blockIdx.x;
  const int begin = slice_begin[s], end = s + 1 < num_slices ? slice_begin[s + 1] : n;
  __shared__ __attribute__((aligned(16))) float rows[3][kSliceChunk];
  __shared__ float c[3], run[3];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (threadIdx.x < 3) run[threadIdx.x] = 0.f;
  for (int p0 = begin; p0 < end; p0 += kSliceChunk) {
    const int m = min(kSliceChunk, end - p0), m_pad = (m + 63) & ~63;
    __syncthreads();
    for (int k = threadIdx.x; k < m_pad; k += blockDim.x) {
      float x = 0.f, y = 0.f, z = 0.f;
      if (k < m) {
        const int i = sorted_index[p0 + k];
        x = xyz[3 * i]; y = xyz[3 * i + 1]; z = xyz[3 * i + 2];
      }
      rows[0][k] = x; rows[1][k] = y; rows[2][k] = z;
    }
    __syncthreads();
    if (wave < 3 && lane == 0) run[wave] = ChainSumLds(rows[wave], m_pad, run[wave]);
  }
  __syncthreads();
  if (threadIdx.x < 3) c[threadIdx.x] = run[threadIdx.x] / static_cast<float>(end - begin);
  __syncthreads();
  for (int p = begin + threadIdx.x; p < end; p += blockDim.x) {
    const int i = sorted_index[p];
    const float dx = xyz[3 * i] - c[0], dy = xyz[3 * i + 1] - c[1];
    unsigned long long key = (static_cast<unsigned long long>(s) << 32) | 0xffffffffull;
    if (!(sqrtf(dx * dx + dy * dy) < kMinDistance)) {
      const unsigned bits = FloatToBits(Atan2fGlibc(dy, dx));
      key = (static_cast<unsigned long long>(s) << 32) |
            (bits ^ ((bits >> 31) ? 0xffffffffu : 0x80000000u));
      if ((key & 0xffffffffull) == 0xffffffffull) key -= 1;     // keep the "dropped" key unique
    }
    keys2[p] = key;
    index2[p] = static_cast<unsigned>(i);
  }
}

// Per slice, in angle order: AddPointCloudSliceToHistogram's walk (:55-83), in three steps.
//  1. The centroid of the sorted slice: sequential sums of the x and y rows (as above).
//  2. The `last_point_position` chain.  It only moves when a point lies further than
//     kMaxDistance from it, so ONE WAVEFRONT walks 64 points per step: every lane tests its point
//     against the current `last`, a ballot finds the first that moves it, the lanes before it
//     record `last` for their point, the walk resumes behind it.  A slice of a range scan moves
//     `last` every few dozen points: n / 64 + (moves) steps instead of n dependent iterations.
//  3. All threads: the vote (bucket, value) of every point from its recorded `last`, with the
//     reference's expressions and libm's atan2f arithmetic, written at the point's position in
//     the global (slice, angle) order -- the order in which the reference adds them.
// The additions themselves are HistogramChainKernel's.
constexpr unsigned kNoVote = 0xffffu;      // bucket key of a point without a vote (> any bucket)

__global__ void __launch_bounds__(256)
SliceWalkKernel(const float* __restrict__ xyz, const unsigned long long* __restrict__ sorted_keys2,
                const unsigned* __restrict__ sorted_index2, const int* __restrict__ slice_begin,
                int num_slices, int n, int histogram_size, float min_distance_sq,
                float max_distance_sq, unsigned* __restrict__ vote_bucket,
                unsigned* __restrict__ vote_value) {
  __shared__ __attribute__((aligned(16))) float X[kSliceChunk], Y[kSliceChunk];
  __shared__ float LX[kSliceChunk], LY[kSliceChunk];
  __shared__ unsigned char near_centroid[kSliceChunk];   // direction.norm() < kMinDistance
  __shared__ int s_kept_end;
  __shared__ float s_c[2], s_run[2], s_last[2];
  const int s = blockIdx.x;
  const int begin = slice_begin[s], end = s + 1 < num_slices ? slice_begin[s + 1] : n;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (threadIdx.x == 0) { s_kept_end = end; s_run[0] = s_run[1] = 0.f; }
  __syncthreads();
  // Dropped points (closer than kMinDistance to the first centroid) sort behind the kept ones.
  for (int p = begin + threadIdx.x; p < end; p += blockDim.x)
    if ((sorted_keys2[p] & 0xffffffffull) == 0xffffffffull) atomicMin(&s_kept_end, p);
  __syncthreads();
  const int kept_end = s_kept_end;
  for (int p = kept_end + threadIdx.x; p < end; p += blockDim.x) {
    vote_bucket[p] = kNoVote;
    vote_value[p] = 0u;
  }
  if (kept_end <= begin) return;
  const float kNaN = __uint_as_float(0x7fc00000u);
  for (int pass = 0; pass < 2; ++pass) {
    for (int p0 = begin; p0 < kept_end; p0 += kSliceChunk) {
      const int m = min(kSliceChunk, kept_end - p0), m_pad = (m + 63) & ~63;
      __syncthreads();
      for (int k = threadIdx.x; k < m_pad; k += blockDim.x) {
        float x = 0.f, y = 0.f;
        if (k < m) {
          const int i = sorted_index2[p0 + k];
          x = xyz[3 * i]; y = xyz[3 * i + 1];
        }
        X[k] = x; Y[k] = y;
        if (pass == 1) {
          const float ex = x - s_c[0], ey = y - s_c[1];
          near_centroid[k] = sqrtf(ex * ex + ey * ey) < kMinDistance ? 1 : 0;
        }
      }
      __syncthreads();
      if (pass == 0) {
        if (wave < 2 && lane == 0) s_run[wave] = ChainSumLds(wave == 0 ? X : Y, m_pad, s_run[wave]);
        continue;
      }
      if (wave == 0) {
        // 64 points per window, kept in registers while `last` moves inside it: a move costs a
        // dozen dependent vector instructions, a ballot and two v_readlane -- no LDS round trip,
        // no square root (distance < kMin <=> d2 < min_distance_sq, distance > kMax <=> d2 >
        // max_distance_sq: the host rounds the two thresholds so that the correctly rounded
        // sqrtf of the reference decides the same way, cmx_compute_histogram).
        float lx = p0 == begin ? X[0] : s_last[0], ly = p0 == begin ? Y[0] : s_last[1];
        for (int c = 0; c < m; c += 64) {
          const int k = c + lane;
          const bool valid = k < m;
          const float px = valid ? X[k] : 0.f, py = valid ? Y[k] : 0.f;
          const bool near = valid && near_centroid[k] != 0;
          float my_lx = 0.f, my_ly = 0.f;
          for (int start = 0;;) {
            const float dx = px - lx, dy = py - ly;
            const float d2 = dx * dx + dy * dy;
            const bool skip = d2 < min_distance_sq || near;
            const unsigned long long moves =
                __ballot(valid && lane >= start && !skip && d2 > max_distance_sq);
            const int j = moves ? __builtin_ctzll(moves) : 64;
            if (lane >= start && lane <= j) {      // (lane j: the point that becomes `last` -- no vote)
              my_lx = lane == j ? kNaN : lx;
              my_ly = ly;
            }
            if (j == 64) break;
            lx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(px), j));
            ly = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(py), j));
            start = j + 1;
          }
          if (valid) { LX[k] = my_lx; LY[k] = my_ly; }
        }
        if (lane == 0) { s_last[0] = lx; s_last[1] = ly; }
      }
      __syncthreads();
      const float cx = s_c[0], cy = s_c[1];
      const float kPi = static_cast<float>(M_PI);
      for (int k = threadIdx.x; k < m; k += blockDim.x) {
        unsigned bucket = kNoVote;
        float v = 0.f;
        const float lx = LX[k];
        if (!(lx != lx)) {
          const float px = X[k], py = Y[k];
          const float dx = px - lx, dy = py - LY[k];
          const float ex = px - cx, ey = py - cy;
          const float distance = sqrtf(dx * dx + dy * dy);
          const float direction_norm = sqrtf(ex * ex + ey * ey);
          // (distance > kMaxDistance cannot occur here: that point moved `last`)
          if (!(distance < kMinDistance || direction_norm < kMinDistance)) {
            float angle = Atan2fGlibc(dy, dx);
            const float ndx = dx / distance, ndy = dy / distance;
            const float nex = ex / direction_norm, ney = ey / direction_norm;
            v = fmaxf(0.f, 1.f - fabsf(ndx * nex + ndy * ney));
            while (angle > kPi) angle -= kPi;
            while (angle < 0.f) angle += kPi;
            const float zero_to_one = angle / kPi;
            bucket = static_cast<unsigned>(min(
                max(LRoundF32(histogram_size * zero_to_one - 0.5f), 0), histogram_size - 1));
          }
        }
        vote_bucket[p0 + k] = bucket;
        vote_value[p0 + k] = __float_as_uint(v);
      }
    }
    __syncthreads();
    if (pass == 0 && threadIdx.x < 2)
      s_c[threadIdx.x] = s_run[threadIdx.x] / static_cast<float>(kept_end - begin);
    __syncthreads();
  }
}

// histogram[b] = the votes of bucket b added one by one in the reference's order -- slices
// ascending, a slice's points by angle (:85-120, :164-177) -- into a zero: `votes` are the
// (bucket, value) pairs stably sorted by bucket (a radix sort keeps the order inside a bucket), a
// wavefront per bucket finds its segment and adds it up.  f32 addition does not associate: any other
// order (per-slice partial histograms, atomics) differs in the last bits.
__global__ void __launch_bounds__(64)
HistogramChainKernel(const unsigned* __restrict__ sorted_bucket,
                     const unsigned* __restrict__ sorted_value, int n, int histogram_size,
                     float* __restrict__ histogram) {
  // One wavefront per bucket: the segment is staged in LDS 2048 values at a time (coalesced
  // loads, zero-padded to 64: h + 0 = h, the votes are >= 0) and one lane adds it up in order
  // (ChainSumLds: ~7 cycles per dependent addition instead of a global round trip per eight).
  __shared__ __attribute__((aligned(16))) float row[kSliceChunk];
  const int b = blockIdx.x, lane = threadIdx.x;
  const auto lower_bound = [&](unsigned key) {
    int lo = 0, hi = n;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (sorted_bucket[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
  };
  const int first = lower_bound(static_cast<unsigned>(b)), last = lower_bound(static_cast<unsigned>(b) + 1);
  float h = 0.f;
  for (int c = first; c < last; c += kSliceChunk) {
    const int m = min(kSliceChunk, last - c), m_pad = (m + 63) & ~63;
    __syncthreads();
    for (int k = lane; k < m_pad; k += 64) row[k] = k < m ? __uint_as_float(sorted_value[c + k]) : 0.f;
    __syncthreads();
    if (lane == 0) h = ChainSumLds(row, m_pad, h);
  }
  if (lane == 0) histogram[b] = h;
}

__global__ void SliceBeginKernel(const unsigned* __restrict__ sorted_keys, int n,
                                 int* __restrict__ is_head) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  is_head[p] = (p == 0 || sorted_keys[p] != sorted_keys[p - 1]) ? 1 : 0;
}
__global__ void SliceBeginScatterKernel(const int* __restrict__ is_head,
                                        const int* __restrict__ slice_rank, int n,
                                        int* __restrict__ slice_begin) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n || !is_head[p]) return;
  slice_begin[slice_rank[p]] = p;
}

}  // namespace

// Stable sort of n (32-bit key, index) pairs on the workspace's stream (hipCUB radix sort; the
// only translation unit that pays for instantiating it): grid_3d.hip orders the returns of a scan
// by intensity-grid cell with it.  Scratch comes from ws.dev[temp_slot].
void StableSortPairs32(Workspace& ws, int temp_slot, const unsigned* keys_in, unsigned* keys_out,
                       const int* values_in, int* values_out, int n) {
  size_t bytes = 0;
  CMX_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, keys_in, keys_out, values_in,
                                             values_out, n, 0, 32, ws.stream));
  void* temp = ws.dev[temp_slot].Reserve(bytes + 256);
  CMX_HIP(hipcub::DeviceRadixSort::SortPairs(temp, bytes, keys_in, keys_out, values_in, values_out,
                                             n, 0, 32, ws.stream));
}

}  // namespace cmx

using cmx::Guard;

extern "C" {

cmx_status cmx_voxel_filter(const float* point_cloud_xyz, int32_t num_points, float resolution,
                            int32_t device, float* filtered_xyz, int32_t* num_filtered) {
  return Guard([&] {
    CMX_REQUIRE(num_points >= 0 && num_points <= (1 << 26) && filtered_xyz && num_filtered &&
                    (point_cloud_xyz || num_points == 0),
                "bad argument");
    CMX_REQUIRE(resolution > 0.f, "resolution must be > 0");
    *num_filtered = 0;
    if (num_points == 0) return;
    if (cmx::UseSmallFilter(num_points)) {
      cmx::SmallFilter(0, point_cloud_xyz, num_points, resolution, 0.f, 0.f, device, filtered_xyz,
                       nullptr, num_filtered);
      return;
    }
    cmx::WorkspaceLease ws(device);
    const int n = num_points;
    float* d_xyz = ws->dev[0].ReserveAs<float>(6 * static_cast<size_t>(n));
    float* d_out = d_xyz + 3 * static_cast<size_t>(n);
    CMX_HIP(hipMemcpyAsync(d_xyz, point_cloud_xyz, 12 * static_cast<size_t>(n),
                           hipMemcpyHostToDevice, ws->stream));
    const cmx::FilterScratch s = cmx::Carve(*ws, n);
    const int kept = cmx::VoxelFilterFlags(*ws, s, d_xyz, n, resolution);
    cmx::Compact(*ws, s, d_xyz, n, d_out);
    CMX_HIP(hipMemcpyAsync(filtered_xyz, d_out, 12 * static_cast<size_t>(kept),
                           hipMemcpyDeviceToHost, ws->stream));
    CMX_HIP(hipStreamSynchronize(ws->stream));
    *num_filtered = kept;
  });
}

cmx_status cmx_voxel_filter_indices(const float* point_cloud_xyz, int32_t num_points,
                                    float resolution, int32_t device, int32_t* kept_indices,
                                    int32_t* num_filtered) {
  return Guard([&] {
    CMX_REQUIRE(num_points >= 0 && num_points <= (1 << 26) && kept_indices && num_filtered &&
                    (point_cloud_xyz || num_points == 0),
                "bad argument");
    CMX_REQUIRE(resolution > 0.f, "resolution must be > 0");
    *num_filtered = 0;
    if (num_points == 0) return;
    if (cmx::UseSmallFilter(num_points)) {
      cmx::SmallFilter(0, point_cloud_xyz, num_points, resolution, 0.f, 0.f, device, nullptr,
                       kept_indices, num_filtered);
      return;
    }
    cmx::WorkspaceLease ws(device);
    const int n = num_points;
    // dev[0]: the cloud | the kept indices
    float* d_xyz = ws->dev[0].ReserveAs<float>(4 * static_cast<size_t>(n));
    int* d_out = reinterpret_cast<int*>(d_xyz + 3 * static_cast<size_t>(n));
    CMX_HIP(hipMemcpyAsync(d_xyz, point_cloud_xyz, 12 * static_cast<size_t>(n),
                           hipMemcpyHostToDevice, ws->stream));
    const cmx::FilterScratch s = cmx::Carve(*ws, n);
    const int kept = cmx::VoxelFilterFlags(*ws, s, d_xyz, n, resolution);
    cmx::CompactIndexKernel<<<cmx::DivUp(n, 256), 256, 0, ws->stream>>>(s.used, s.offset, n, d_out);
    CMX_HIP(hipMemcpyAsync(kept_indices, d_out, 4 * static_cast<size_t>(kept),
                           hipMemcpyDeviceToHost, ws->stream));
    CMX_HIP(hipStreamSynchronize(ws->stream));
    *num_filtered = kept;
  });
}

namespace cmx {
namespace {
// sensor::AdaptiveVoxelFilter; the kept points (`filtered_xyz`, may be null) and / or their
// indices in the input (`kept_indices`, may be null), ascending.
void AdaptiveVoxelFilterImpl(const float* point_cloud_xyz, int32_t num_points, float max_length,
                             float min_num_points, float max_range, int32_t device,
                             float* filtered_xyz, int32_t* kept_indices, int32_t* num_filtered) {
  CMX_REQUIRE(num_points >= 0 && num_points <= (1 << 26) && (filtered_xyz || kept_indices) &&
                  num_filtered && (point_cloud_xyz || num_points == 0),
              "bad argument");
  CMX_REQUIRE(max_length > 0.f, "max_length must be > 0");
  *num_filtered = 0;
  if (num_points == 0) return;
  if (UseSmallFilter(num_points)) {
    SmallFilter(1, point_cloud_xyz, num_points, max_length, min_num_points, max_range, device,
                filtered_xyz, kept_indices, num_filtered);
    return;
  }
  WorkspaceLease ws(device);
  const int n0 = num_points;
  // dev[0]: raw cloud | in-range cloud | result | candidate | the same three as index lists.
  float* d_raw = ws->dev[0].ReserveAs<float>(15 * static_cast<size_t>(n0));
  float* d_in = d_raw + 3 * static_cast<size_t>(n0);
  float* d_result = d_in + 3 * static_cast<size_t>(n0);
  float* d_candidate = d_result + 3 * static_cast<size_t>(n0);
  int* i_in = reinterpret_cast<int*>(d_candidate + 3 * static_cast<size_t>(n0));
  int* i_result = i_in + n0;
  int* i_candidate = i_result + n0;
  const bool indices = kept_indices != nullptr;
  hipStream_t st = ws->stream;
  CMX_HIP(hipMemcpyAsync(d_raw, point_cloud_xyz, 12 * static_cast<size_t>(n0),
                         hipMemcpyHostToDevice, st));
  const FilterScratch s = Carve(*ws, n0);
  // FilterByMaxRange.
  RangeFlagKernel<<<DivUp(n0, 256), 256, 0, st>>>(d_raw, n0, max_range, s.used);
  size_t bytes = s.temp_bytes;
  CMX_HIP(hipcub::DeviceScan::ExclusiveSum(s.temp, bytes, s.used, s.offset, n0, st));
  Compact(*ws, s, d_raw, n0, d_in);
  if (indices) CompactIndexKernel<<<DivUp(n0, 256), 256, 0, st>>>(s.used, s.offset, n0, i_in);
  int last_used = 0, last_offset = 0;
  CMX_HIP(hipMemcpyAsync(&last_used, s.used + (n0 - 1), 4, hipMemcpyDeviceToHost, st));
  CMX_HIP(hipMemcpyAsync(&last_offset, s.offset + (n0 - 1), 4, hipMemcpyDeviceToHost, st));
  CMX_HIP(hipStreamSynchronize(st));
  const int n = last_used + last_offset;
  const float* d_final = d_in;
  const int* i_final = i_in;
  int kept = n;
  // AdaptivelyVoxelFiltered (voxel_filter.cc:38-75): the same sequence of VoxelFilter calls.
  const auto filter = [&](float length, float* d_out, int* i_out) {
    const int m = VoxelFilterFlags(*ws, s, d_in, n, length);
    Compact(*ws, s, d_in, n, d_out);
    if (indices && n > 0)
      CompactMappedIndexKernel<<<DivUp(n, 256), 256, 0, st>>>(s.used, s.offset, n, i_in, i_out);
    return m;
  };
  if (!(static_cast<float>(n) <= min_num_points)) {
    bool done = false;
    kept = filter(max_length, d_result, i_result);
    d_final = d_result;
    i_final = i_result;
    if (static_cast<float>(kept) >= min_num_points) done = true;
    for (float high_length = max_length; !done && high_length > 1e-2f * max_length;
         high_length /= 2.f) {
      float low_length = high_length / 2.f;
      kept = filter(low_length, d_result, i_result);
      if (static_cast<float>(kept) >= min_num_points) {
        while ((high_length - low_length) / low_length > 1e-1f) {
          const float mid_length = (low_length + high_length) / 2.f;
          const int m = filter(mid_length, d_candidate, i_candidate);
          if (static_cast<float>(m) >= min_num_points) {
            low_length = mid_length;
            std::swap(d_result, d_candidate);
            std::swap(i_result, i_candidate);
            d_final = d_result;
            i_final = i_result;
            kept = m;
          } else {
            high_length = mid_length;
          }
        }
        done = true;
      }
    }
  }
  if (filtered_xyz)
    CMX_HIP(hipMemcpyAsync(filtered_xyz, d_final, 12 * static_cast<size_t>(kept),
                           hipMemcpyDeviceToHost, st));
  if (indices)
    CMX_HIP(hipMemcpyAsync(kept_indices, i_final, 4 * static_cast<size_t>(kept),
                           hipMemcpyDeviceToHost, st));
  CMX_HIP(hipStreamSynchronize(st));
  *num_filtered = kept;
}
}  // namespace
}  // namespace cmx

cmx_status cmx_adaptive_voxel_filter(const float* point_cloud_xyz, int32_t num_points,
                                     float max_length, float min_num_points, float max_range,
                                     int32_t device, float* filtered_xyz, int32_t* num_filtered) {
  return Guard([&] {
    CMX_REQUIRE(filtered_xyz != nullptr, "bad argument");
    cmx::AdaptiveVoxelFilterImpl(point_cloud_xyz, num_points, max_length, min_num_points,
                                 max_range, device, filtered_xyz, nullptr, num_filtered);
  });
}

cmx_status cmx_adaptive_voxel_filter_indices(const float* point_cloud_xyz, int32_t num_points,
                                             float max_length, float min_num_points,
                                             float max_range, int32_t device,
                                             int32_t* kept_indices, int32_t* num_filtered) {
  return Guard([&] {
    CMX_REQUIRE(kept_indices != nullptr, "bad argument");
    cmx::AdaptiveVoxelFilterImpl(point_cloud_xyz, num_points, max_length, min_num_points,
                                 max_range, device, nullptr, kept_indices, num_filtered);
  });
}

cmx_status cmx_compute_histogram(const float* point_cloud_xyz, int32_t num_points,
                                 int32_t histogram_size, int32_t device, float* histogram) {
  return Guard([&] {
    CMX_REQUIRE(num_points >= 0 && num_points <= (1 << 24) && histogram && histogram_size >= 1 &&
                    histogram_size <= 8192 && (point_cloud_xyz || num_points == 0),
                "bad argument");
    for (int b = 0; b < histogram_size; ++b) histogram[b] = 0.f;
    cmx::WorkspaceLease ws(device);
    if (num_points == 0) return;
    const int n = num_points;
    hipStream_t st = ws->stream;
    const size_t N = static_cast<size_t>(n);
    float* d_xyz = ws->dev[0].ReserveAs<float>(3 * N);
    CMX_HIP(hipMemcpyAsync(d_xyz, point_cloud_xyz, 12 * N, hipMemcpyHostToDevice, st));
    // keys(u32) x2 | index x2 | keys2(u64) x2 | index2 x2 | is_head | slice_rank | slice_begin |
    // bucket | value | histogram | temp
    size_t sort32 = 0, sort64 = 0, scan = 0;
    unsigned *k = nullptr, *v = nullptr;
    unsigned long long* k2 = nullptr;
    int* ip = nullptr;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, sort32, k, k, v, v, n, 0, 32, st);
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, sort64, k2, k2, v, v, n, 0, 64, st);
    (void)hipcub::DeviceScan::InclusiveSum(nullptr, scan, ip, ip, n, st);
    const size_t temp_bytes = std::max(sort32, std::max(sort64, scan)) + 256;
    char* base = static_cast<char*>(ws->dev[9].Reserve(
        N * (4 * 4 + 2 * 8 + 2 * 4 + 5 * 4) + 4 * static_cast<size_t>(histogram_size) + 1024 + temp_bytes));
    const auto take = [&](size_t bytes) { char* p = base; base += (bytes + 15) & ~size_t(15); return p; };
    unsigned long long* keys2 = reinterpret_cast<unsigned long long*>(take(8 * N));
    unsigned long long* keys2_sorted = reinterpret_cast<unsigned long long*>(take(8 * N));
    unsigned* keys = reinterpret_cast<unsigned*>(take(4 * N));
    unsigned* keys_sorted = reinterpret_cast<unsigned*>(take(4 * N));
    unsigned* index = reinterpret_cast<unsigned*>(take(4 * N));
    unsigned* index_sorted = reinterpret_cast<unsigned*>(take(4 * N));
    unsigned* index2 = reinterpret_cast<unsigned*>(take(4 * N));
    unsigned* index2_sorted = reinterpret_cast<unsigned*>(take(4 * N));
    int* is_head = reinterpret_cast<int*>(take(4 * N));
    int* slice_rank = reinterpret_cast<int*>(take(4 * N));
    int* slice_begin = reinterpret_cast<int*>(take(4 * N));
    float* d_hist = reinterpret_cast<float*>(take(4 * static_cast<size_t>(histogram_size)));
    base = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(base) + 255) & ~uintptr_t(255));
    void* temp = base;
    const int blocks = cmx::DivUp(n, 256);
    cmx::SliceKeyKernel<<<blocks, 256, 0, st>>>(d_xyz, n, keys, index);
    size_t bytes = temp_bytes;
    CMX_HIP(hipcub::DeviceRadixSort::SortPairs(temp, bytes, keys, keys_sorted, index, index_sorted,
                                               n, 0, 32, st));
    cmx::SliceBeginKernel<<<blocks, 256, 0, st>>>(keys_sorted, n, is_head);
    bytes = temp_bytes;
    CMX_HIP(hipcub::DeviceScan::InclusiveSum(temp, bytes, is_head, slice_rank, n, st));
    int num_slices = 0;
    CMX_HIP(hipMemcpyAsync(&num_slices, slice_rank + (n - 1), 4, hipMemcpyDeviceToHost, st));
    CMX_HIP(hipStreamSynchronize(st));
    // slice_rank is 1-based after the inclusive sum: shift by one through the pointer.
    cmx::SliceBeginScatterKernel<<<blocks, 256, 0, st>>>(is_head, slice_rank, n, slice_begin - 1);
    cmx::SliceAngleKernel<<<num_slices, 256, 0, st>>>(d_xyz, index_sorted, slice_begin, num_slices,
                                                      n, keys2, index2);
    bytes = temp_bytes;
    CMX_HIP(hipcub::DeviceRadixSort::SortPairs(temp, bytes, keys2, keys2_sorted, index2,
                                               index2_sorted, n, 0, 64, st));
    // votes in (slice, angle) order, then stably by bucket (the first sort's buffers are free)
    unsigned* vote_bucket = keys;
    unsigned* vote_value = index;
    unsigned* vote_bucket_sorted = keys_sorted;
    unsigned* vote_value_sorted = index_sorted;
    // distance < kMin <=> d2 < min_sq, distance > kMax <=> d2 > max_sq for the correctly rounded
    // f32 square root of the reference (monotone): the smallest d2 whose root reaches kMin, the
    // largest whose root does not exceed kMax.
    static const float min_sq = [] {
      float x = cmx::kMinDistance * cmx::kMinDistance;
      while (std::sqrt(x) >= cmx::kMinDistance) x = std::nextafter(x, 0.f);
      while (std::sqrt(x) < cmx::kMinDistance) x = std::nextafter(x, 1.f);
      return x;
    }();
    static const float max_sq = [] {
      float x = cmx::kMaxDistance * cmx::kMaxDistance;
      while (std::sqrt(x) <= cmx::kMaxDistance) x = std::nextafter(x, 2.f);
      while (std::sqrt(x) > cmx::kMaxDistance) x = std::nextafter(x, 0.f);
      return x;
    }();
    cmx::SliceWalkKernel<<<num_slices, 256, 0, st>>>(d_xyz, keys2_sorted, index2_sorted,
                                                     slice_begin, num_slices, n, histogram_size,
                                                     min_sq, max_sq, vote_bucket, vote_value);
    bytes = temp_bytes;
    CMX_HIP(hipcub::DeviceRadixSort::SortPairs(temp, bytes, vote_bucket, vote_bucket_sorted,
                                               vote_value, vote_value_sorted, n, 0, 16, st));
    cmx::HistogramChainKernel<<<histogram_size, 64, 0, st>>>(
        vote_bucket_sorted, vote_value_sorted, n, histogram_size, d_hist);
    CMX_HIP(hipGetLastError());
    CMX_HIP(hipMemcpyAsync(histogram, d_hist, 4 * static_cast<size_t>(histogram_size),
                           hipMemcpyDeviceToHost, st));
    CMX_HIP(hipStreamSynchronize(st));
  });
}

}  // extern "C"
