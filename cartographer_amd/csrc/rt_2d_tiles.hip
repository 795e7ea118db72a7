// RealTimeCorrelativeScanMatcher2D::Match on probability grids: the integer bulk pass out of LDS
// TILES of the grid (round 4), then exact finalists.
//
// Reference: SM2/real_time_correlative_scan_matcher_2d.cc:61-75 (ComputeCandidateScore),
// :83-115 (GenerateExhaustiveSearchCandidates), :117-149 (Match), :151-176 (ScoreCandidates).
//
// The reference's score of a candidate is mean_p P(cell_p + d) summed in f32 in point order
// (:61-75), and P is affine in the stored uint16: P = 0.1 + u * kScale with u = 32767 - value
// (0 for unknown / outside).  Integer sums of u are exact and order-free, so the bulk of the
// search needs neither the f32 chain nor one gather per (candidate, point):
//   * cells are quantised to q = u >> kQShift (10 bits) in 16-bit fields; all (2 nl + 1)^2
//     candidates of a rotation read, for one point, a (2 nl + 1)^2 window of cells, four
//     candidates per ds_read_b64 (rt_2d_device.h, RowPairAccumulate);
//   * every candidate whose weighted upper bound reaches the best weighted lower bound is
//     re-summed with the EXACT integers (wave-parallel, order-free), which leaves the handful
//     within the rounding of the f32 chain of the best: those repeat the reference's sequential
//     f32 sum, and the host applies the libm weight and the first-maximum rule.
//
// Until round 3 a workgroup staged the WHOLE grid (plus halo) in LDS: 110 KB for 200 x 200, one
// workgroup per CU, nothing at all for the 400 x 400 the reference's active submap grows to
// (mapping/2d/submap_2d.cc:194, grid_2d.cc:130-164), and the preparation (rotate, discretise,
// sort by phase) ran inside that one-workgroup-per-CU kernel.  Now:
//   Rt2DQuantKernel     the grid as a quantised image with a zero halo in HBM, once per grid
//                       VERSION (cached with a resident cmx_grid2d; built inside the call when
//                       the grid has changed -- the real caller inserts a scan after every match);
//   Rt2DTilePrepKernel  one workgroup per (match, rotation) at full occupancy: discretise, bin by
//                       TILE of the scan's bounding box and by phase, lists to HBM (L2);
//   Rt2DTileKernel      one workgroup per (match, tile, rotation group): the tile's image enters
//                       LDS by gather-DMA (global_load_lds_dwordx4 with per-lane row addresses),
//                       its lists by plain copies, then the window tasks; two or more workgroups
//                       share a CU, so one computes while another stages.  Sums of a match's
//                       tiles meet in HBM by integer atomics;
//   Rt2DFinishKernel    one workgroup per match: bounds, exact integer re-sums, f32 finalists.
// Grids of any size stay on this path: the image a workgroup holds depends on the tile size, not
// on the grid.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <string>
#include <unordered_map>

#include "rt_2d_device.h"
#include "scan_matching_2d.h"

namespace cmx {
namespace {

constexpr int kTileMaxThreads = 1024;        // a tile workgroup: 1024 threads alone on a CU (one tile
                                            // per match), or 512 with two per CU (several tiles)
constexpr int kFinishThreads = 512;
#ifndef CMX_RT2D_JOB_CHUNKS
#define CMX_RT2D_JOB_CHUNKS 8
#endif
constexpr int kJobChunks = CMX_RT2D_JOB_CHUNKS;                // chunks of 64 points of one rotation a wavefront discretises in one job
constexpr int kFusedMaxPoints = 2048;        // (clouds beyond: the prep kernel, more rotations per workgroup)
constexpr int kMaxTiles = 16;               // tiles of a match's bounding box (x 4 phases = 64 keys)
constexpr int kStage1Cap = 1024;            // candidates the exact integer pass takes per match
constexpr unsigned kFlat = 0xffffffffu;     // misc[1]: more candidates than the lists hold
constexpr unsigned kOutOfBox = 0x80000000u; // misc[0] (next to the prep tickets): a point fell outside the predicted box
constexpr unsigned kBoundFlat = 0x20000000u;       // misc[0]: more blocks reach the bound than the bound kernel lists (flat landscape)
constexpr unsigned kBoundViolated = 0x40000000u;   // misc[0], debug switch rt2d_bounds_verify: a block bound below one of its candidates

struct Rt2DTileParams {
  // grid and initial pose
  const uint16_t* cells;
  int nx, ny;
  double res, max_x, max_y, inv_res;
  float tx, ty, init_qw, init_qz;
  int nl, num_scans, num_angular;
  double step, wt, wr;
  const float2* scan_rot;
  const float* xyz;
  int n, n_pad;
  // window geometry (rt_2d_device.h): blocks per window row, rows per half-wave, rows per lane
  int B, H, rpl;
  int hl, ht;                // image (X, Y) = grid (X - hl, Y - ht)
  // quantised image of the whole grid in HBM
  uint16_t* qimage;
  int gpitch;                // bytes per image row (multiple of 16)
  int grows;
  int image_build;           // 1: Rt2DQuantKernel fills qimage in this call
  // tiles of the scan's bounding box (window-start coordinates)
  int box_x0, box_y0;        // of tile (0, 0); box_x0 is a multiple of 8
  int T;                     // tile core: window starts [k T, (k + 1) T) per axis (multiple of 8)
  unsigned T_magic;          // ceil(2^32 / T)
  int ntx, nty;
  int lp;                    // LDS row pitch in bytes (multiple of 16, conflict-free)
  int th_img;                // image rows of a tile: T + rpl * H
  int tile_image_bytes;      // (th_img + rpl * H null rows) * lp, whole KiB
  int null_addr;             // th_img * lp: the all-zero block padding slots read
  // lists (HBM scratch): per rotation cap_s u16 entries, keys (tile, phase) in order
  uint16_t* lists;
  int cap_s;
  uint32_t* hdr;             // [num_scans][ntiles * 4]: start | count << 16
  // rotation groups: work item (tile, g, G) takes rotations g, g + G, ...; G is chosen per TILE
  // on the device from its entry count (the planner at the end of the prep kernel)
  int gmin, gmax;            // ceil(num_scans / 64) <= G <= gmax
  int target;                // entries per work item the planner aims at
  int rw;                    // rotations a workgroup's LDS holds: ceil(num_scans / gmin) <= 64
  int list_lds;              // u16 entries of the LDS list buffer
  int task_cap;
  int flush_atomic;          // more than one tile: sums meet by atomics (qsum zeroed by the prep)
  int fused;                 // one tile per match: the tile kernel discretises its rotations itself
                             // (no prep kernel, no lists in HBM, work items listed by the host)
  int* qsum;                 // [num_scans][side^2]
  unsigned* misc;            // [0] prep workgroups done (the last one plans), [1] finalist count, then pairs
  unsigned* overflow;        // finalist pairs beyond kFinalistHead
  unsigned* stage;           // [0] candidates of the exact integer pass, [1] f32 finalists | rotations with finalists << 16
  unsigned long long* timeline;   // debug switch `timeline`: 16 stamps per tile / finish workgroup, else null
  int timeline_finish_base;       // first slot of the finish kernel's workgroups
  // block bounds (rt_2d_bounds.h): the four parity planes of the 2 x 2 max-pooled image behind
  // the quantised image (null: window beyond 16 x 16), and this match's LDS copy of them
  const uint8_t* m2;
  int m2_pitch, m2_rows;     // bytes per plane row (multiple of 4), rows per plane
  int b_c0, b_r0;            // first plane column (multiple of 4) / row of the LDS copy
  int b_lpb, b_lh;           // its pitch in bytes (multiple of 4) and rows
  unsigned* bstat;           // misc + 124: [0] blocks summed exactly, [1] block bounds evaluated
  int b_verify;              // debug switch rt2d_bounds_verify
  int b_ub_at;               // this match's block sums in the launch's HBM scratch (rotation groups > 1)
  int b_tail_at;             // LDS offset of rotations | block sums | list | sums (behind the fused finish's region)
  // blocks of 4 x 4 translations (round 6): the sixteen phase planes of the 4 x 4 max-pooled image
  // behind the 2 x 2 ones, and this match's LDS copy of them
  const uint8_t* m4;
  int m4_pitch, m4_rows;     // bytes per plane row (multiple of 4), rows per plane
  int b4_c0, b4_r0;          // first plane column (multiple of 4) / row of the LDS copy
  int b4_lpb, b4_lh;         // its pitch in bytes (an odd number of 8-byte pieces) and rows
  int b4_pstride;            // LDS bytes from plane to plane (= 8 mod 128: two banks apart)
  int b4_tail_at;            // LDS offset of rotations | block sums
  int b4_ub_at;              // this match's 4 x 4 block sums in the launch's HBM scratch
  // the image once more as BYTES, q8 = u >> 7, rows of gpitch / 2 bytes behind the planes: the tail
  // kernel holds the match's box of it in LDS (b8_lh rows of b8_lp bytes from (box_x0, box_y0) on)
  uint8_t* q8;
  int b8_lp, b8_lh;
};

// A barrier for data that travels through LDS only.  __syncthreads() carries a workgroup-scope
// fence, i.e. s_waitcnt vmcnt(0): with an LDS-DMA copy in flight it would wait for the copy
// (cdna_hip_programming.md, "glds in flight across the barrier").
__device__ __forceinline__ void LdsBarrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// A match's parameters into LDS, a dword per thread (callers: barrier before the first use).
__device__ __forceinline__ void CopyParams(Rt2DTileParams* dst, const Rt2DTileParams* src, int tid) {
  static_assert(sizeof(Rt2DTileParams) % 4 == 0 && sizeof(Rt2DTileParams) / 4 <= 256,
                "one dword per thread of the smallest workgroup");
  if (tid < static_cast<int>(sizeof(Rt2DTileParams) / 4))
    reinterpret_cast<uint32_t*>(dst)[tid] = AsGlobal(reinterpret_cast<const uint32_t*>(src))[tid];
}

__device__ __forceinline__ Rt2DFrame FrameOf(const Rt2DTileParams& P) {
  return Rt2DFrame{P.res, P.inv_res, P.max_x, P.max_y, P.tx, P.ty, P.init_qw, P.init_qz,
                   P.nx, P.ny, P.nl};
}

// exp(-(hypot(x, y) w_t + |theta| w_r)^2) in f32 (relative error ~1e-6): only used for bounds,
// which carry 1e-5 of relative slack on top; the returned score is weighted on the host with libm.
__device__ __forceinline__ float TileWeight(const Rt2DTileParams& P, int s, int dx, int dy) {
  const float res = static_cast<float>(P.res);
  const float cx = -dy * res, cy = -dx * res;
  const float theta = static_cast<float>((s - P.num_angular) * P.step);
  const float t = sqrtf(cx * cx + cy * cy) * static_cast<float>(P.wt) +
                  fabsf(theta) * static_cast<float>(P.wr);
  return __expf(-(t * t));
}

// ---------------------------------------------------------------------------------------------
// grid (ceil(grows * gpitch / 16 / 256), items): the quantised image, 8 cells per thread.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
Rt2DQuantKernel(const Rt2DTileParams* __restrict__ params) {
  const Rt2DTileParams& P = params[blockIdx.y];
  const int v = blockIdx.x * 256 + threadIdx.x;
  if (!P.image_build || v >= ((P.grows * P.gpitch) >> 4)) return;
  const int byte = v << 4;
  const int Y = byte / P.gpitch, X0 = (byte - Y * P.gpitch) >> 1;
  const auto* cells = AsGlobal(P.cells);
  const int gy = Y - P.ht;
  unsigned q[8];
  unsigned long long bytes = 0;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int gx = X0 + c - P.hl;
    unsigned val = 0;
    if (static_cast<unsigned>(gx) < static_cast<unsigned>(P.nx) &&
        static_cast<unsigned>(gy) < static_cast<unsigned>(P.ny)) {
      const unsigned raw = cells[gy * P.nx + gx] & 32767u;
      val = raw ? (32767u - raw) >> kQShift : 0u;
    }
    q[c] = val;
    bytes |= static_cast<unsigned long long>(val >> (kQ8Shift - kQShift)) << (8 * c);
  }
  if (P.q8) reinterpret_cast<unsigned long long*>(P.q8)[v] = bytes;
  reinterpret_cast<uint4*>(P.qimage)[v] = make_uint4(q[0] | (q[1] << 16), q[2] | (q[3] << 16),
                                                     q[4] | (q[5] << 16), q[6] | (q[7] << 16));
}

// ---------------------------------------------------------------------------------------------
// grid (max rotations, matches), 256 threads: one rotation of one match.
// Dynamic LDS: tmp[n_pad] u32 (entry | key << 16) | loc[n_pad] u16 | cnt[64] | start[64].
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
Rt2DTilePrepKernel(const Rt2DTileParams* __restrict__ params, int* __restrict__ work_count,
                   int4* __restrict__ work_items, int work_cap) {
  extern __shared__ __attribute__((aligned(16))) unsigned char prep_smem[];
  // (the match's parameters through LDS: ONE round of loads instead of a scalar-load round trip
  // whenever the code reaches for another field)
  __shared__ Rt2DTileParams P;
  const int tid = threadIdx.x, lane = tid & 63;
  CopyParams(&P, params + blockIdx.y, tid);
  __syncthreads();
  const int s = blockIdx.x;
  if (s >= P.num_scans) return;
  const int n = P.n, n_pad = P.n_pad;
  uint32_t* tmp = reinterpret_cast<uint32_t*>(prep_smem);
  uint16_t* loc = reinterpret_cast<uint16_t*>(tmp + n_pad);
  int* cnt = reinterpret_cast<int*>(loc + n_pad);
  int* start = cnt + 64;
  if (tid < 64) cnt[tid] = 0;
  const int side = 2 * P.nl + 1, cands = side * side;
  if (P.flush_atomic)
    for (int e = tid; e < cands; e += 256) P.qsum[static_cast<size_t>(s) * cands + e] = 0;
  __syncthreads();
  const Rt2DFrame F = FrameOf(P);
  const float2 rot = P.scan_rot[s];
  const auto* xyz = AsGlobal(P.xyz);
  const int nkeys = P.ntx * P.nty * 4;
  bool outside = false;
  for (int i = tid; i < n; i += 256) {
    int ix, iy;
    Rt2DCellOf(F, rot.x, rot.y, xyz[3 * i], xyz[3 * i + 1], &ix, &iy);
    // window start in image coordinates, relative to the box of this match
    const int rx = ix - P.nl + P.hl - P.box_x0, ry = iy - P.nl + P.ht - P.box_y0;
    const unsigned tX = __umulhi(static_cast<unsigned>(max(rx, 0)), P.T_magic);
    const unsigned tY = __umulhi(static_cast<unsigned>(max(ry, 0)), P.T_magic);
    uint32_t packed = 0xffffffffu;
    if (rx < 0 || ry < 0 || tX >= static_cast<unsigned>(P.ntx) ||
        tY >= static_cast<unsigned>(P.nty)) {
      outside = true;                          // (the host's box is conservative: never expected)
    } else {
      const int lx = rx - static_cast<int>(tX) * P.T, ly = ry - static_cast<int>(tY) * P.T;
      const int entry = (ly * P.lp + (lx & ~3) * 2) >> 3;
      const int key = (static_cast<int>(tY) * P.ntx + static_cast<int>(tX)) * 4 + (lx & 3);
      loc[i] = static_cast<uint16_t>(atomicAdd(&cnt[key], 1));
      packed = static_cast<uint32_t>(entry) | (static_cast<uint32_t>(key) << 16);
    }
    tmp[i] = packed;
  }
  if (outside) atomicOr(&P.misc[0], kOutOfBox);
  __syncthreads();
  if (tid < 64) {
    // list starts: every key's list is padded to 16 entries (a group of the window update)
    const int c = lane < nkeys ? cnt[lane] : 0;
    const int padded = (c + 15) & ~15;
    const int incl = WaveInclusiveScan(padded);
    start[lane] = incl - padded;
    // (agent-scope atomic store: written through, so that the planner below -- another
    // workgroup, possibly on another XCD -- reads it without a device-wide fence here: an
    // agent-scope release writes back the whole L2, 400 us for the 3456 workgroups of a batch)
    if (lane < nkeys)
      __hip_atomic_store(&P.hdr[static_cast<size_t>(s) * nkeys + lane],
                         static_cast<uint32_t>(incl - padded) | (static_cast<uint32_t>(c) << 16),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  uint16_t* out = P.lists + static_cast<size_t>(s) * P.cap_s;
  for (int i = tid; i < n; i += 256) {
    const uint32_t packed = tmp[i];
    if (packed == 0xffffffffu) continue;
    out[start[packed >> 16] + loc[i]] = static_cast<uint16_t>(packed & 0xffffu);
  }
  // ---- the planner: the LAST workgroup of a match to arrive here cuts the match into work
  // items.  A tile's entries are spread over as many rotation groups as its entry count asks for
  // (the points of a scan cluster: one tile of four may hold 80 % of them), an empty tile gets
  // none ----------------------------------------------------------------------------------------
  __shared__ int plan[2 * kMaxTiles + 4];
  __syncthreads();                           // this workgroup's header words have been written
  if (tid == 0)                              // through (the barrier waits for its stores) before its
    plan[0] = static_cast<int>(__hip_atomic_fetch_add(&P.misc[0], 1u, __ATOMIC_RELAXED,   // ticket
                                                      __HIP_MEMORY_SCOPE_AGENT));
  __syncthreads();
  if ((plan[0] & 0x7fffffff) != P.num_scans - 1) return;
  const int ntiles = P.ntx * P.nty;
  int* total = plan + 4;                     // [ntiles] entries, then [ntiles] first item
  if (tid < 2 * kMaxTiles) total[tid] = 0;
  __syncthreads();
  for (int idx = tid; idx < P.num_scans * nkeys; idx += 256) {
    const uint32_t h = __hip_atomic_load(&P.hdr[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (h >> 16) atomicAdd(&total[(idx % nkeys) >> 2], static_cast<int>(h >> 16));
  }
  __syncthreads();
  if (tid == 0) {
    int items = 0;
    for (int t = 0; t < ntiles; ++t) {
      const int entries = total[t];
      int G = entries == 0 ? 0 : max(1, (entries + P.target / 2) / P.target);   // (to nearest)
      if (G) G = min(max(G, P.gmin), min(P.gmax, P.num_scans));
      if (!P.flush_atomic) G = max(G, P.gmin);           // (a single tile writes its sums: always run)
      total[t] = G;
      total[kMaxTiles + t] = items;
      items += G;
    }
    const int base = atomicAdd(work_count, items);
    plan[1] = base;
    plan[2] = items;
    if (base + items > work_cap) atomicOr(&P.misc[0], kOutOfBox);    // (never: host bound)
  }
  __syncthreads();
  const int base = plan[1];
  if (base + plan[2] > work_cap) return;
  for (int t = 0; t < ntiles; ++t)
    for (int g = tid; g < total[t]; g += 256)
      work_items[base + total[kMaxTiles + t] + g] =
          make_int4(static_cast<int>(blockIdx.y), t, g, total[t]);
}

// ---------------------------------------------------------------------------------------------
// One match's candidates from their integer sums to the finalists' f32 scores, by a whole
// workgroup (kThreads threads): the finish kernel -- grid (matches), 512 threads -- or the tile
// kernel's last workgroup of a match (fused path, 1024 threads).  Three stages, each narrower and more exact than the one before:
//   1. bounds from the quantised sums: every candidate whose weighted upper bound reaches the
//      best weighted lower bound (a few dozen);
//   2. those candidates with the EXACT integers, one wavefront per candidate (order-free sums:
//      the points of a rotation are discretised once into LDS): what remains undecided is the
//      rounding of the reference's f32 chain;
//   3. the candidates within that rounding of the best (one or two): the reference's sequential
//      f32 sum (:61-75) -- all threads fetch the probabilities of a finalist's points into LDS,
//      one lane per finalist runs the chain -- left for the host as (index, score bits) pairs.
// Dynamic LDS: cells[n_pad] u32 | prob[group][n_pad + 4] f32 | rot_flag[num_scans] |
//   fin[kStage1Cap] | exact[kStage1Cap] | fin2[kStage1Cap]
// ---------------------------------------------------------------------------------------------
// `cand_e` / `cand_q` (LDS, outside the region this function lays out; or null): the candidates
// the caller has summed and their quantised sums -- the bound kernel, rt_2d_bounds.h: every other
// candidate of the match lies below the best lower bound and is not looked at.
template <int kThreads, bool kCoherent, bool kTimeline, int kShift = kQShift>
__device__ __forceinline__ void Rt2DFinishMatch(Rt2DTileParams& P, unsigned char* fin_smem, int group,
                                                unsigned* __restrict__ host_out, int match,
                                                const int* cand_e = nullptr, const int* cand_q = nullptr,
                                                int cand_count = 0) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int kWaves = kThreads / 64;
  const int side = 2 * P.nl + 1, cands = side * side, n = P.n, n_pad = P.n_pad;
  uint32_t* cellbuf = reinterpret_cast<uint32_t*>(fin_smem);
  float* prob = reinterpret_cast<float*>(cellbuf + n_pad);
  int* rot_flag = reinterpret_cast<int*>(prob + group * (n_pad + 4));
  int* fin = rot_flag + ((P.num_scans + 3) & ~3);
  int* exact = fin + kStage1Cap;
  int* fin2 = exact + kStage1Cap;
  __shared__ unsigned red[kWaves];
  __shared__ int nfin, nfin2;
  __shared__ int sel[18];
  // The match's 128 result words go straight to the caller's pinned buffer (mapped into the
  // device's address space): no copy command, no copy kernel after this one.
  const auto publish = [&]() {
    __syncthreads();
    if (tid < 128)
      host_out[static_cast<size_t>(match) * 128 + tid] =
          __hip_atomic_load(&P.misc[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  // (a match whose prep flagged a point outside the box is finished like any other -- one more
  // round trip at the head of this latency-bound kernel would cost every match 1.5 us -- and the
  // flag rides in misc[0] to the host)
  unsigned long long* const tl = kTimeline ? P.timeline : nullptr;
  const int tl_block = P.timeline_finish_base + match;
  Stamp(tl, tl_block, 0);
  if (tid == 0) { nfin = 0; nfin2 = 0; }
  for (int s = tid; s < P.num_scans; s += kThreads) rot_flag[s] = 0;
  const float kScale = ((1.f - 0.1f) - (1.f - (1.f - 0.1f))) / 32766.f;   // (kMaxCC - kMinCC) / 32766
  const float slack = Rt2DBoundSlack(n);
  // (kShift: how coarsely the caller's sums were quantised -- the bound kernel's tail sums bytes)
  const float per_q = kScale * static_cast<float>(1 << kShift) / static_cast<float>(n);
  const float width = kScale * static_cast<float>((1 << kShift) - 1);
  const int* __restrict__ qsum = P.qsum;
  // (kCoherent: the sums were written by other workgroups of the SAME launch, through to memory)
  const auto load_sum = [&](int at) {
    if constexpr (kCoherent) return __hip_atomic_load(&qsum[at], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return qsum[at];
  };
  // ---- stage 1: the best weighted lower bound, then everyone whose upper bound reaches it.
  // Bounds only SELECT candidates (scores are recomputed exactly), so f32 with slack is enough.
  // A thread owns translations c = tid, tid + 512, ... and walks the rotations: the translation
  // part of the weight's exponent once per c, no division inside the loop, coalesced qsum reads --
  const float res_f = static_cast<float>(P.res);
  const float wt_f = static_cast<float>(P.wt), wr_f = static_cast<float>(P.wr);
  const float step_f = static_cast<float>(P.step);
  const auto weight = [&](float t_translation, int s) {
    const float t = t_translation + fabsf(static_cast<float>(s - P.num_angular) * step_f) * wr_f;
    return __expf(-(t * t));
  };
  float lb_max = 0.f;
  constexpr int kOwn = kThreads > 512 ? 8 : 16;   // candidates a thread keeps in registers (8192 in all)
  const int total = P.num_scans * cands;
  const bool listed = cand_e != nullptr;
  const bool in_registers = !listed && total <= kOwn * kThreads;
  float ub_own[kOwn];
  const int owned = (total + kThreads - 1) / kThreads;      // (uniform)
  // weighted [lower, upper] bound of candidate e from its quantised sum
  const auto bounds_of = [&](int e, int q, float* lower, float* upper) {
    const int s = e / cands, c = e - s * cands;
    const int dxi = c / side, dyi = c - dxi * side;
    const float cx = -(dyi - P.nl) * res_f, cy = -(dxi - P.nl) * res_f;
    const float w = weight(sqrtf(cx * cx + cy * cy) * wt_f, s);
    const float base = 0.1f + per_q * static_cast<float>(q);
    *lower = (base - slack) * w * (1.f - 1e-5f);
    *upper = (base + width + slack) * w * (1.f + 1e-5f);
  };
  if (listed) {
    for (int k = tid; k < cand_count; k += kThreads) {
      float lower, upper;
      bounds_of(cand_e[k], cand_q[k], &lower, &upper);
      lb_max = fmaxf(lb_max, lower);
    }
  } else if (in_registers) {
    // The usual case (C1: 4563 candidates, 9 per thread): ONE round of loads, every upper bound
    // stays in a register until the best lower bound is known.
    int q_own[kOwn];
#pragma unroll
    for (int k = 0; k < kOwn; ++k) {
      const int e = tid + k * kThreads;
      q_own[k] = k < owned && e < total ? load_sum(e) : 0;
    }
    // s = e / cands from an f32 estimate: exact for e < 2^21 (the estimate is off by less than
    // 2^-22 e / cands < 1 / (2 cands), and (e + 0.5) / cands is 1 / (2 cands) away from integers)
    const float inv_cands = 1.f / static_cast<float>(cands), inv_side = 1.f / static_cast<float>(side);
#pragma unroll
    for (int k = 0; k < kOwn; ++k) {
      ub_own[k] = -1.f;
      if (k >= owned) continue;                                          // (uniform)
      const int e = tid + k * kThreads;
      const int s = static_cast<int>((static_cast<float>(e) + 0.5f) * inv_cands);
      const int c = e - s * cands;
      const int dxi = static_cast<int>((static_cast<float>(c) + 0.5f) * inv_side), dyi = c - dxi * side;
      const float cx = -(dyi - P.nl) * res_f, cy = -(dxi - P.nl) * res_f;
      const float w = weight(sqrtf(cx * cx + cy * cy) * wt_f, s);
      const float base = 0.1f + per_q * static_cast<float>(q_own[k]);
      if (e < total) {
        ub_own[k] = (base + width + slack) * w * (1.f + 1e-5f);
        lb_max = fmaxf(lb_max, (base - slack) * w * (1.f - 1e-5f));
      }
    }
  } else {
    for (int c = tid; c < cands; c += kThreads) {
      const int dxi = c / side, dyi = c - dxi * side;
      const float cx = -(dyi - P.nl) * res_f, cy = -(dxi - P.nl) * res_f;
      const float tt = sqrtf(cx * cx + cy * cy) * wt_f;
#pragma unroll 4
      for (int s = 0; s < P.num_scans; ++s) {
        const float base = 0.1f + per_q * static_cast<float>(load_sum(s * cands + c));
        lb_max = fmaxf(lb_max, (base - slack) * weight(tt, s) * (1.f - 1e-5f));
      }
    }
  }
  {
    unsigned bits = __float_as_uint(fmaxf(lb_max, 0.f));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) bits = max(bits, __shfl_xor(bits, off, 64));
    if (lane == 0) red[wave] = bits;
  }
  __syncthreads();
  unsigned best_bits = 0;
#pragma unroll
  for (int w = 0; w < kWaves; ++w) best_bits = max(best_bits, red[w]);
  const float best_lb = __uint_as_float(best_bits);
  if (listed) {
    for (int k = tid; k < cand_count; k += kThreads) {
      float lower, upper;
      const int e = cand_e[k];
      bounds_of(e, cand_q[k], &lower, &upper);
      if (upper >= best_lb) {
        const int at = atomicAdd(&nfin, 1);
        if (at < kStage1Cap) fin[at] = e;
        rot_flag[e / cands] = 1;
      }
    }
  } else if (in_registers) {
    const float inv_cands = 1.f / static_cast<float>(cands);
#pragma unroll
    for (int k = 0; k < kOwn; ++k) {
      if (ub_own[k] >= best_lb) {
        const int e = tid + k * kThreads;
        const int at = atomicAdd(&nfin, 1);
        if (at < kStage1Cap) fin[at] = e;
        rot_flag[static_cast<int>((static_cast<float>(e) + 0.5f) * inv_cands)] = 1;
      }
    }
  } else {
    for (int c = tid; c < cands; c += kThreads) {
      const int dxi = c / side, dyi = c - dxi * side;
      const float cx = -(dyi - P.nl) * res_f, cy = -(dxi - P.nl) * res_f;
      const float tt = sqrtf(cx * cx + cy * cy) * wt_f;
#pragma unroll 4
      for (int s = 0; s < P.num_scans; ++s) {
        const float base = 0.1f + per_q * static_cast<float>(load_sum(s * cands + c));
        if ((base + width + slack) * weight(tt, s) * (1.f + 1e-5f) >= best_lb) {
          const int at = atomicAdd(&nfin, 1);
          if (at < kStage1Cap) fin[at] = s * cands + c;
          rot_flag[s] = 1;
        }
      }
    }
  }
  __syncthreads();
  Stamp(tl, tl_block, 1);                  // stage 1 done
  const int count = nfin;
  if (tid == 0) P.stage[0] = static_cast<unsigned>(count);
  if (count > kStage1Cap) {              // flat landscape: the host repeats the match on the
    if (tid == 0)                        // per-candidate kernels
      __hip_atomic_store(&P.misc[1], kFlat, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    publish();
    return;
  }
  const auto* cells = AsGlobal(P.cells);
  const auto* xyz = AsGlobal(P.xyz);
  const Rt2DFrame F = FrameOf(P);
  const auto discretise = [&](int s) {
    const float2 r = P.scan_rot[s];
    for (int i = tid; i < n; i += kThreads) {
      int ix, iy;
      Rt2DCellOf(F, r.x, r.y, xyz[3 * i], xyz[3 * i + 1], &ix, &iy);
      cellbuf[i] = (static_cast<uint32_t>(ix) & 0xffffu) | (static_cast<uint32_t>(iy) << 16);
    }
  };
  // (a handful of candidates -- the usual case -- go straight to the f32 chain: re-summing them
  // exactly first would only add a pass)
  const bool direct = count <= group;
  if (direct) {
    if (tid < count) fin2[tid] = fin[tid];
    if (tid == 0) {
      nfin2 = count;
      int rotations = 0;
      for (int s = 0; s < P.num_scans; ++s) rotations += rot_flag[s];
      P.stage[1] = static_cast<unsigned>(count) | (static_cast<unsigned>(rotations) << 16);
    }
    __syncthreads();
  } else {
    // ---- stage 2: exact integer sums, a wavefront per candidate -------------------------------
    for (int s = 0; s < P.num_scans; ++s) {
      if (!rot_flag[s]) continue;           // (uniform: LDS value, no writer since the barrier)
      __syncthreads();                      // the previous rotation's cells are done with
      discretise(s);
      __syncthreads();
      for (int j = wave; j < count; j += kWaves) {
        const int e = fin[j];
        if (e / cands != s) continue;       // (wave-uniform)
        const int c = e - s * cands;
        const int dx = c / side - P.nl, dy = c % side - P.nl;
        int sum = 0;
        for (int i0 = 0; i0 < n; i0 += 256) {
          unsigned raw[4];
          bool inside[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {      // four gathers in flight
            const int i = i0 + k * 64 + lane;
            const uint32_t pc = cellbuf[min(i, n - 1)];
            const int x = static_cast<short>(pc & 0xffffu) + dx;
            const int y = static_cast<short>(pc >> 16) + dy;
            inside[k] = i < n && static_cast<unsigned>(x) < static_cast<unsigned>(P.nx) &&
                        static_cast<unsigned>(y) < static_cast<unsigned>(P.ny);
            raw[k] = cells[inside[k] ? P.nx * y + x : 0];     // unconditional load, masked below
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const unsigned v = raw[k] & 32767u;
            sum += (inside[k] && v) ? static_cast<int>(32767u - v) : 0;
          }
        }
        sum = WaveSum(sum);
        if (lane == 0) exact[j] = sum;
      }
    }
    __syncthreads();
    Stamp(tl, tl_block, 2);                  // exact integer sums done
    // ---- bounds from the exact sums: what is left is the rounding of the f32 chain --------------
    const float per_u = kScale / static_cast<float>(n);
    lb_max = 0.f;
    for (int j = tid; j < count; j += kThreads) {
      const int e = fin[j];
      const int s = e / cands, c = e - s * cands;
      const float base = 0.1f + per_u * static_cast<float>(exact[j]);
      const float w = TileWeight(P, s, c / side - P.nl, c % side - P.nl);
      lb_max = fmaxf(lb_max, (base - slack) * w * (1.f - 1e-5f));
    }
    {
      unsigned bits = __float_as_uint(fmaxf(lb_max, 0.f));
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) bits = max(bits, __shfl_xor(bits, off, 64));
      __syncthreads();                      // (red[] of stage 1 has been read by everyone)
      if (lane == 0) red[wave] = bits;
    }
    for (int s = tid; s < P.num_scans; s += kThreads) rot_flag[s] = 0;
    __syncthreads();
    best_bits = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) best_bits = max(best_bits, red[w]);
    const float best_lb2 = __uint_as_float(best_bits);
    for (int j = tid; j < count; j += kThreads) {
      const int e = fin[j];
      const int s = e / cands, c = e - s * cands;
      const float base = 0.1f + per_u * static_cast<float>(exact[j]);
      const float w = TileWeight(P, s, c / side - P.nl, c % side - P.nl);
      if ((base + slack) * w * (1.f + 1e-5f) >= best_lb2) {
        fin2[atomicAdd(&nfin2, 1)] = e;
        rot_flag[s] = 1;
      }
    }
    __syncthreads();
    if (tid == 0) {
      int rotations = 0;
      for (int s = 0; s < P.num_scans; ++s) rotations += rot_flag[s];
      P.stage[1] = static_cast<unsigned>(nfin2) | (static_cast<unsigned>(rotations) << 16);
    }
  }
  const int count2 = nfin2;
  Stamp(tl, tl_block, 3);                  // finalists selected
  // ---- stage 3: the reference's sequential f32 sums ---------------------------------------
  // (rows of n_pad + 4 floats: 16-byte aligned for ds_read_b128, the chain lanes four banks
  // apart; the slots beyond the cloud hold +0, which leaves a positive sum as it is)
  const int row = n_pad + 4;
  for (int s = 0; s < P.num_scans; ++s) {
    if (!rot_flag[s]) continue;           // (uniform)
    __syncthreads();
    discretise(s);
    __syncthreads();
    // this rotation's finalists, `group` (<= 16) at a time: thread 0 picks them from the list
    int next = 0;
    for (;;) {
      if (tid == 0) {
        int gcount = 0;
        for (; next < count2 && gcount < group; ++next) {
          const int e = fin2[next];
          if (e / cands == s) sel[gcount++] = e - s * cands;
        }
        sel[16] = gcount;
        sel[17] = next;
      }
      __syncthreads();
      const int gcount = sel[16];
      next = sel[17];
      if (gcount == 0) break;
      for (int f = 0; f < gcount; ++f) {                  // all threads: one round of gathers
        const int c = sel[f];
        const int dx = c / side - P.nl, dy = c % side - P.nl;
        for (int i0 = 0; i0 < n; i0 += 4 * kThreads) {
          unsigned raw[4];
          bool inside[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int i = i0 + k * kThreads + tid;
            const uint32_t pc = cellbuf[min(i, n - 1)];
            const int x = static_cast<short>(pc & 0xffffu) + dx;
            const int y = static_cast<short>(pc >> 16) + dy;
            inside[k] = static_cast<unsigned>(x) < static_cast<unsigned>(P.nx) &&
                        static_cast<unsigned>(y) < static_cast<unsigned>(P.ny);
            raw[k] = cells[inside[k] ? P.nx * y + x : 0];
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int i = i0 + k * kThreads + tid;
            if (i < n) prob[f * row + i] = inside[k] ? CellProbability(raw[k]) : 0.1f;   // kMinProbability
          }
        }
        for (int i = n + tid; i < n_pad; i += kThreads) prob[f * row + i] = 0.f;
      }
      __syncthreads();
      Stamp(tl, tl_block, 4);              // probabilities of a group of finalists in LDS
      if (tid < gcount) {
        // The reference's sum: N dependent additions in point order (then zeros up to a multiple
        // of 64 points: x + 0 = x).  The values come out of LDS 32 at a time (8 x ds_read_b128),
        // the next 32 on their way while these are added.
        const float sum = ChainSumLds(prob + tid * row, n_pad, 0.f);
        const float score = sum / static_cast<float>(n);
        const int c = sel[tid];
        const int dxi = c / side, dyi = c - dxi * side;
        const int cg = (s * side + dxi) * side + dyi;           // x outer, y inner (:99-113)
        const unsigned at = atomicAdd(&P.misc[1], 1u);
        if (at < static_cast<unsigned>(kFinalistCap)) {
          unsigned* pair = at < static_cast<unsigned>(kFinalistHead)
                               ? P.misc + 2 + 2 * at
                               : P.overflow + 2 * (at - kFinalistHead);
          pair[0] = static_cast<unsigned>(cg);
          pair[1] = __float_as_uint(score);
        }
      }
      __syncthreads();
      Stamp(tl, tl_block, 5);              // chains done
      if (gcount < group) break;
    }
  }
  Stamp(tl, tl_block, 6);
  publish();
}

// ---------------------------------------------------------------------------------------------
// grid (persistent: one workgroup of 1024 threads per CU when every match is ONE tile, else two of
// 512), work items = (match, tile, rotation
// group g, groups G of the tile) from the prep kernel's planner, pulled through a counter.
// Dynamic LDS:
//   image[tile_image_bytes] | acc[rw][side^2] | hdrs[rw][4] | slot[rw + 1] | tasks[task_cap][4] |
//   ctl[16] | list[list_lds] u16
// ---------------------------------------------------------------------------------------------
// (Sixteen wavefronts per CU either way.  Two workgroups of sixteen, registers capped at 64 --
// the compiler takes 80 for two rows per lane -- ran the same batch 1.4x SLOWER, stragglers of
// 54 us among items of 14: profiles/r04_c1_two_workgroups_per_cu.txt.  The other half of the
// CU's wavefront slots is what lets the prep and finish kernels of the other parts of a batch
// run beside this one.)
template <int RPL, int kRowStride, bool kTimeline>
__global__ void __launch_bounds__(kTileMaxThreads)
Rt2DTileKernel(const Rt2DTileParams* __restrict__ params, const int4* __restrict__ work,
               const int* __restrict__ work_count, int* __restrict__ next_item, int work_stride) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tile_smem[];
  // the next work item (x < 0: none).  Planned on the device: (match, tile, g, G).  Fused (listed by
  // the host, work_stride 3): + (cloud address, rotation table address) + (points, rotations) --
  // the loads of the cloud and of the rotations then leave together with those of the parameters
  __shared__ int4 fetched[3];
  __shared__ Rt2DTileParams P;               // the current item's match (one round of loads)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int num_threads = blockDim.x, num_waves = num_threads >> 6;
  const int num_items = *work_count;
  if (tid == 0) {
    const int v = atomicAdd(next_item, 1);
    if (v < num_items) {
      for (int q = 0; q < work_stride; ++q) fetched[q] = work[v * work_stride + q];
    } else {
      fetched[0] = make_int4(-1, 0, 0, 0);
    }
  }
  int item_index = 0;
  for (;; ++item_index) {
  __syncthreads();                           // the previous item's LDS is done with; `fetched` is set
  const int4 item = fetched[0];
  if (item.x < 0) break;
  const int4 aux = fetched[1], aux2 = fetched[2];
  __syncthreads();                           // (everyone has read `fetched` before it is overwritten)
  // the NEXT item: its ticket is drawn now and arrives under this item's image copy; the item
  // itself is read behind the copy's wait and stored at the end
  int ticket = 0;
  int4 next = make_int4(-1, 0, 0, 0);
  if (tid == 0) ticket = atomicAdd(next_item, 1);
  float px = 0.f, py = 0.f;
  float2 rot_mine = make_float2(1.f, 0.f);
  if (work_stride == 3) {
    const auto* xyz = AsGlobal(reinterpret_cast<const float*>(
        static_cast<uintptr_t>(static_cast<uint32_t>(aux.x)) | (static_cast<uintptr_t>(static_cast<uint32_t>(aux.y)) << 32)));
    const auto* rot = AsGlobal(reinterpret_cast<const float*>(
        static_cast<uintptr_t>(static_cast<uint32_t>(aux.z)) | (static_cast<uintptr_t>(static_cast<uint32_t>(aux.w)) << 32)));
    if (tid < aux2.x) { px = xyz[3 * tid]; py = xyz[3 * tid + 1]; }
    if (tid < (aux2.y - item.z + item.w - 1) / item.w) {
      const int at = item.z + tid * item.w;
      rot_mine = make_float2(rot[2 * at], rot[2 * at + 1]);
    }
  }
  CopyParams(&P, params + item.x, tid);
  __syncthreads();
  const int tile = item.y, g = item.z, G = item.w;
  const int side = 2 * P.nl + 1, cands = side * side;
  const int B = P.B, H = P.H, lp = P.lp;
  // rotations g, g + G, ... of this item.  Tiled matches: all of them fit the workgroup's LDS (the
  // planner's G >= gmin); one tile per match: taken in ROUNDS of P.rw rotations.
  const int rw = (P.num_scans - g + G - 1) / G;
  // (in-kernel timeline of the profiling tools: compiled in only for the instrumented
  // instantiation the debug switch `timeline` selects; slots = workgroup x its first items)
  const auto stamp = [&](int k) {
    if constexpr (kTimeline) {
      if (item_index < 4) Stamp(P.timeline, blockIdx.x * 4 + item_index, k);
    }
  };
  stamp(0);
  const int nkeys = P.ntx * P.nty * 4;
  int* acc = reinterpret_cast<int*>(tile_smem + P.tile_image_bytes);
  int* hdrs = acc + ((P.rw * cands + 3) & ~3);           // [rw][4] start | count << 16 of this tile
  int* slot = hdrs + P.rw * 4;                           // [rw + 1] cumulative list lengths
  int* tasks = slot + ((P.rw + 1 + 3) & ~3);             // [task_cap][4]
  int* ctl = tasks + P.task_cap * 4;                     // [0] tasks, [1] next task, [2] next job, [4] ticket
  // (fused: + rots[num_scans] | ax[n_pad] ay[n_pad] before the lists)
  const bool fused = P.fused != 0;
  const int pchunks = P.n_pad >> 6;
  float2* rots = reinterpret_cast<float2*>(ctl + 16);
  float* ax = reinterpret_cast<float*>(rots + (fused ? (P.num_scans + 1) & ~1 : 0));   // the cloud rotated by
  float* ay = ax + (fused ? P.n_pad : 0);                                              // the initial yaw
  uint16_t* list = reinterpret_cast<uint16_t*>(ay + (fused ? P.n_pad : 0));

  if (fused) {
    // the cloud rotated by the initial yaw and this item's rotations: into LDS BEFORE the image
    // copy is issued (loads return in order: behind the copy they would wait for it to land)
    const auto* xyz = AsGlobal(P.xyz);
    for (int i = tid; i < P.n_pad; i += num_threads) {
      float x = 0.f, y = 0.f;
      if (i < P.n) RotateZ(P.init_qw, P.init_qz, i == tid ? px : xyz[3 * i], i == tid ? py : xyz[3 * i + 1], &x, &y);
      ax[i] = x;
      ay[i] = y;
    }
    if (tid < rw) rots[tid] = rot_mine;
  }
  // ---- the tile's image: th_img rows of the quantised grid image + rpl * H rows of zeros,
  // LDS-DMA with one row piece (16 bytes) per lane.  The first piece of the grid image is halo
  // (zeros): the source of the null rows and of every piece outside the image -----------------
  const auto issue_image_copy = [&]() {
    const int tY = tile / P.ntx, tX = tile - tY * P.ntx;
    const int gx0 = P.box_x0 + tX * P.T, gy0 = P.box_y0 + tY * P.T;
    const int ppr = lp >> 4;
    const int pieces_img = P.th_img * ppr;
    const auto* src = (const __attribute__((address_space(1))) unsigned char*)P.qimage;
    auto* dst = (__attribute__((address_space(3))) unsigned char*)tile_smem;
    const int kib = P.tile_image_bytes >> 10;
    const int gw = P.gpitch >> 1;                         // cells per image row
    for (int k = wave; k < kib; k += num_waves) {
      const int p = (k << 6) + lane;
      const int row = p / ppr, c = p - row * ppr;
      const int X0 = gx0 + (c << 3), Y = gy0 + row;
      // (pieces right of / below the image are outside the grid: zeros, like the corner)
      const bool zero = p >= pieces_img || X0 >= gw || Y >= P.grows;
      const size_t at = zero ? 0 : static_cast<size_t>(Y) * P.gpitch + static_cast<size_t>(X0) * 2;
      __builtin_amdgcn_global_load_lds(src + at, dst + (k << 10), 16, 0, 0);
    }
  };
  issue_image_copy();
  stamp(1);                                              // image DMA issued
  const int lds_image = static_cast<int>(reinterpret_cast<uintptr_t>(
      (const __attribute__((address_space(3))) unsigned char*)tile_smem));
  // Lane geometry inside a half-wavefront.
  const int li = lane & 31;
  const int row = li / B, blk = li - row * B;
  const bool lane_used = row < H;
  // (lanes beyond H * B read the image's first rows like everyone else and drop the result)
  const int lane_off = lds_image + (lane_used ? row * lp + blk * 8 : 0);
  const int row_stride = H * lp;
  // ---- the window tasks of a round, dealt dynamically: the halves of a wavefront run two phases --
  const auto run_tasks = [&](int num_tasks) {
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
      RowPairAccumulate<RPL, kRowStride>(list + start, my_len, iters, lane, lane_off, row_stride,
                                         P.null_addr, acc32);
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
  };

  if (fused) {
    // ---- one tile per match: the rotations of this item are discretised HERE, from the cloud in
    // LDS (what the prep kernel does for tiled matches: same cells, same entries), in rounds of
    // P.rw rotations -- the image and the cloud are staged once per item, however many rounds.
    // A JOB = up to kJobChunks chunks (64 consecutive points each) of one rotation, done by ONE
    // wavefront in ONE pass: the entries stay in registers while the four phase counts are taken
    // (ballots: wave-uniform), the job's own fixed region of the list buffer is laid out from them
    // (sub-lists padded to 16 entries), the entries go to their places (rank by mbcnt: no atomics)
    // and the job's tasks -- its phases paired by size -- are appended to the round's.
    // The cell of a point comes from an f32 ESTIMATE of the reference's value
    //     t = (max - translation) / res - 0.5 - (rotated coordinate) / res
    // in two FMAs per coordinate; it differs from the f64 value the reference rounds
    // (GetCellIndex over RotateZ's f32 chain) by less than
    //     2^-24 [((k_z + 4) (|ax| + |ay|) + |translation|) / res + 3 |K|]
    // (RotateZ's chain for the rotation (w, z): a (2 + 4 z^2) + b (1 + 6 |w z|) <= k_z r; its sum
    // with the translation: r + |t|; the estimate's rounded constants and two FMAs: 3 r, 3 |K|),
    // and when it lies further than 1.25 x that from every half-integer its rounding is the
    // reference's cell.  Otherwise -- a chunk in forty or so -- the whole chunk runs the exact
    // expressions (Rt2DCellOfPrerotated).
    const int rcap = P.rw;
    const int jpr = (pchunks + kJobChunks - 1) / kJobChunks;          // jobs per rotation
    const int rot_stride = P.n_pad + 64 * jpr;                        // list entries per rotation
    const int n_pts = P.n, off_x = P.hl - P.nl - P.box_x0, off_y = P.ht - P.nl - P.box_y0, T = P.T;
    const int half_pitch = lp >> 1;
    const int ix_lo = -(P.nl + 1), ix_hi = P.nx + P.nl, iy_hi = P.ny + P.nl;
    const double inv_res = P.inv_res;
    const double Kyd = (P.max_y - static_cast<double>(P.ty)) * inv_res - 0.5;
    const double Kxd = (P.max_x - static_cast<double>(P.tx)) * inv_res - 0.5;
    const float Ky = static_cast<float>(Kyd), Kx = static_cast<float>(Kxd);
    // (bound = 1.25 x 2^-24 [((k_z + 4) r + |translation|) / res + 3 |K| + 1], r = |ax| + |ay|,
    // k_z = max(2 + 4 z^2, 1 + 6 |z|) for the rotation (w, z): the terms of RotateZ's chain)
    const double bound_unit = 1.25 * 0x1p-24 * inv_res;
    const float bound_fixed = static_cast<float>(
        1.25 * 0x1p-24 * (inv_res * fmax(fabs(static_cast<double>(P.tx)), fabs(static_cast<double>(P.ty))) +
                          3.0 * fmax(fabs(Kxd), fabs(Kyd)) + 1.0));
    bool outside = false;
    for (int rbase = 0; rbase < rw; rbase += rcap) {
      const int rw_round = min(rcap, rw - rbase);
      // (acc is all zero here: by the loop below in the first round, by the flush afterwards)
      if (rbase == 0)
        for (int i = tid; i < rw_round * cands; i += num_threads) acc[i] = 0;
      if (tid < 3) ctl[tid] = tid == 2 ? num_waves : 0;
      LdsBarrier();                                        // (not __syncthreads: the copy stays in flight)
      if (rbase == 0) stamp(2);                            // cloud in LDS
      const int jobs = rw_round * jpr;
      // (the first job of a wavefront is its own number, the others are drawn: a job whose chunks
      // need the exact expressions takes twice as long as one that does not)
#pragma unroll 1
      for (int job = wave; job < jobs;
           job = __builtin_amdgcn_readfirstlane(lane == 0 ? atomicAdd(&ctl[2], 1) : 0)) {
        const int rr = job / jpr, jj = job - rr * jpr;
        const float2 rot = rots[rbase + rr];
        const double wd = rot.x, zd = rot.y;
        const float Ci = static_cast<float>((1.0 - 2.0 * zd * zd) * inv_res);
        const float Si = static_cast<float>(2.0 * wd * zd * inv_res);
        const float bound_per_m =
            static_cast<float>(bound_unit * (4.0 + fmax(2.0 + 4.0 * zd * zd, 1.0 + 6.0 * fabs(zd))));
        int packed[kJobChunks];
        unsigned inexact = 0;                                // chunks the estimate could not decide
        const auto pack = [&](int ix, int iy, bool valid) {
          const int rx = ix + off_x, ry = iy + off_y;
          const bool inside = static_cast<unsigned>(rx) < static_cast<unsigned>(T) &&
                              static_cast<unsigned>(ry) < static_cast<unsigned>(T);
          if (valid && !inside) outside = true;              // (never: the host's box)
          // entry << 2 | phase = ((ry lp + (rx & ~3) 2) >> 3) << 2 | (rx & 3) = ry lp / 2 + rx
          return valid && inside ? ry * half_pitch + rx : -1;
        };
#pragma unroll
        for (int c = 0; c < kJobChunks; ++c) {
          // (every chunk of the job, without a branch: the chunks beyond the cloud read what lies
          // behind it in LDS and are masked by i < n)
          const int i = (jj * kJobChunks + c) * 64 + lane;
          const float x = ax[i], y = ay[i];
          const bool valid = i < n_pts;
          const float tY = fmaf(-Ci, y, fmaf(-Si, x, Ky));   // cell x index from the map's y
          const float tX = fmaf(-Ci, x, fmaf(Si, y, Kx));
          const float nY = rintf(tY), nX = rintf(tX);
          const float margin = fminf(0.5f - fabsf(tY - nY), 0.5f - fabsf(tX - nX));
          const float bound = fmaf(fabsf(x) + fabsf(y), bound_per_m, bound_fixed);
          if (__ballot(valid && !(margin > bound))) inexact |= 1u << c;   // (NaN: not greater)
          const int ix = min(max(static_cast<int>(nY), ix_lo), ix_hi);
          const int iy = min(max(static_cast<int>(nX), ix_lo), iy_hi);
          packed[c] = pack(ix, iy, valid);
        }
#pragma unroll 1
        while (inexact) {                                    // (uniform; a chunk in twelve or so)
          const int c = __builtin_ctz(inexact);
          inexact &= inexact - 1;
          const int i = (jj * kJobChunks + c) * 64 + lane;
          int ix, iy;
          Rt2DCellOfPrerotated(FrameOf(P), rot.x, rot.y, ax[i], ay[i], &ix, &iy);
          const int pk = pack(ix, iy, i < n_pts);
#pragma unroll
          for (int k = 0; k < kJobChunks; ++k) packed[k] = k == c ? pk : packed[k];
        }
        int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
#pragma unroll
        for (int c = 0; c < kJobChunks; ++c) {
          const int pk = packed[c];
          c0 += __popcll(__ballot(pk >= 0 && (pk & 3) == 0));
          c1 += __popcll(__ballot(pk >= 0 && (pk & 3) == 1));
          c2 += __popcll(__ballot(pk >= 0 && (pk & 3) == 2));
          c3 += __popcll(__ballot(pk >= 0 && (pk & 3) == 3));
        }
        // the job's region: sub-lists in phase order, each padded to 16 entries
        const int region = rr * rot_stride + jj * (kJobChunks * 64 + 64);
        int s0 = region, s1 = s0 + ((c0 + 15) & ~15), s2 = s1 + ((c1 + 15) & ~15),
            s3 = s2 + ((c2 + 15) & ~15);
        {
          // tasks: the four phases by count, descending (sorting network of five exchanges),
          // paired (1st, 2nd), (3rd, 4th); a task = at most kPairTaskIters entries of each
          int key[4] = {(c0 << 2) | 0, (c1 << 2) | 1, (c2 << 2) | 2, (c3 << 2) | 3};
#define CMX_CSWAP(I, J) { const int hi_k = max(key[I], key[J]), lo_k = min(key[I], key[J]); key[I] = hi_k; key[J] = lo_k; }
          CMX_CSWAP(0, 1) CMX_CSWAP(2, 3) CMX_CSWAP(0, 2) CMX_CSWAP(1, 3) CMX_CSWAP(1, 2)
#undef CMX_CSWAP
          const int nA = ((key[0] >> 2) + kPairTaskIters - 1) / kPairTaskIters;
          const int nB = ((key[2] >> 2) + kPairTaskIters - 1) / kPairTaskIters;
          int t0 = 0;
          if (lane == 0) t0 = atomicAdd(&ctl[0], nA + nB);
          t0 = __builtin_amdgcn_readfirstlane(t0);
          if (lane < nA + nB) {
            const bool second = lane >= nA;
            const int off = (lane - (second ? nA : 0)) * kPairTaskIters;
            const int ka = second ? key[2] : key[0], kb = second ? key[3] : key[1];
            const int pa = ka & 3, pb = kb & 3, la = ka >> 2, lb = kb >> 2;
            const int sa = pa == 0 ? s0 : pa == 1 ? s1 : pa == 2 ? s2 : s3;
            const int sb = pb == 0 ? s0 : pb == 1 ? s1 : pb == 2 ? s2 : s3;
            reinterpret_cast<int4*>(tasks)[t0 + lane] =
                make_int4(rr | (pa << 8) | (pb << 16), sa + off, sb + off,
                          min(kPairTaskIters, la - off) | (max(0, min(kPairTaskIters, lb - off)) << 16));
          }
        }
#pragma unroll
        for (int c = 0; c < kJobChunks; ++c) {
          const int pk = packed[c];
          const int ph = pk & 3;
          const unsigned long long m0 = __ballot(pk >= 0 && ph == 0);
          const unsigned long long m1 = __ballot(pk >= 0 && ph == 1);
          const unsigned long long m2 = __ballot(pk >= 0 && ph == 2);
          const unsigned long long m3 = __ballot(pk >= 0 && ph == 3);
          const unsigned long long mine = ph == 0 ? m0 : ph == 1 ? m1 : ph == 2 ? m2 : m3;
          const int at = ph == 0 ? s0 : ph == 1 ? s1 : ph == 2 ? s2 : s3;
          const int rank = __builtin_amdgcn_mbcnt_hi(static_cast<unsigned>(mine >> 32),
                                                     __builtin_amdgcn_mbcnt_lo(static_cast<unsigned>(mine), 0));
          if (pk >= 0) list[at + rank] = static_cast<uint16_t>(pk >> 2);
          s0 += __popcll(m0); s1 += __popcll(m1); s2 += __popcll(m2); s3 += __popcll(m3);
        }
      }
      if (rbase == 0) {
        stamp(4);                                            // this wave's jobs done
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's share of the image has landed
        if (tid == 0) ctl[4] = ticket;                       // (the ticket arrived with that wait)
      }
      LdsBarrier();
      if (rbase == 0) {
        stamp(5);                                            // image landed
        if (tid < work_stride) {
          const int t = ctl[4];
          next = t < num_items ? work[t * work_stride + tid] : make_int4(-1, 0, 0, 0);
        }
      }
      run_tasks(ctl[0]);                                     // <= task_cap by construction (host)
      if (rbase == 0) stamp(6);                              // wave 0 out of tasks (first round)
      LdsBarrier();
      // ---- this round's sums (plain stores: the match's only tile), the accumulators zeroed for
      // the next round ---------------------------------------------------------------------------
      for (int e = tid; e < rw_round * cands; e += num_threads) {
        const int rr = e / cands, c = e - rr * cands;
        AsGlobal(P.qsum)[static_cast<size_t>(g + (rbase + rr) * G) * cands + c] = acc[e];
        acc[e] = 0;
      }
    }
    if (outside) atomicOr(&P.misc[0], kOutOfBox);
    stamp(7);
  } else {
  for (int i = tid; i < rw * cands; i += num_threads) acc[i] = 0;
  if (tid < rw * 4) {
    const int rr = tid >> 2, ph = tid & 3;
    hdrs[tid] = static_cast<int>(P.hdr[static_cast<size_t>(g + rr * G) * nkeys + tile * 4 + ph]);
  }
  __syncthreads();
  if (wave == 0) {                   // cumulative (padded) list lengths of this tile's rotations
    const int rr = lane;
    int len = 0;
    if (rr < rw) {
#pragma unroll
      for (int ph = 0; ph < 4; ++ph) len += ((hdrs[rr * 4 + ph] >> 16) + 15) & ~15;
    }
    const int incl = WaveInclusiveScan(len);
    if (rr < rw) slot[rr + 1] = incl;
    if (lane == 0) slot[0] = 0;
  }
  __syncthreads();
  stamp(2);                                              // headers, list lengths

  // ---- rounds: as many rotations as the LDS list buffer holds ------------------------------
  for (int rr0 = 0; rr0 < rw;) {
    const int base = slot[rr0];
    int rr1 = rr0 + 1;                                 // (one rotation always fits: host)
    while (rr1 < rw && slot[rr1 + 1] - base <= P.list_lds) ++rr1;
    __syncthreads();                                   // the previous round's lists are done with
    if (tid < 2) ctl[tid] = 0;
    // lists: rotation rr's entries of this tile are contiguous in HBM (keys in order)
    // (wavefront 0 builds the round's tasks below while the others copy its lists)
    for (int rr = rr0 + wave - 1; wave >= 1 && rr < rr1; rr += num_waves - 1) {
      const int first = hdrs[rr * 4] & 0xffff;
      const int len = slot[rr + 1] - slot[rr];
      typedef unsigned U4 __attribute__((ext_vector_type(4)));
      const auto* src = AsGlobal(reinterpret_cast<const U4*>(
          P.lists + static_cast<size_t>(g + rr * G) * P.cap_s + first));
      U4* dst = reinterpret_cast<U4*>(list + (slot[rr] - base));
      for (int q = lane; q < (len >> 3); q += 64) dst[q] = src[q];
    }
    // ---- per rotation: phases paired by size, tasks of at most kPairTaskIters iterations ---
    if (wave == 0) {
      const int rr = rr0 + lane;
      const bool live = rr < rr1;
      int key[4], first_slot[4];                 // count << 2 | phase
      const int g_first = live ? hdrs[rr * 4] & 0xffff : 0;
#pragma unroll
      for (int ph = 0; ph < 4; ++ph) {
        const int h = live ? hdrs[rr * 4 + ph] : 0;
        key[ph] = ((h >> 16) << 2) | ph;
        first_slot[ph] = (h & 0xffff) - g_first + (live ? slot[rr] - base : 0);
      }
      // the four phases by count, descending (sorting network of five exchanges)
#define CMX_CSWAP(I, J) { const int hi_k = max(key[I], key[J]), lo_k = min(key[I], key[J]); key[I] = hi_k; key[J] = lo_k; }
      CMX_CSWAP(0, 1) CMX_CSWAP(2, 3) CMX_CSWAP(0, 2) CMX_CSWAP(1, 3) CMX_CSWAP(1, 2)
#undef CMX_CSWAP
      const int la0 = key[0] >> 2, la1 = key[2] >> 2;
      const int mine = (la0 + kPairTaskIters - 1) / kPairTaskIters +
                       (la1 + kPairTaskIters - 1) / kPairTaskIters;
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
    if (rr0 == 0) stamp(4);                              // tasks built / this wave's lists in LDS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's share of the image has landed
    if (rr0 == 0 && tid == 0) ctl[4] = ticket;           // (the ticket arrived with that wait)
    __syncthreads();
    if (rr0 == 0) stamp(5);                              // image landed
    if (rr0 == 0 && tid < work_stride) {
      const int t = ctl[4];
      next = t < num_items ? work[t * work_stride + tid] : make_int4(-1, 0, 0, 0);
    }
    run_tasks(ctl[0]);                                   // <= task_cap by construction (host)
    if (rr0 == 0) stamp(6);                              // wave 0 out of tasks (first round)
    rr0 = rr1;
  }
  __syncthreads();
  stamp(7);                                              // all rounds done
  // ---- this tile's share of the candidates' integer sums --------------------------------------
  for (int e = tid; e < rw * cands; e += num_threads) {
    const int rr = e / cands, c = e - rr * cands;
    auto* out = AsGlobal(P.qsum) + static_cast<size_t>(g + rr * G) * cands + c;
    const int v = acc[e];
    if (P.flush_atomic) {
      if (v) __hip_atomic_fetch_add(out, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      *out = v;
    }
  }
  }   // tiled matches
  stamp(8);
  if (tid < work_stride) fetched[tid] = next;
  }   // work items
}

// (kShift: what the sums in qsum were quantised by -- the tile kernels' 10-bit cells, or the bytes
// of the bound kernel's tail in its verify mode)
template <int kShift>
__global__ void __launch_bounds__(kFinishThreads)
Rt2DFinishKernel(const Rt2DTileParams* __restrict__ params, int group,
                 unsigned* __restrict__ host_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char fin_smem[];
  __shared__ Rt2DTileParams P;
  CopyParams(&P, params + blockIdx.x, threadIdx.x);
  __syncthreads();
  Rt2DFinishMatch<kFinishThreads, false, true, kShift>(P, fin_smem, group, host_out, blockIdx.x);
}

#include "rt_2d_bounds.h"

size_t Align16(size_t v) { return (v + 15) & ~static_cast<size_t>(15); }

// Conflict-free LDS row pitch (bytes, a multiple of 16, >= min_bytes; 0: none exists): the H x B
// 8-byte blocks a half-wavefront reads in one LDS cycle fall into 2 H B distinct banks for every
// base address.  The bank pattern depends on the pitch modulo 256 only: the sixteen residues are
// tested once per (H, B).
int ConflictFreePitch(int H, int B, int min_bytes) {
  static std::mutex mu;
  static int valid[33][33];                  // bit k: pitch = 16 k (mod 256) is conflict-free; -1 unset
  static bool init = false;
  unsigned mask;
  {
    std::lock_guard<std::mutex> lock(mu);
    if (!init) {
      for (auto& row : valid) for (int& v : row) v = -1;
      init = true;
    }
    if (valid[H][B] < 0) {
      int m = 0;
      for (int k = 0; k < 16; ++k) {
        const int cand = 16 * k + 256;
        unsigned long long used = 0;
        bool ok = true;
        for (int r = 0; r < H && ok; ++r)
          for (int b = 0; b < B && ok; ++b)
            for (int w = 0; w < 2; ++w) {
              const int bank = ((r * cand + b * 8) / 4 + w) & 63;
              if (used >> bank & 1) ok = false;
              used |= 1ull << bank;
            }
        if (ok) m |= 1 << k;
      }
      valid[H][B] = m;
    }
    mask = static_cast<unsigned>(valid[H][B]);
  }
  if (mask == 0) return 0;
  for (int cand = (min_bytes + 15) & ~15;; cand += 16)
    if (mask >> ((cand >> 4) & 15) & 1) return cand;
}

struct TileGeometry {
  int B, H, hl, ht, gpitch, grows;
  int box_x0, box_y0, T, ntx, nty, lp, th_img, tile_image_bytes;
  int cap_s, gmin, gmax, target, rw, list_lds, task_cap;
  int groups;               // fused: the rotation groups (work items) of this match, chosen by the host
  size_t lds;               // of the tile kernel
  size_t image_bytes;       // of the grid image in HBM: quantised image + pooled planes
  size_t q_bytes;           // of the quantised image alone (the planes start behind it)
  int m2_pitch, m2_rows;    // pooled planes (0: window beyond 16 x 16, no planes)
  int b_lpb, b_lh;          // bound kernel: LDS copy of the planes
  size_t b_lds;             // bound kernel: dynamic LDS of this match
  size_t b_finish_room;     // of it: planes | zeros | cloud, which its fused finish lays out anew
  size_t b_tail_at;         // where the rest starts: behind those, and behind what the finish needs
  int m4_pitch, m4_rows;    // 4 x 4 pooled planes (round 6)
  size_t q8_bytes;          // the byte image behind them
  int b8_lp, b8_lh;         // tail kernel: LDS box of the byte image
  size_t t4_lds;
  int b4_lpb, b4_lh, b4_pstride;
  size_t b4_tail_at, b4_lds;
};

}  // namespace

// ---------------------------------------------------------------------------------------------
// The image of a resident grid: two buffers, so that the image of a NEW grid version is built
// while matches on the old one may still be reading theirs.
// ---------------------------------------------------------------------------------------------
Rt2DImageCache::~Rt2DImageCache() {
  for (Buffer& b : buffer)
    if (b.image) (void)hipFree(b.image);
}

// Returns the buffer holding the image of (version, nl, geometry), pinned for reading, or a
// buffer to build it into (`*build` = true; exclusive until Publish / Abandon), or -1 when both
// buffers are in use by other calls (the caller builds into scratch of its own).
int Rt2DImageCache::Acquire(unsigned long long version, int nl, int gpitch, int grows,
                            size_t bytes, bool* build) {
  std::lock_guard<std::mutex> lock(mutex);
  *build = false;
  for (int k = 0; k < 2; ++k) {
    Buffer& b = buffer[k];
    if (b.valid && b.version == version && b.nl == nl && b.gpitch == gpitch && b.grows == grows) {
      ++b.readers;
      return k;
    }
  }
  // a buffer nobody reads and nobody builds: prefer the one that does not hold the newest image
  int pick = -1;
  for (int k = 0; k < 2; ++k) {
    const Buffer& b = buffer[k];
    if (b.readers != 0 || b.building) continue;
    if (pick < 0 || !b.valid || (buffer[pick].valid && b.version < buffer[pick].version)) pick = k;
  }
  if (pick < 0) return -1;
  Buffer& b = buffer[pick];
  if (b.capacity < bytes) {
    if (b.image) (void)hipFree(b.image);
    b.image = nullptr;
    b.capacity = 0;
    b.valid = false;
    CMX_HIP(hipMalloc(reinterpret_cast<void**>(&b.image), bytes + bytes / 4));
    b.capacity = bytes + bytes / 4;
  }
  b.valid = false;
  b.building = true;
  *build = true;
  return pick;
}

void Rt2DImageCache::Publish(int k, unsigned long long version, int nl, int gpitch, int grows) {
  std::lock_guard<std::mutex> lock(mutex);
  Buffer& b = buffer[k];
  b.version = version; b.nl = nl; b.gpitch = gpitch; b.grows = grows;
  b.building = false;
  b.valid = true;
}

void Rt2DImageCache::Release(int k, bool was_building) {
  std::lock_guard<std::mutex> lock(mutex);
  Buffer& b = buffer[k];
  if (was_building) { b.building = false; b.valid = false; }
  else --b.readers;
}

namespace {

// One grid image per distinct (grid, version, window) of a call.
struct GridKey {
  const void* cells; const void* device_cells; const void* cache;
  unsigned long long version; int nl, gpitch, grows;
  bool operator==(const GridKey& o) const {
    return cells == o.cells && device_cells == o.device_cells && cache == o.cache &&
           version == o.version && nl == o.nl && gpitch == o.gpitch && grows == o.grows;
  }
};
struct GridKeyHash {
  size_t operator()(const GridKey& k) const {
    size_t h = reinterpret_cast<size_t>(k.cells) * 0x9e3779b97f4a7c15ull;
    h ^= reinterpret_cast<size_t>(k.device_cells) + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2);
    h ^= static_cast<size_t>(k.version) * 0xff51afd7ed558ccdull + static_cast<size_t>(k.nl);
    return h;
  }
};

// Releases / publishes the cache buffers of a call on every path out of it.
struct CacheHold {
  Rt2DImageCache* cache = nullptr;
  int buffer = -1;
  bool building = false;
  unsigned long long version = 0;
  int nl = 0, gpitch = 0, grows = 0;
};
struct CacheHolds {
  std::vector<CacheHold> holds;
  hipStream_t stream = nullptr;
  bool built = false;                   // the stream has been waited for: builds are complete
  ~CacheHolds() {
    // (a call that fails between its launches and its wait: nothing of it may still be writing
    // or reading a buffer that is handed back)
    if (!built && !holds.empty()) (void)hipStreamSynchronize(stream);
    for (CacheHold& h : holds) {
      if (h.buffer < 0) continue;
      if (h.building && built) {
        h.cache->Publish(h.buffer, h.version, h.nl, h.gpitch, h.grows);
        // (the builder keeps reading it until this call returns: nothing to release, Publish
        // cleared `building`; a buffer that is neither read nor built may be picked again, but
        // only by a call that starts after this one's kernels have completed)
      } else {
        h.cache->Release(h.buffer, h.building);
      }
    }
  }
};

}  // namespace

void Rt2DComputeSearch(const cmx_rt_options* options, const Rt2DItem& it, Rt2DSearch* out) {
  const int n = it.n;
  const double res = it.limits->resolution;
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
  out->step = kSafetyMargin * std::acos(1. - (res * (res * 1.)) / (2. * range_sq));
  out->na = std::ceil(options->angular_search_window / out->step);
  out->q0w = q0w; out->q0z = q0z;
  out->max_range = std::sqrt(max_all);
  out->num_scans = 2 * out->na + 1;
  out->nl = std::ceil(options->linear_search_window / res);
}

void Rt2DFinishOnHost(const cmx_rt_options* options, const Rt2DItem& it, const Rt2DSearch& sr,
                      const std::pair<int, float>* finalists, size_t count) {
  // Exact weighting + first-maximum on the finalists (:142-143,170-174); `finalists` ascend by
  // candidate index (the reference's generation order), so the first maximum wins.
  const int side_i = 2 * sr.nl + 1, nl = sr.nl, na = sr.na;
  const double res = it.limits->resolution, step = sr.step;
  float best_score = -1.f;
  int best = -1;
  for (size_t k = 0; k < count; ++k) {
    const int c = finalists[k].first;
    const int s = c / (side_i * side_i);
    const int rem = c - s * side_i * side_i;
    const int dx = rem / side_i - nl, dy = rem % side_i - nl;
    const double cx = -dy * res, cy = -dx * res;
    const double theta = (s - na) * step;
    const double t = std::hypot(cx, cy) * options->translation_delta_cost_weight +
                     std::abs(theta) * options->rotation_delta_cost_weight;
    float sc = finalists[k].second;
    sc *= std::exp(-(t * (t * 1.)));
    if (sc > best_score) { best_score = sc; best = c; }
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
}

// ---------------------------------------------------------------------------------------------
// The tile path for a batch of probability-grid matches, in three steps so that ONE host thread
// can keep several batches (the parts of a large call) in flight on streams of their own:
//   Plan()     geometry; false: a match is not eligible (huge window or cloud);
//   Enqueue()  staging buffer, grid images, upload and the four launches -- asynchronous;
//   Collect()  waits for the stream; exact weighting and first maximum on the host; false -- and
//              nothing written to the items -- when a score landscape is flat (more candidates
//              within the bounds than the lists hold) or a point fell outside the predicted box:
//              the caller then runs the batch on the per-candidate kernels.
// ---------------------------------------------------------------------------------------------
struct Rt2DTileCall::Impl {
  const cmx_rt_options* options;
  const Rt2DItem* items;
  const Rt2DSearch* search;
  int num, device;
  std::vector<TileGeometry> geo;
  int rpl = 1, max_scans = 0, common_stride = -1, group = 8, cus = 256;
  int share = 1;                        // calls sharing the device (the parts of a batch)
  int batch_matches = 0;                // matches of the whole batch (>= num)
  size_t tile_lds = 0, prep_lds = 0, finish_lds = 0;
  size_t lists_total = 0, hdr_total = 0, qsum_total = 0;
  long long work_cap = 0, entries_total = 0;
  int tile_grid = 0, tile_threads = 512;
  bool fused = false;                   // one tile per match, prep fused into the tile kernel
  bool bounds = false;                  // block bounds first (rt_2d_bounds.h): fused shape only
  bool split = false;                   // ... with the tail of every match in a kernel of its own
  bool level4 = false;                  // ... and blocks of 4 x 4 translations first (always split)
  size_t tail_lds = 0;
  size_t bound4_lds = 0, tail4_lds = 0;
  int bound_nb = 0;                     // blocks per window axis (launch-wide: the largest)
  size_t bound_lds = 0;
  // (the per-rotation (cos, sin) pairs of every match, libm: worked out by Plan() -- off the
  // calling thread when the batch goes out in parts -- and copied into the staging buffer)
  std::vector<float2> tables;
  std::vector<size_t> table_at;
  std::unique_ptr<WorkspaceLease> ws;
  CacheHolds holds;
  unsigned* h_misc = nullptr;
  unsigned* d_overflow = nullptr;
  unsigned long long* d_timeline = nullptr;
  bool enqueued = false, synced = false;
};

Rt2DTileCall::Rt2DTileCall(const cmx_rt_options* options, const Rt2DItem* items,
                           const Rt2DSearch* search, int num, int32_t device, int concurrent_calls,
                           int batch_matches)
    : impl_(new Impl) {
  impl_->options = options; impl_->items = items; impl_->search = search;
  impl_->num = num; impl_->device = device;
  impl_->share = std::max(1, concurrent_calls);
  impl_->batch_matches = std::max(num, batch_matches);
}
Rt2DTileCall::~Rt2DTileCall() {
  // (a call abandoned between its launches and its wait -- an exception in a later part: nothing
  // of it may still be running when its workspace goes back to the pool)
  if (impl_->enqueued && !impl_->synced && impl_->ws) (void)hipStreamSynchronize((*impl_->ws)->stream);
}

bool Rt2DTileCall::Plan() {
  Impl& I = *impl_;
  const DebugOptions& dbg = Debug();
  const Rt2DItem* items = I.items;
  const Rt2DSearch* search = I.search;
  const int num = I.num;
  (void)hipDeviceGetAttribute(&I.cus, hipDeviceAttributeMultiprocessorCount, I.device);
  const int cus = I.cus;
  std::vector<TileGeometry>& geo = I.geo;
  geo.assign(num, TileGeometry{});
  // ---- window geometry per item; launch-wide rows per lane: the largest any item needs.
  // (H = the most rows of B blocks a half-wavefront holds for which a conflict-free pitch of whole
  // 16-byte DMA pieces exists: e.g. 8 rather than 10 rows of 3 blocks)
  int rpl = 1;
  for (int m = 0; m < num; ++m) {
    const int side = 2 * search[m].nl + 1;
    const int B = (side + 3 + 3) / 4;
    if (B > 32 || items[m].n > kRt2DMaxPoints || search[m].num_scans > 4096) return false;
    if (items[m].limits->num_x_cells > 32000 || items[m].limits->num_y_cells > 32000)
      return false;                                        // (cells travel as int16 pairs)
    int H = m > 0 && geo[m - 1].B == B ? geo[m - 1].H : 32 / B;
    while (H > 1 && ConflictFreePitch(H, B, 16) == 0) --H;
    geo[m].B = B;
    geo[m].H = H;
    rpl = std::max(rpl, (side + H - 1) / H);
  }
  if (rpl > kMaxRowsPerLane) return false;
  if (rpl == 5) rpl = 6;
  if (rpl == 7) rpl = 8;                         // (instantiated: 1, 2, 3, 4, 6, 8 rows per lane)
  I.rpl = rpl;
  // ---- per item: the box of window starts the scan can reach.  Every rotated point lies within
  // max_range of the initial translation (plus a cell for the f32 roundings of rotation and
  // index, plus the half cell of lround); cells are clamped to [-(nl + 1), n + nl] as on the
  // device.  Items with the same window (the usual batch) share ONE tile geometry, chosen for the
  // largest box, cloud and rotation count among them.
  struct Class { int nl, span_x, span_y, n_pad, scans, items; TileGeometry g; };
  std::vector<Class> classes;
  std::vector<int> class_of(num), span_x(num), span_y(num);
  for (int m = 0; m < num; ++m) {
    const Rt2DItem& it = items[m];
    const Rt2DSearch& sr = search[m];
    TileGeometry& g = geo[m];
    const int nx = it.limits->num_x_cells, ny = it.limits->num_y_cells, nl = sr.nl;
    g.hl = (2 * nl + 1 + 7) & ~7;
    g.ht = 2 * nl + 1;
    const double res = it.limits->resolution;
    const double reach = (sr.max_range * (1.0 + 1e-5)) / res + 2.0;
    const double cxc = (it.limits->max_y - it.initial->y) / res - 0.5;   // cell x from the map's y
    const double cyc = (it.limits->max_x - it.initial->x) / res - 0.5;
    const auto clampi = [](double v, int lo, int hi) {
      return static_cast<int>(std::min<double>(std::max<double>(v, lo), hi));
    };
    const int ix_lo = clampi(std::floor(cxc - reach), -(nl + 1), nx + nl);
    const int ix_hi = clampi(std::ceil(cxc + reach), -(nl + 1), nx + nl);
    const int iy_lo = clampi(std::floor(cyc - reach), -(nl + 1), ny + nl);
    const int iy_hi = clampi(std::ceil(cyc + reach), -(nl + 1), ny + nl);
    g.box_x0 = (ix_lo - nl + g.hl) & ~7;
    g.box_y0 = iy_lo - nl + g.ht;
    span_x[m] = ix_hi - nl + g.hl - g.box_x0 + 1;
    span_y[m] = iy_hi - nl + g.ht - g.box_y0 + 1;
    const int n_pad = (it.n + 63) / 64 * 64;
    int c = 0;
    while (c < static_cast<int>(classes.size()) && classes[c].nl != nl) ++c;
    if (c == static_cast<int>(classes.size())) classes.push_back(Class{nl, 0, 0, 0, 0, 0, g});
    Class& k = classes[c];
    k.span_x = std::max(k.span_x, span_x[m]); k.span_y = std::max(k.span_y, span_y[m]);
    k.n_pad = std::max(k.n_pad, n_pad); k.scans = std::max(k.scans, sr.num_scans);
    ++k.items;
    class_of[m] = c;
    I.entries_total += static_cast<long long>(it.n) * sr.num_scans;
  }
  // Two shapes.  (1) ONE tile per match, a workgroup of 1024 threads alone on its CU (up to
  // 150 KB of LDS): whenever the whole box of a match fits -- the items of a match are then its
  // rotation groups, all of one size, their sums plain stores.  C1's box (217 window starts per
  // axis at 5 cm) is such a case: as four tiles one of them held most of the points, and the
  // planner's smaller items each paid the ~4 us before their first window update.  (2) Several
  // tiles, workgroups of 512 threads, two per CU (76 KB each): boxes beyond that (long-range
  // scans, fine grids), sums by atomics.
  const size_t kLdsTwoPerCu = (dbg.rt2d_lds_kb > 0 ? dbg.rt2d_lds_kb : 76) * size_t{1024};
  const size_t kLdsOnePerCu = 150 * size_t{1024};
  bool single_tile = dbg.rt2d_tile <= 0;
  bool want_fused = !dbg.rt2d_unfused;
  for (const Class& k : classes) want_fused = want_fused && k.n_pad <= kFusedMaxPoints;
  for (Class& k : classes) {
    TileGeometry& g = k.g;
    const int side = 2 * k.nl + 1;
    const int rows_extra = rpl * g.H;
    const int cap_tile = k.n_pad + 64;                // a rotation's entries of ONE tile, padded
    const bool fuse = want_fused;
    const auto try_tile = [&](int T, size_t budget, bool one_tile, TileGeometry* out) {
      const int ntx = (k.span_x + T - 1) / T, nty = (k.span_y + T - 1) / T;
      if (ntx * nty > (one_tile ? 1 : kMaxTiles)) return false;
      const int lp = ConflictFreePitch(g.H, g.B, 2 * (T + 4 * g.B));
      if (lp == 0) return false;
      const int th_img = T + rows_extra;
      const int image = ((th_img + rows_extra) * lp + 1023) & ~1023;
      // rotations a workgroup's LDS holds: all of them (the planner gives a light tile ONE item)
      // or -- one tile per match -- as many as fit beside the image
      for (int gmin = std::max(1, (k.scans + 63) / 64); gmin <= k.scans; ++gmin) {
        const int rw = (k.scans + gmin - 1) / gmin;
        const bool fused_shape = one_tile && fuse;
        // (fused: rw = the rotations of one ROUND; a discretisation job's region of the list buffer
        // holds its chunks' entries plus the padding of four sub-lists, its tasks two more than its
        // chunks: Rt2DTileKernel)
        const int jpr = (k.n_pad / 64 + kJobChunks - 1) / kJobChunks;
        const int cap_rot = fused_shape ? k.n_pad + 64 * jpr : cap_tile;
        const int task_cap = fused_shape ? rw * jpr * (kJobChunks + 2) : rw * (k.n_pad / kPairTaskIters + 2);
        size_t fixed = static_cast<size_t>(image) + 4 * ((static_cast<size_t>(rw) * side * side + 3) & ~size_t{3}) +
                       16 * static_cast<size_t>(rw) + 4 * ((static_cast<size_t>(rw) + 1 + 3) & ~size_t{3}) +
                       16 * static_cast<size_t>(task_cap) + 64;
        if (fused_shape) {
          // (all rotations of the match, the cloud rotated by the initial yaw)
          fixed += 8 * ((static_cast<size_t>(k.scans) + 1) & ~size_t{1}) + 8 * static_cast<size_t>(k.n_pad);
        }
        if (fixed + 2 * static_cast<size_t>(cap_rot) > budget) {
          if (!one_tile) return false;
          continue;                                      // fewer rotations per workgroup
        }
        // list buffer: what is left, at most every rotation's entries at once
        const size_t want = static_cast<size_t>(rw) * cap_rot;
        const size_t room = (budget - fixed) / 2;
        if (one_tile && room < want) continue;           // (all of an item's lists in one round)
        out->list_lds = static_cast<int>(std::min(want, room) & ~size_t{7});
        out->T = T; out->ntx = ntx; out->nty = nty; out->lp = lp; out->th_img = th_img;
        out->tile_image_bytes = image; out->gmin = gmin; out->rw = rw; out->task_cap = task_cap;
        out->lds = fixed + 2 * static_cast<size_t>(out->list_lds);
        return true;
      }
      return false;
    };
    bool found = false;
    if (single_tile) {
      const int T = (std::max(k.span_x, k.span_y) + 7) & ~7;
      found = try_tile(T, kLdsOnePerCu, true, &g);
      if (!found) single_tile = false;
    }
    if (!found) {
      // The largest tile core that leaves room for two workgroups per CU, then the smallest core
      // with the same number of tiles (less image to stage, more room for lists).
      const int t_first = dbg.rt2d_tile > 0 ? (dbg.rt2d_tile & ~7) : 128;
      for (int T = t_first; T >= 16 && !found; T -= 8) found = try_tile(T, kLdsTwoPerCu, false, &g);
      if (found && dbg.rt2d_tile <= 0) {
        TileGeometry smaller = g;
        for (int T = g.T - 8; T >= 16; T -= 8) {
          if ((k.span_x + T - 1) / T != g.ntx || (k.span_y + T - 1) / T != g.nty) break;
          if (try_tile(T, kLdsTwoPerCu, false, &smaller)) g = smaller;
        }
      }
    }
    if (!found) return false;
  }
  // (classes of one launch share the shape: a class that needs several tiles puts all on shape 2)
  if (!single_tile) {
    for (Class& k : classes) {
      if (k.g.ntx * k.g.nty == 1 && k.g.lds > kLdsTwoPerCu) {
        // re-plan this class with the two-per-CU budget
        TileGeometry& g = k.g;
        const int side = 2 * k.nl + 1;
        const int rows_extra = rpl * g.H;
        const int cap_tile = k.n_pad + 64;
        bool found = false;
        for (int T = 128; T >= 16 && !found; T -= 8) {
          const int ntx = (k.span_x + T - 1) / T, nty = (k.span_y + T - 1) / T;
          if (ntx * nty > kMaxTiles) break;
          const int lp = ConflictFreePitch(g.H, g.B, 2 * (T + 4 * g.B));
          if (lp == 0) continue;
          const int th_img = T + rows_extra;
          const int image = ((th_img + rows_extra) * lp + 1023) & ~1023;
          const int gmin = std::max(1, (k.scans + 63) / 64);
          const int rw = (k.scans + gmin - 1) / gmin;
          const int task_cap = rw * (k.n_pad / kPairTaskIters + 2);
          const size_t fixed = static_cast<size_t>(image) + 4 * ((static_cast<size_t>(rw) * side * side + 3) & ~size_t{3}) +
                               16 * static_cast<size_t>(rw) + 4 * ((static_cast<size_t>(rw) + 1 + 3) & ~size_t{3}) +
                               16 * static_cast<size_t>(task_cap) + 64;
          if (fixed + 2 * static_cast<size_t>(cap_tile) > kLdsTwoPerCu) continue;
          const size_t want = static_cast<size_t>(rw) * cap_tile;
          const size_t room = (kLdsTwoPerCu - fixed) / 2;
          g.list_lds = static_cast<int>(std::min(want, room) & ~size_t{7});
          g.T = T; g.ntx = ntx; g.nty = nty; g.lp = lp; g.th_img = th_img;
          g.tile_image_bytes = image; g.gmin = gmin; g.rw = rw; g.task_cap = task_cap;
          g.lds = fixed + 2 * static_cast<size_t>(g.list_lds);
          found = true;
        }
        if (!found) return false;
      }
    }
  }
  I.tile_threads = single_tile ? 1024 : 512;
  I.fused = single_tile && want_fused;
  // Entries per work item: the whole batch in about two items per resident workgroup (an item
  // costs ~4 us before its first window update: ticket, headers, lists, tasks, image), not less
  // than two thousand entries (sixteen tasks: one per wavefront).
  int launch_nb = 1;                     // block bounds: blocks per window axis, launch-wide
  for (int m = 0; m < num; ++m) launch_nb = std::max(launch_nb, search[m].nl + 1);
  const int tile_slots = std::max(1, (single_tile ? cus : 2 * cus) / I.share);
  const int target = dbg.rt2d_target > 0
                         ? dbg.rt2d_target
                         : static_cast<int>(std::min<long long>(16384, std::max<long long>(2048, I.entries_total / (2ll * tile_slots))));
  for (int m = 0; m < num; ++m) {
    const Rt2DItem& it = items[m];
    const Rt2DSearch& sr = search[m];
    TileGeometry& g = geo[m];
    const TileGeometry& cg = classes[class_of[m]].g;
    const int nx = it.limits->num_x_cells, ny = it.limits->num_y_cells, nl = sr.nl;
    const int side = 2 * nl + 1, n_pad = (it.n + 63) / 64 * 64;
    g.T = cg.T; g.lp = cg.lp; g.th_img = cg.th_img; g.tile_image_bytes = cg.tile_image_bytes;
    g.task_cap = cg.task_cap; g.list_lds = cg.list_lds; g.lds = cg.lds;
    g.ntx = (span_x[m] + g.T - 1) / g.T;
    g.nty = (span_y[m] + g.T - 1) / g.T;
    // (rotations per workgroup: the class's -- its LDS is sized for that -- in as few groups as
    // this match's own rotation count needs)
    g.rw = std::min(cg.rw, sr.num_scans);
    g.gmin = (sr.num_scans + g.rw - 1) / g.rw;
    g.gmax = dbg.rt2d_groups > 0 ? std::max(I.fused ? 1 : g.gmin, std::min(dbg.rt2d_groups, sr.num_scans))
                                 : std::max(g.gmin, std::min(sr.num_scans, 32));
    g.target = target;
    g.groups = 0;
    if (I.fused) {                       // the planner's rule (Rt2DTilePrepKernel), on the host
      const long long entries = static_cast<long long>(it.n) * sr.num_scans;
      // (any number of groups: an item takes its rotations in rounds of g.rw)
      long long G = std::max<long long>(1, (entries + target / 2) / target);
      G = std::min<long long>(G, std::min(g.gmax, sr.num_scans));
      g.groups = static_cast<int>(std::max<long long>(G, 1));
    }
    g.cap_s = n_pad + 16 * 4 * g.ntx * g.nty;
    // the grid image: halo + grid, rows of whole 16-byte pieces, one zero row below (what lies
    // right of or below it is zeros by definition: the tile DMA substitutes the zero corner)
    g.gpitch = 2 * ((g.hl + nx + 7) & ~7);
    g.grows = g.ht + ny + 1;
    g.q_bytes = Align16(static_cast<size_t>(g.gpitch) * g.grows);
    // the pooled planes (rt_2d_bounds.h) are part of the image whenever the window allows them,
    // whichever kernel this call runs: a cached image serves both
    g.m2_pitch = g.m2_rows = 0;
    if (side <= 2 * kBoundMaxBlocks) {
      // (room behind the last column a window reads: the bound kernel copies 8-byte pieces)
      g.m2_pitch = (((nx + g.hl) >> 1) + 28 + 3) & ~3;
      g.m2_rows = ((ny + g.ht) >> 1) + kBoundMaxBlocks + 2;
    }
    g.m4_pitch = g.m4_rows = 0;
    if (side <= 2 * kBoundMaxBlocks) {
      // (the sixteen phase planes of the 4 x 4 pooling: four block columns / rows and the 8-byte
      // pieces of the LDS copy behind the last window start)
      g.m4_pitch = (((nx + g.hl) >> 2) + 20 + 3) & ~3;
      g.m4_rows = ((ny + g.ht) >> 2) + 4 + 2;
    }
    g.q8_bytes = g.m4_rows ? Align16(static_cast<size_t>(g.gpitch >> 1) * g.grows) : 0;
    g.image_bytes = g.q_bytes + 4 * static_cast<size_t>(g.m2_pitch) * g.m2_rows +
                    16 * static_cast<size_t>(g.m4_pitch) * g.m4_rows + g.q8_bytes;
    {
      const int nb = launch_nb;                         // (the kernel is instantiated for the largest)
      // (rows are read as three aligned dwords and copied in 8-byte pieces; an odd number of
      // 8-byte pieces: the rows of a wall on different banks)
      g.b_lpb = ((g.T >> 1) + 12 + 4 + 7) & ~7;
      if ((g.b_lpb & 15) == 0) g.b_lpb += 8;
      g.b_lh = (g.T >> 1) + nb + 1;
      // planes | zeros | cloud: what the finish of the match is laid out in afterwards
      g.b_finish_room = 4 * static_cast<size_t>(g.b_lh) * g.b_lpb + ((static_cast<size_t>(nb) * g.b_lpb + 16 + 15) & ~size_t{15}) +
                        8 * static_cast<size_t>(n_pad);
      // | rotations | block sums (later: the summed candidates and their sums) | list | its sums
      g.b_lds = 8 * ((static_cast<size_t>(sr.num_scans) + 1) & ~size_t{1}) +
                4 * BoundSumWords(sr.num_scans * nb * nb) + 20 * size_t{kBoundListCap};   // (+ b_tail_at: below)
      // 4 x 4 blocks: sixteen planes of (T / 4 + nb4 + 1) rows; a row holds T / 4 window starts,
      // nb4 block columns, up to two columns in front (b4_c0 is a multiple of 4) and the seven
      // bytes a shifted pair of dwords reads behind -- in an ODD number of 8-byte pieces (rows on
      // different banks), the planes two banks apart (neighbouring cells lie in different planes)
      const int nb4 = (launch_nb + 1) / 2;
      g.b4_lpb = ((g.T >> 2) + nb4 + 9 + 7) & ~7;
      if ((g.b4_lpb & 15) == 0) g.b4_lpb += 8;
      g.b4_lh = (g.T >> 2) + nb4 + 1;
      g.b4_pstride = g.b4_lh * g.b4_lpb;
      g.b4_pstride += (8 - g.b4_pstride % 128 + 128) % 128;
      g.b4_tail_at = 16 * static_cast<size_t>(g.b4_pstride) + ((static_cast<size_t>(nb4) * g.b4_lpb + 16 + 15) & ~size_t{15}) +
                     8 * static_cast<size_t>(n_pad);
      // the tail kernel's box of the byte image: every window start of the box plus the 4 nb4
      // cells of the blocks, eight bytes for the shifted dword pair; an odd number of 8-byte pieces
      g.b8_lp = (g.T + 4 * nb4 + 8 + 7) & ~7;
      if ((g.b8_lp & 15) == 0) g.b8_lp += 8;
      g.b8_lh = g.T + 4 * nb4;
      g.t4_lds = BoundTail4Lds(g.b8_lp, g.b8_lh, n_pad, sr.num_scans, nb4);
      // | rotations | block sums | discretisation constants (the tail kernel has LDS of its own)
      g.b4_lds = g.b4_tail_at + 8 * ((static_cast<size_t>(sr.num_scans) + 1) & ~size_t{1}) +
                 4 * ((static_cast<size_t>(sr.num_scans) * nb4 * nb4 + 3) & ~size_t{3}) +
                 sizeof(BoundDisc) * static_cast<size_t>(sr.num_scans);
    }
    I.tile_lds = std::max(I.tile_lds, g.lds);
    I.prep_lds = std::max<size_t>(I.prep_lds, 6 * static_cast<size_t>(n_pad) + 512);
    I.lists_total += Align16(2 * static_cast<size_t>(sr.num_scans) * g.cap_s);
    I.hdr_total += static_cast<size_t>(sr.num_scans) * g.ntx * g.nty * 4;
    I.qsum_total += static_cast<size_t>(sr.num_scans) * side * side;
    // (an upper bound of the planner's items: a tile's groups are capped by gmax and by its share
    // of the entries, and the shares of a match's tiles add up to all its entries)
    I.work_cap += I.fused ? g.groups
                          : std::min<long long>(static_cast<long long>(g.ntx) * g.nty * g.gmax,
                                                static_cast<long long>(it.n) * sr.num_scans / target +
                                                    static_cast<long long>(g.ntx) * g.nty * (g.gmin + 1));
    I.max_scans = std::max(I.max_scans, sr.num_scans);
    const int stride = g.H * g.lp;
    I.common_stride = m == 0 ? stride : (I.common_stride == stride ? stride : 0);
    // finish kernel: cells + `group` probability rows + lists within 96 KB
    const size_t fin_fixed = 4 * static_cast<size_t>(n_pad) + 4 * ((static_cast<size_t>(sr.num_scans) + 3) & ~size_t{3}) +
                             12 * static_cast<size_t>(kStage1Cap) + 64 + 128;
    const size_t row = 4 * (static_cast<size_t>(n_pad) + 4);
    const size_t budget = 96 * 1024;
    if (fin_fixed + row > budget) return false;
    I.group = std::min<int>(I.group, static_cast<int>((budget - fin_fixed) / row));
    I.finish_lds = std::max(I.finish_lds, fin_fixed);
  }
  {
    size_t max_row = 0;
    for (int m = 0; m < num; ++m) max_row = std::max<size_t>(max_row, 4 * (static_cast<size_t>((items[m].n + 63) / 64 * 64) + 4));
    I.finish_lds += static_cast<size_t>(I.group) * max_row;
  }
  // ---- block bounds first (rt_2d_bounds.h): every match ONE work item of the bound kernel, two
  // workgroups of 512 threads per CU where their LDS allows.  Not for windows beyond 16 x 16
  // cells (the block columns of a row are one 8-byte read) -- the exhaustive tile kernel keeps
  // those, and stays the parity partner behind the debug switch.
  // From kBoundMinMatches matches per call on: below, the tile kernel's finer work items (a match
  // in four or more, none of which waits for a last one) give the shorter call -- measured on C1,
  // bounds / tiles: 1 match 62 / 52 us, 16: 92 / 74, 128: 114 - 126 / 117 (on 400 x 400 grids
  // 139 - 161 / 106), 256: 144 / 168, 1024: 330 - 357 / 449 - 503 (profiles/r05_c1_bounds.txt).
  // Debug switches: rt2d_bounds = 1 always, rt2d_no_bounds never.
  I.bounds = I.fused && !dbg.rt2d_no_bounds && (I.batch_matches >= kBoundMinMatches || dbg.rt2d_bounds);
  // (round 6) the tail of a match -- phases B, C and the finish: ~30 us of dependent round trips
  // during which a workgroup's 70 KB of LDS and its place on the CU did nothing -- in a kernel of
  // its own behind the bound kernel (Rt2DBoundTailKernel: 33 KB, every match of the part in flight
  // at once).  rt2d_bounds_fused = 1: one kernel as in round 5 (the parity partner; verify mode).
  I.split = I.bounds && !dbg.rt2d_bounds_verify && !dbg.rt2d_bounds_fused;
  // (round 6) blocks of 4 x 4 translations first, their survivors refined by the tail kernel:
  // rt2d_bounds_level = 2 keeps the 2 x 2 blocks as the first level (the parity partner)
  I.level4 = I.bounds && !dbg.rt2d_bounds_fused && dbg.rt2d_bounds_level != 2;
  for (int m = 0; m < num && I.bounds; ++m) {
    const int side = 2 * search[m].nl + 1;
    // (the fused bound kernel finishes a match itself, in the LDS of its planes and cloud -- a
    // small box leaves less than the finish needs: the rest of the layout moves back)
    geo[m].b_tail_at = I.split ? geo[m].b_finish_room : std::max(geo[m].b_finish_room, Align16(I.finish_lds));
    I.tail_lds = std::max(I.tail_lds, BoundTailLds((items[m].n + 63) / 64 * 64, search[m].num_scans, launch_nb));
    geo[m].b_lds += geo[m].b_tail_at;
    I.bounds = side <= 2 * kBoundMaxBlocks && geo[m].b_lds <= 150 * size_t{1024};
    I.bound_nb = std::max(I.bound_nb, (side + 1) / 2);
    I.bound_lds = std::max(I.bound_lds, geo[m].b_lds);
    I.bound4_lds = std::max(I.bound4_lds, geo[m].b4_lds);
    I.tail4_lds = std::max(I.tail4_lds, geo[m].t4_lds);
  }
  I.level4 = I.level4 && I.bounds && I.bound4_lds <= 150 * size_t{1024};
  if (I.level4) { I.split = true; I.bound_lds = I.bound4_lds; }
  if (I.bounds) {
    // rotation groups per match: about two items per CU over the whole call (an item stages the
    // match's planes and cloud: ~4 us), one per match in large batches
    const int per_cu = I.bound_lds <= 78 * size_t{1024} ? 2 : 1;
    const int slots = std::max(1, per_cu * cus / I.share);
    I.work_cap = 0;
    for (int m = 0; m < num; ++m) {
      const int G = dbg.rt2d_groups > 0 ? dbg.rt2d_groups : slots / num;
      geo[m].groups = std::max(1, std::min(G, search[m].num_scans));
      I.work_cap += geo[m].groups;
    }
  }
  CMX_REQUIRE(I.work_cap < (1ll << 24) && num <= 65535, "too many matches in one batch");
  {
    I.table_at.resize(num);
    size_t at = 0;
    for (int m = 0; m < num; ++m) { I.table_at[m] = at; at += search[m].num_scans; }
    I.tables.resize(at);
    // (libm, ~5 ns per pair, a third of a match's host time and nothing shared: blocks of >= 32
    // matches over the host pool)
    // matches, six blocks at most, over the host pool)
    const int blocks = std::min(6, (num + 31) / 32);
    ParallelFor(blocks, dbg.rt2d_tables_serial ? (1 << 30) : 3, [&](int b) {
      const int begin = static_cast<int>(static_cast<long long>(num) * b / blocks);
      const int end = static_cast<int>(static_cast<long long>(num) * (b + 1) / blocks);
      for (int m = begin; m < end; ++m)
        FillRotationTable(search[m].step, search[m].na, I.tables.data() + I.table_at[m]);
    });
  }
  I.tile_grid = static_cast<int>(std::max<long long>(1, std::min<long long>(tile_slots, I.work_cap)));
  if (I.bounds) {
    const int per_cu = I.bound_lds <= 78 * size_t{1024} ? 2 : 1;
    // (the whole chip even when the call is one part of a batch: the workgroups are persistent and
    // two of them share a CU -- a part whose grid is its share of the CUs leaves a wavefront per
    // SIMD short of hiding the LDS latency of phase A whenever the parts do not overlap)
    I.tile_grid = static_cast<int>(std::max<long long>(1, std::min<long long>(per_cu * cus, I.work_cap)));
  }
  return true;
}

void Rt2DTileCall::Enqueue(hipStream_t on_stream) {
  Impl& I = *impl_;
  const DebugOptions& dbg = Debug();
  const auto t_enter = std::chrono::steady_clock::now();
  const auto lap_us = [&]() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_enter).count();
  };
  double t_images = 0, t_params = 0, t_upload = 0;
  const Rt2DItem* items = I.items;
  const Rt2DSearch* search = I.search;
  const int num = I.num, device = I.device, rpl = I.rpl;
  const std::vector<TileGeometry>& geo = I.geo;
  // ---- staging: params | per item: xyz, rotations, cells | counters | misc ------------------
  struct Off { size_t xyz, rot, cells; };
  std::vector<Off> off(num);
  size_t in_bytes = Align16(sizeof(Rt2DTileParams) * num);
  for (int m = 0; m < num; ++m) {
    const Rt2DItem& it = items[m];
    off[m].xyz = in_bytes;
    off[m].rot = off[m].xyz + (it.device_xyz ? 0 : Align16(3 * sizeof(float) * it.n));
    off[m].cells = off[m].rot + Align16(sizeof(float2) * search[m].num_scans);
    in_bytes = off[m].cells + (it.device_cells ? 0 : Align16(sizeof(uint16_t) * static_cast<size_t>(it.limits->num_x_cells) * it.limits->num_y_cells));
  }
  const size_t off_counters = in_bytes;               // [0] work items, [1] next item (zeroed)
  in_bytes += 64;
  const size_t off_tickets = in_bytes;                // block bounds: a ticket counter per match (zeroed)
  if (I.bounds) in_bytes += Align16(sizeof(int) * static_cast<size_t>(num));
  const size_t off_work = in_bytes;                   // fused: the work items, listed here
  if (I.fused) in_bytes += Align16(3 * sizeof(int4) * static_cast<size_t>(I.work_cap));
  const size_t off_misc = in_bytes;
  in_bytes += Align16(sizeof(unsigned) * 128 * static_cast<size_t>(num));

  I.ws.reset(new WorkspaceLease(device));
  WorkspaceLease& ws = *I.ws;
  if (on_stream) ws->stream = on_stream;      // (a lease sets its workspace's stream anew every time)
  char* h_in = ws->pinned[0].ReserveAs<char>(in_bytes);
  char* d_in = ws->dev[0].ReserveAs<char>(in_bytes);
  uint16_t* d_lists = I.fused ? nullptr : reinterpret_cast<uint16_t*>(ws->dev[1].Reserve(I.lists_total + 64));
  uint32_t* d_hdr = I.fused ? nullptr : ws->dev[2].ReserveAs<uint32_t>(I.hdr_total + 16);
  int* d_qsum = ws->dev[3].ReserveAs<int>(I.qsum_total + 16);
  I.d_overflow = ws->dev[4].ReserveAs<unsigned>(static_cast<size_t>(num) * 2 * (kFinalistCap - kFinalistHead));
  int4* d_work = I.fused ? reinterpret_cast<int4*>(d_in + off_work)
                         : ws->dev[8].ReserveAs<int4>(static_cast<size_t>(I.work_cap) + 1);
  I.h_misc = ws->pinned[1].ReserveAs<unsigned>(static_cast<size_t>(num) * 128);
  unsigned* d_misc = reinterpret_cast<unsigned*>(d_in + off_misc);
  int* d_counters = reinterpret_cast<int*>(d_in + off_counters);
  std::memset(h_in + off_counters, 0, 64);
  if (I.bounds) std::memset(h_in + off_tickets, 0, Align16(sizeof(int) * static_cast<size_t>(num)));
  // (block sums of matches bounded by several workgroups; read only where groups > 1)
  std::vector<int> ub_at(num, 0);
  size_t ub_total = 0;
  for (int m = 0; m < num && I.bounds; ++m) {
    ub_at[m] = static_cast<int>(ub_total);
    const int nb_level = I.level4 ? (I.bound_nb + 1) / 2 : I.bound_nb;
    ub_total += static_cast<size_t>(search[m].num_scans) * nb_level * nb_level;
  }
  int* d_ub = I.bounds ? ws->dev[6].ReserveAs<int>(ub_total + 16) : nullptr;
  if (I.fused) {
    int4* h_work = reinterpret_cast<int4*>(h_in + off_work);
    int count = 0;
    for (int m = 0; m < num; ++m) {
      const uintptr_t xyz = reinterpret_cast<uintptr_t>(
          items[m].device_xyz ? items[m].device_xyz : reinterpret_cast<const float*>(d_in + off[m].xyz));
      const uintptr_t rot = reinterpret_cast<uintptr_t>(d_in + off[m].rot);
      for (int g = 0; g < geo[m].groups; ++g, ++count) {
        h_work[3 * count] = make_int4(m, 0, g, geo[m].groups);
        h_work[3 * count + 1] = make_int4(static_cast<int>(static_cast<uint32_t>(xyz)), static_cast<int>(static_cast<uint32_t>(xyz >> 32)),
                                          static_cast<int>(static_cast<uint32_t>(rot)), static_cast<int>(static_cast<uint32_t>(rot >> 32)));
        h_work[3 * count + 2] = make_int4(items[m].n, search[m].num_scans, 0, 0);
      }
    }
    reinterpret_cast<int*>(h_in + off_counters)[0] = count;
  }
  const int timeline_tile_slots = I.tile_grid * 4;
  if (dbg.timeline) {
    const size_t bytes = (static_cast<size_t>(timeline_tile_slots) + num) * kTimelineStamps * 8;
    I.d_timeline = static_cast<unsigned long long*>(ws->dev[7].Reserve(bytes));
    CMX_HIP(hipMemsetAsync(I.d_timeline, 0, bytes, ws->stream));
  }

  // ---- grid images: a resident grid keeps its own (two buffers per grid); everything else is
  // built into scratch by this call ---------------------------------------------------------
  size_t scratch_images = 0;
  std::vector<uint16_t*> image_of(num, nullptr);
  std::vector<long long> scratch_at(num, -1);       // offset in this call's scratch (-1: cached)
  std::vector<int> build_image(num, 0), same_as(num, -1);
  I.holds.stream = ws->stream;
  std::unordered_map<GridKey, int, GridKeyHash> first_of;
  first_of.reserve(static_cast<size_t>(num));
  for (int m = 0; m < num; ++m) {
    const TileGeometry& g = geo[m];
    Rt2DImageCache* c = items[m].image_cache;
    {                                             // the same grid earlier in this batch
      const GridKey key{items[m].cells, items[m].device_cells, c, items[m].grid_version,
                        search[m].nl, g.gpitch, g.grows};
      const auto found = first_of.find(key);
      if (found != first_of.end()) same_as[m] = found->second;
      else first_of.emplace(key, m);
    }
    if (same_as[m] >= 0) continue;
    int buffer = -1;
    bool build = true;
    if (c && items[m].device_cells && !dbg.rt2d_no_image_cache) {
      buffer = c->Acquire(items[m].grid_version, search[m].nl, g.gpitch, g.grows, g.image_bytes, &build);
      if (buffer >= 0) {
        I.holds.holds.push_back(CacheHold{c, buffer, build, items[m].grid_version, search[m].nl, g.gpitch, g.grows});
        image_of[m] = c->buffer[buffer].image;
      }
    }
    if (buffer < 0) {
      scratch_at[m] = static_cast<long long>(scratch_images);
      scratch_images += Align16(g.image_bytes);
      build = true;
    }
    build_image[m] = build ? 1 : 0;
  }
  char* d_images = static_cast<char*>(ws->dev[5].Reserve(scratch_images + 64));
  for (int m = 0; m < num; ++m)
    if (scratch_at[m] >= 0) image_of[m] = reinterpret_cast<uint16_t*>(d_images + scratch_at[m]);
  for (int m = 0; m < num; ++m)
    if (same_as[m] >= 0) image_of[m] = image_of[same_as[m]];

  t_images = lap_us();
  // ---- parameters -------------------------------------------------------------------------------
  Rt2DTileParams* h_params = reinterpret_cast<Rt2DTileParams*>(h_in);
  {
    size_t lists_at = 0, hdr_at = 0, qsum_at = 0;
    std::vector<size_t> lists_off(num), hdr_off(num), qsum_off(num);
    for (int m = 0; m < num; ++m) {
      const TileGeometry& g = geo[m];
      const int side = 2 * search[m].nl + 1;
      lists_off[m] = lists_at; lists_at += Align16(2 * static_cast<size_t>(search[m].num_scans) * g.cap_s);
      hdr_off[m] = hdr_at; hdr_at += static_cast<size_t>(search[m].num_scans) * g.ntx * g.nty * 4;
      qsum_off[m] = qsum_at; qsum_at += static_cast<size_t>(search[m].num_scans) * side * side;
    }
    const cmx_rt_options* options = I.options;
    unsigned long long* d_timeline = I.d_timeline;
    unsigned* d_overflow = I.d_overflow;
    // (resident grids and clouds: a fraction of a microsecond per match, and dispatching to the
    // host pool costs ~25 us; host grids are copied into the staging buffer here: worth the pool)
    bool copies = false;
    for (int m = 0; m < num; ++m) copies = copies || !items[m].device_cells;
    ParallelFor(num, copies ? 16 : (dbg.rt2d_host_par > 0 ? dbg.rt2d_host_par : 4096), [&](int m) {
      const Rt2DItem& it = items[m];
      const Rt2DSearch& sr = search[m];
      const TileGeometry& g = geo[m];
      const size_t cell_count = static_cast<size_t>(it.limits->num_x_cells) * it.limits->num_y_cells;
      if (!it.device_xyz) std::memcpy(h_in + off[m].xyz, it.xyz, 3 * sizeof(float) * it.n);
      std::memcpy(h_in + off[m].rot, I.tables.data() + I.table_at[m], sizeof(float2) * sr.num_scans);
      if (!it.device_cells) std::memcpy(h_in + off[m].cells, it.cells, sizeof(uint16_t) * cell_count);
      Rt2DTileParams P{};
      P.cells = it.device_cells ? it.device_cells : reinterpret_cast<const uint16_t*>(d_in + off[m].cells);
      P.nx = it.limits->num_x_cells; P.ny = it.limits->num_y_cells;
      P.res = it.limits->resolution; P.max_x = it.limits->max_x; P.max_y = it.limits->max_y;
      P.inv_res = 1.0 / P.res;
      P.tx = static_cast<float>(it.initial->x);
      P.ty = static_cast<float>(it.initial->y);
      P.init_qw = sr.q0w; P.init_qz = sr.q0z;
      P.nl = sr.nl; P.num_scans = sr.num_scans; P.num_angular = sr.na;
      P.step = sr.step;
      P.wt = options->translation_delta_cost_weight;
      P.wr = options->rotation_delta_cost_weight;
      P.scan_rot = reinterpret_cast<const float2*>(d_in + off[m].rot);
      P.xyz = it.device_xyz ? it.device_xyz : reinterpret_cast<const float*>(d_in + off[m].xyz);
      P.n = it.n; P.n_pad = (it.n + 63) / 64 * 64;
      P.B = g.B; P.H = g.H; P.rpl = rpl; P.hl = g.hl; P.ht = g.ht;
      P.qimage = image_of[m]; P.gpitch = g.gpitch; P.grows = g.grows; P.image_build = build_image[m];
      P.box_x0 = g.box_x0; P.box_y0 = g.box_y0; P.T = g.T;
      P.T_magic = static_cast<unsigned>((0x100000000ull + g.T - 1) / g.T);
      P.ntx = g.ntx; P.nty = g.nty; P.lp = g.lp; P.th_img = g.th_img;
      P.tile_image_bytes = g.tile_image_bytes; P.null_addr = g.th_img * g.lp;
      P.lists = d_lists ? d_lists + lists_off[m] / 2 : nullptr; P.cap_s = g.cap_s;
      P.hdr = d_hdr ? d_hdr + hdr_off[m] : nullptr;
      P.fused = I.fused ? 1 : 0;
      P.gmin = g.gmin; P.gmax = g.gmax; P.target = g.target; P.rw = g.rw;
      P.list_lds = g.list_lds; P.task_cap = g.task_cap;
      P.flush_atomic = g.ntx * g.nty > 1 ? 1 : 0;
      P.qsum = d_qsum + qsum_off[m];
      P.misc = d_misc + static_cast<size_t>(m) * 128;
      P.overflow = d_overflow + static_cast<size_t>(m) * 2 * (kFinalistCap - kFinalistHead);
      P.stage = P.misc + 126;
      P.timeline = d_timeline;
      P.timeline_finish_base = timeline_tile_slots;
      P.m2 = g.m2_rows ? reinterpret_cast<const uint8_t*>(image_of[m]) + g.q_bytes : nullptr;
      P.m2_pitch = g.m2_pitch; P.m2_rows = g.m2_rows;
      P.b_c0 = (g.box_x0 >> 1) & ~3; P.b_r0 = g.box_y0 >> 1;
      P.b_lpb = g.b_lpb; P.b_lh = g.b_lh; P.b_tail_at = static_cast<int>(g.b_tail_at);
      P.m4 = g.m4_rows ? P.m2 + 4 * static_cast<size_t>(g.m2_pitch) * g.m2_rows : nullptr;
      P.m4_pitch = g.m4_pitch; P.m4_rows = g.m4_rows;
      P.b4_c0 = (g.box_x0 >> 2) & ~3; P.b4_r0 = g.box_y0 >> 2;
      P.b4_lpb = g.b4_lpb; P.b4_lh = g.b4_lh; P.b4_pstride = g.b4_pstride;
      P.b4_tail_at = static_cast<int>(g.b4_tail_at);
      P.b4_ub_at = ub_at[m];
      P.q8 = g.q8_bytes ? const_cast<uint8_t*>(P.m4) + 16 * static_cast<size_t>(g.m4_pitch) * g.m4_rows : nullptr;
      P.b8_lp = g.b8_lp; P.b8_lh = g.b8_lh;
      P.bstat = P.misc + 124;
      P.b_verify = dbg.rt2d_bounds_verify ? 1 : 0;
      P.b_ub_at = ub_at[m];
      h_params[m] = P;
    });
  }
  // (the stage counters ride in the last words of a match's slot: the finalist head must stop short)
  static_assert(2 + 2 * kFinalistHead <= 124, "a match's head, bound and stage counters share 128 words");
  t_params = lap_us();
  // (the matches' result words -- 512 bytes each, the last region of the buffer -- are zeroed by
  // the copy kernel itself: 40 % of the upload did not have to cross the bus)
  SmallCopyAsync(d_in, h_in, off_misc, /*to_device=*/true, ws->stream, d_in + off_misc, in_bytes - off_misc);
  t_upload = lap_us();
  const Rt2DTileParams* d_params = reinterpret_cast<const Rt2DTileParams*>(d_in);

  RecordEvent(ws->ev_begin, ws->stream);
  {
    bool any_build = false;
    size_t max_vecs = 0, max_words = 0, max_words4 = 0;
    int max_xe = 0, max_ye = 0;
    for (int m = 0; m < num; ++m) {
      if (!build_image[m]) continue;
      any_build = true;
      max_vecs = std::max(max_vecs, (static_cast<size_t>(geo[m].gpitch) * geo[m].grows) >> 4);
      max_words = std::max(max_words, static_cast<size_t>(geo[m].m2_pitch) * geo[m].m2_rows);
      max_words4 = std::max(max_words4, 4 * static_cast<size_t>(geo[m].m4_pitch) * geo[m].m4_rows);
      max_xe = std::max(max_xe, std::max(geo[m].gpitch >> 1, std::max(2 * geo[m].m2_pitch, 4 * geo[m].m4_pitch)));
      max_ye = std::max(max_ye, std::max(geo[m].grows, std::max(2 * geo[m].m2_rows, 4 * geo[m].m4_rows)));
    }
    if (any_build && !dbg.rt2d_image_kernels) {
      // (round 6) every derived image of a grid in ONE launch: Rt2DImageKernel
      const int tiles_x = DivUp(max_xe, kImageTileX), tiles_y = DivUp(max_ye, kImageTileY);
      Rt2DImageKernel<<<dim3(tiles_x * tiles_y, num), 256, 0, ws->stream>>>(d_params, tiles_x);
    } else if (any_build) {
      Rt2DQuantKernel<<<dim3(DivUp(max_vecs, 256), num), 256, 0, ws->stream>>>(d_params);
      if (max_words)
        Rt2DPoolKernel<<<dim3(DivUp(max_words, 256), num), 256, 0, ws->stream>>>(d_params);
      if (max_words4)
        Rt2DPool4Kernel<<<dim3(DivUp(max_words4, 256), num), 256, 0, ws->stream>>>(d_params);
    }
  }
  if (!I.fused)
    Rt2DTilePrepKernel<<<dim3(I.max_scans, num), 256, I.prep_lds, ws->stream>>>(
        d_params, d_counters, d_work, static_cast<int>(I.work_cap));
  RecordEvent(ws->ev_k0, ws->stream);
  {
    const auto launch = [&](auto kernel) {
      OptInLds(reinterpret_cast<const void*>(kernel), device, 160 * 1024 - 1024);   // (minus the static words)
      if (dbg.host_trace) {
        int resident = 0;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, kernel, I.tile_threads, I.tile_lds);
        fprintf(stderr, "[cmx host] rt2d tile kernel: %d workgroups of %d threads, %zu B of LDS each, "
                        "%d resident per CU\n", I.tile_grid, I.tile_threads, I.tile_lds, resident);
      }
      kernel<<<static_cast<unsigned>(I.tile_grid), I.tile_threads, I.tile_lds, ws->stream>>>(
          d_params, d_work, d_counters, d_counters + 1, I.fused ? 3 : 1);
    };
    const auto launch_bounds = [&](auto kernel) {
      OptInLds(reinterpret_cast<const void*>(kernel), device, 160 * 1024 - 4096);
      if (dbg.host_trace) {
        int resident = 0;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, kernel, kBoundThreads, I.bound_lds);
        fprintf(stderr, "[cmx host] rt2d bound kernel<%d>: %d workgroups of %d threads, %zu B of LDS each, "
                        "%d resident per CU\n", I.bound_nb, I.tile_grid, kBoundThreads, I.bound_lds, resident);
      }
      kernel<<<static_cast<unsigned>(I.tile_grid), kBoundThreads, I.bound_lds, ws->stream>>>(
          d_params, d_work, d_counters, d_counters + 1, reinterpret_cast<int*>(d_in + off_tickets), d_ub,
          I.group, I.h_misc, I.split ? 1 : 0);
    };
    const auto launch_tail = [&](auto kernel) {
      OptInLds(reinterpret_cast<const void*>(kernel), device, 160 * 1024 - 4096);
      kernel<<<static_cast<unsigned>(num), I.level4 ? kBoundTail4Threads : kBoundTailThreads,
               I.level4 ? I.tail4_lds : I.tail_lds, ws->stream>>>(d_params, d_ub, I.h_misc);
    };
    const int common_stride = I.common_stride;
    if (I.level4) {
      // (ev_k1 between the two: the bound kernel's own span)
      switch ((I.bound_nb + 1) / 2) {
        case 1: launch_bounds(Rt2DBoundKernel<1, 2>); RecordEvent(ws->ev_k1, ws->stream); launch_tail(Rt2DBoundTail4Kernel<1>); break;
        case 2: launch_bounds(Rt2DBoundKernel<2, 2>); RecordEvent(ws->ev_k1, ws->stream); launch_tail(Rt2DBoundTail4Kernel<2>); break;
        case 3: launch_bounds(Rt2DBoundKernel<3, 2>); RecordEvent(ws->ev_k1, ws->stream); launch_tail(Rt2DBoundTail4Kernel<3>); break;
        default: launch_bounds(Rt2DBoundKernel<4, 2>); RecordEvent(ws->ev_k1, ws->stream); launch_tail(Rt2DBoundTail4Kernel<4>); break;
      }
    } else if (I.bounds) {
      switch (I.bound_nb) {
        case 1: launch_bounds(Rt2DBoundKernel<1, 1>); break;
        case 2: launch_bounds(Rt2DBoundKernel<2, 1>); break;
        case 3: launch_bounds(Rt2DBoundKernel<3, 1>); break;
        case 4: launch_bounds(Rt2DBoundKernel<4, 1>); break;
        case 5: launch_bounds(Rt2DBoundKernel<5, 1>); break;
        case 6: launch_bounds(Rt2DBoundKernel<6, 1>); break;
        case 7: launch_bounds(Rt2DBoundKernel<7, 1>); break;
        default: launch_bounds(Rt2DBoundKernel<8, 1>); break;
      }
      if (I.split) {
        switch (I.bound_nb) {
          case 1: launch_tail(Rt2DBoundTailKernel<1>); break;
          case 2: launch_tail(Rt2DBoundTailKernel<2>); break;
          case 3: launch_tail(Rt2DBoundTailKernel<3>); break;
          case 4: launch_tail(Rt2DBoundTailKernel<4>); break;
          case 5: launch_tail(Rt2DBoundTailKernel<5>); break;
          case 6: launch_tail(Rt2DBoundTailKernel<6>); break;
          case 7: launch_tail(Rt2DBoundTailKernel<7>); break;
          default: launch_tail(Rt2DBoundTailKernel<8>); break;
        }
      }
    } else if (I.d_timeline) {               // the instrumented instantiations (runtime row stride)
      if (rpl == 1) launch(Rt2DTileKernel<1, 0, true>);
      else if (rpl == 2) launch(Rt2DTileKernel<2, 0, true>);
      else if (rpl == 3) launch(Rt2DTileKernel<3, 0, true>);
      else if (rpl == 4) launch(Rt2DTileKernel<4, 0, true>);
      else if (rpl <= 6) launch(Rt2DTileKernel<6, 0, true>);
      else launch(Rt2DTileKernel<8, 0, true>);
    } else if (rpl == 1) launch(Rt2DTileKernel<1, 0, false>);
    else if (rpl == 2 && common_stride == 8 * 288) launch(Rt2DTileKernel<2, 8 * 288, false>);
    else if (rpl == 2 && common_stride == 8 * 224) launch(Rt2DTileKernel<2, 8 * 224, false>);
    else if (rpl == 2 && common_stride == 8 * 480) launch(Rt2DTileKernel<2, 8 * 480, false>);
    else if (rpl == 2) launch(Rt2DTileKernel<2, 0, false>);
    else if (rpl == 3) launch(Rt2DTileKernel<3, 0, false>);
    else if (rpl == 4) launch(Rt2DTileKernel<4, 0, false>);
    else if (rpl <= 6) launch(Rt2DTileKernel<6, 0, false>);
    else launch(Rt2DTileKernel<8, 0, false>);
  }
  if (!I.level4) RecordEvent(ws->ev_k1, ws->stream);
  // (the bound kernel has finished its matches itself; in its verify mode it leaves every sum)
  if (!I.bounds || dbg.rt2d_bounds_verify) {
    if (I.level4) {     // (verify mode of the 4 x 4 level: byte sums)
      OptInLds(reinterpret_cast<const void*>(Rt2DFinishKernel<kQ8Shift>), device, 128 * 1024);
      Rt2DFinishKernel<kQ8Shift><<<num, kFinishThreads, I.finish_lds, ws->stream>>>(d_params, I.group, I.h_misc);
    } else {
      OptInLds(reinterpret_cast<const void*>(Rt2DFinishKernel<kQShift>), device, 128 * 1024);
      Rt2DFinishKernel<kQShift><<<num, kFinishThreads, I.finish_lds, ws->stream>>>(d_params, I.group, I.h_misc);
    }
  }
  CMX_HIP(hipGetLastError());
  RecordEvent(ws->ev_end, ws->stream);
  I.enqueued = true;
  if (dbg.host_trace)
    fprintf(stderr, "[cmx host] rt2d enqueue(%d): workspace + images %.0f, parameters %.0f, upload call %.0f, "
                    "launches %.0f us (%zu bytes up)\n", num, t_images, t_params, t_upload, lap_us(), in_bytes);
}

bool Rt2DTileCall::Collect(cmx_match_stats* stats, std::vector<int>* redo) {
  Impl& I = *impl_;
  CMX_REQUIRE(I.enqueued, "internal error: Collect before Enqueue");
  WorkspaceLease& ws = *I.ws;
  const int num = I.num;
  const Rt2DItem* items = I.items;
  const Rt2DSearch* search = I.search;
  const unsigned* h_misc = I.h_misc;
  const auto t_enter = std::chrono::steady_clock::now();
  CMX_HIP(hipStreamSynchronize(ws->stream));
  const double t_wait = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_enter).count();
  I.synced = true;
  I.holds.built = true;
  if (I.d_timeline) {
    ReportTimeline(I.bounds ? "Rt2DBoundKernel" : "Rt2DTileKernel", I.d_timeline, I.tile_grid * 4, ws->stream);
    ReportTimeline("Rt2DFinishKernel", I.d_timeline + static_cast<size_t>(I.tile_grid) * 4 * kTimelineStamps,
                   num, ws->stream);
  }
  // Matches the bulk path could not decide -- a flat score landscape (more blocks or candidates
  // within the bounds than the lists hold), a point outside the predicted box -- are left to the
  // caller, match by match (round 6: until then one of them sent its whole batch to the
  // per-candidate kernels); without a list to put them on, the whole call is.
  std::vector<char> undecided(num, 0);
  for (int m = 0; m < num; ++m) {
    const unsigned count = h_misc[static_cast<size_t>(m) * 128 + 1];
    CMX_REQUIRE(!(h_misc[static_cast<size_t>(m) * 128] & kBoundViolated),
                "internal error: rt2d block bound below one of its candidates (rt2d_bounds_verify)");
    bool bad = (h_misc[static_cast<size_t>(m) * 128] & kBoundFlat) != 0;   // (flat landscape)
    if (h_misc[static_cast<size_t>(m) * 128] & kOutOfBox) {
      fprintf(stderr, "[cmx] rt2d: a point fell outside the predicted box of match %d; it is "
                      "repeated on the per-candidate kernels\n", m);
      bad = true;
    }
    if (count > static_cast<unsigned>(kFinalistCap)) bad = true;    // kFlat, overflow
    if (bad) {
      if (!redo) return false;
      undecided[m] = 1;
      redo->push_back(m);
    }
  }
  // Exact weighting and first maximum, item by item: on the host pool (exp, hypot, a sort of a
  // handful of pairs); a match with more finalists than its head holds fetches the rest first.
  std::vector<std::vector<unsigned>> extra(num);
  for (int m = 0; m < num; ++m) {
    if (undecided[m]) continue;
    const long long count = h_misc[static_cast<size_t>(m) * 128 + 1];
    CMX_REQUIRE(count >= 1, "internal error: no candidate collected");
    if (count > kFinalistHead) {
      extra[m].resize(2 * (count - kFinalistHead));
      CMX_HIP(hipMemcpyAsync(extra[m].data(), I.d_overflow + static_cast<size_t>(m) * 2 * (kFinalistCap - kFinalistHead),
                             sizeof(unsigned) * extra[m].size(), hipMemcpyDeviceToHost, ws->stream));
      CMX_HIP(hipStreamSynchronize(ws->stream));
    }
  }
  const cmx_rt_options* options = I.options;
  ParallelFor(num, Debug().rt2d_host_par > 0 ? Debug().rt2d_host_par : 4096, [&](int m) {
    if (undecided[m]) return;
    const unsigned* head = h_misc + static_cast<size_t>(m) * 128;
    const long long count = head[1];
    const long long in_head = std::min<long long>(count, kFinalistHead);
    std::pair<int, float> small[kFinalistHead];
    std::vector<std::pair<int, float>> big;
    std::pair<int, float>* finalists = small;
    if (count > kFinalistHead) { big.resize(count); finalists = big.data(); }
    for (long long i = 0; i < count; ++i) {
      const unsigned* pair = i < in_head ? head + 2 + 2 * i : extra[m].data() + 2 * (i - in_head);
      float v;
      std::memcpy(&v, &pair[1], sizeof(float));
      finalists[i] = {static_cast<int>(pair[0]), v};
    }
    std::sort(finalists, finalists + count);
    Rt2DFinishOnHost(options, items[m], search[m], finalists, static_cast<size_t>(count));
  });
  cmx_match_stats total{};
  for (int m = 0; m < num; ++m) {
    if (undecided[m]) continue;               // (counted by whoever repeats it)
    const unsigned* head = h_misc + static_cast<size_t>(m) * 128;
    const long long cands = static_cast<long long>(search[m].num_scans) * (2 * search[m].nl + 1) * (2 * search[m].nl + 1);
    total.candidates_scored += cands;
    // (what the device summed: the whole search space, or -- block bounds -- one bound per
    // 2 x 2 block of translations plus the four candidates of the blocks that reach the bound)
    // (head[124]: 2 x 2 blocks summed, four candidates each -- or, 4 x 4 level, the candidates summed)
    total.coarse_candidates += I.bounds ? static_cast<long long>(head[125]) + (I.level4 ? 1ll : 4ll) * head[124] : cands;
    total.num_scans += search[m].num_scans;
    total.refined_candidates += head[126];        // candidates re-summed with exact integers
    total.finalists += head[127] & 0xffffu;       // candidates scored with the reference's f32 chain
  }
  if (stats) {
    float ms = 0.f;
    ms = ElapsedMs(ws->ev_begin, ws->ev_end);
    total.device_ms = ms;
    ms = ElapsedMs(ws->ev_k0, ws->ev_k1);
    total.dominant_kernel_ms = ms;
    *stats = total;
  }
  if (Debug().host_trace)
    fprintf(stderr, "[cmx host] rt2d collect(%d): wait %.0f, all %.0f us\n", num, t_wait,
            std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_enter).count());
  return true;
}

}  // namespace cmx
