// RealTimeCorrelativeScanMatcher3D::Match on gfx950, plus the dense-brick
// helpers shared with the fast 3D matcher.
//
// Reference: SM3/real_time_correlative_scan_matcher_3d.cc:34-53 (Match),
// :55-95 (GenerateExhaustiveSearchTransforms), :97-114 (ScoreCandidate).
//
// Parity notes
//   * Candidate = init.cast<float>() * Rigid3f(t, q) (renormalised product).
//     The (2A+1)^3 rotations and (2L+1)^3 translations are composed on the
//     host with the reference's f32/f64 operation order (sin/cos/atan2 from
//     libm); the device only performs IEEE +,-,*,/ on them.
//   * The reference transforms every point by every candidate and sums the N
//     probabilities sequentially in f32.  One thread per candidate does exactly
//     that, so the unweighted score is bit-identical; a wave holds 64
//     translations of one rotation, so the point load and the rotation are
//     wave-uniform and neighbouring lanes read neighbouring voxels.
//   * exp() weighting and the strict-'>' first-maximum rule are finished on the
//     host for the candidates within 1e-5 (relative) of the device maximum.
#include <algorithm>
#include <functional>
#include <type_traits>

#include "scan_matching_3d.h"

namespace cmx {
namespace {

__global__ void ScatterVoxelsKernel(const cmx_voxel* __restrict__ voxels, long long n, Brick b,
                                    int bytes_per_cell) {
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const cmx_voxel v = voxels[i];
  const size_t off = (static_cast<size_t>(v.z - b.lo_z) * b.ny + (v.y - b.lo_y)) * b.nx +
                     (v.x - b.lo_x);
  if (bytes_per_cell == 2) {
    static_cast<uint16_t*>(const_cast<void*>(b.cells))[off] = v.value;
  } else {
    // ConvertToPrecomputationGrid (SM3/precomputation_grid_3d.cc:49-62).
    const float kMinP = 0.1f;
    const float kMaxP = 1.f - kMinP;
    int value = LRoundF32((ValueToProbabilityDev(v.value) - kMinP) * (255.f / (kMaxP - kMinP)));
    value = min(max(value, 0), 255);
    static_cast<uint8_t*>(const_cast<void*>(b.cells))[off] = static_cast<uint8_t>(value);
  }
}

// Dense f32 probability brick with a one-cell border of kMinProbability: a voxel
// index clamped to the border reads what HybridGrid::GetProbability returns for any
// cell outside the stored box (value 0 -> kMinProbability, hybrid_grid.h:521-523).
struct PaddedBrick {
  const float* cells;   // [(nz+2)][(ny+2)][(nx+2)]
  int lo_x, lo_y, lo_z; // index of padded cell (1,1,1)
  int nx, ny, nz;
};

__global__ void FillFloatKernel(float* __restrict__ out, size_t n, float value) {
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) out[i] = value;
}

__global__ void ScatterProbabilitiesKernel(const cmx_voxel* __restrict__ voxels, long long n,
                                           PaddedBrick b, float* __restrict__ out) {
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const cmx_voxel v = voxels[i];
  const size_t off = (static_cast<size_t>(v.z - b.lo_z + 1) * (b.ny + 2) + (v.y - b.lo_y + 1)) *
                         (b.nx + 2) + (v.x - b.lo_x + 1);
  out[off] = ValueToProbabilityDev(v.value);
}

struct Rt3DParams {
  PaddedBrick grid;
  float resolution, inv_resolution;
  int num_translations, num_rotations, side_t, side_r;
  const float4* rotation;     // [R] candidate rotation (x,y,z,w) = normalized(init.q * q_r)
  const float4* translation;  // [T] candidate translation (xyz) = init.q * t_c + init.t; w = |t_c|
  const float* rotation_angle;  // [R] GetAngle(transform)
  double wt, wr;
};

// lround(c / resolution) (HybridGrid::GetCellIndex, hybrid_grid.h:428-433) without the
// IEEE division in the common case.  q0 = c * RN(1/res) differs from the correctly
// rounded quotient qe by less than |q0| * 2^-21; when q0 is further than |q0| * 2^-20
// from every half-integer, qe lies strictly inside (n - 0.5, n + 0.5) with n = rint(q0),
// so lround(qe) == n.  Otherwise (about |q0| * 2^-19 of the inputs, NaN/inf included)
// `ok` is cleared and the caller recomputes with the reference's exact expression.
__device__ __forceinline__ int FastCellIndex(float c, float inv_resolution, bool* ok) {
  const float q0 = c * inv_resolution;
  const float n = rintf(q0);
  const float margin = 0.5f - fabsf(q0 - n);          // exact
  *ok = *ok && (margin > fabsf(q0) * 0x1p-20f);
  return static_cast<int>(n);
}

// One wavefront = 64 translations of one rotation.  Each lane rotates one point of the
// next 64-point chunk into LDS (the rotation is wave-uniform, so this removes the 64x
// redundant rotation), then all lanes walk the chunk in point order: broadcast LDS read,
// add the lane's translation, cell index, one gather from the padded brick, and the
// reference's sequential f32 accumulation.
__global__ void __launch_bounds__(64)
Rt3DScoreKernel(Rt3DParams P, const float* __restrict__ xyz, int n,
                float* __restrict__ unweighted, float* __restrict__ weighted,
                unsigned* __restrict__ max_bits) {
  __shared__ float4 rotated[64];
  const int lane = threadIdx.x;
  const int t = blockIdx.x * 64 + lane;
  const int r = blockIdx.y;
  const bool valid = t < P.num_translations;
  const float4 q4 = P.rotation[r];
  const Quat q{q4.w, q4.x, q4.y, q4.z};
  const float4 tr = P.translation[valid ? t : 0];
  const float res = P.resolution, inv = P.inv_resolution;
  const int sx = P.grid.nx + 2, sy = P.grid.ny + 2;
  const int ox = 1 - P.grid.lo_x, oy = 1 - P.grid.lo_y, oz = 1 - P.grid.lo_z;
  const int mx = P.grid.nx + 1, my = P.grid.ny + 1, mz = P.grid.nz + 1;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(P.grid.cells), 0, sx * sy * (P.grid.nz + 2) * 4, 0x00020000);

  const auto cell_offset = [&](const float4& rp) -> int {
    const float cx = rp.x + tr.x, cy = rp.y + tr.y, cz = rp.z + tr.z;   // rigid * point
    bool ok = true;
    int ix = FastCellIndex(cx, inv, &ok);
    int iy = FastCellIndex(cy, inv, &ok);
    int iz = FastCellIndex(cz, inv, &ok);
    if (!ok) {
      const int3 idx = CellIndex3(F3{cx, cy, cz}, res);
      ix = idx.x; iy = idx.y; iz = idx.z;
    }
    ix = min(max(ix + ox, 0), mx);
    iy = min(max(iy + oy, 0), my);
    iz = min(max(iz + oz, 0), mz);
    return ((iz * sy + iy) * sx + ix) * 4;
  };

  float acc = 0.f;
  constexpr int kBatch = 8;
  for (int base = 0; base < n; base += 64) {
    const int cnt = min(64, n - base);
    __syncthreads();                       // previous chunk fully consumed
    if (lane < cnt) {
      const int i = base + lane;
      const F3 rp = Rotate(q, F3{xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]});
      rotated[lane] = make_float4(rp.x, rp.y, rp.z, 0.f);
    }
    __syncthreads();
    int j = 0;
    for (; j + kBatch <= cnt; j += kBatch) {
      float v[kBatch];
#pragma unroll
      for (int k = 0; k < kBatch; ++k)
        v[k] = __uint_as_float(
            __builtin_amdgcn_raw_buffer_load_b32(rsrc, cell_offset(rotated[j + k]), 0, 0));
#pragma unroll
      for (int k = 0; k < kBatch; ++k) acc += v[k];
    }
    for (; j < cnt; ++j)
      acc += __uint_as_float(
          __builtin_amdgcn_raw_buffer_load_b32(rsrc, cell_offset(rotated[j]), 0, 0));
  }

  float w = 0.f;
  if (valid) {
    acc /= static_cast<float>(n);
    // Candidate order of the reference: z, y, x, rz, ry, rx nesting.
    const size_t c = static_cast<size_t>(t) * P.num_rotations + r;
    unweighted[c] = acc;
    const double penalty = static_cast<double>(tr.w) * P.wt +
                           static_cast<double>(P.rotation_angle[r]) * P.wr;
    w = static_cast<float>(static_cast<double>(acc) * exp(-(penalty * penalty)));
    weighted[c] = w;
  }
  unsigned bits = __float_as_uint(w);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) bits = max(bits, __shfl_xor(bits, off, 64));
  if (threadIdx.x == 0) atomicMax(max_bits, bits);
}

__global__ void Rt3DCollectKernel(const float* __restrict__ weighted, long long num_candidates,
                                  const unsigned* __restrict__ max_bits, int* __restrict__ count,
                                  long long* __restrict__ finalists, int capacity) {
  const long long c = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (c >= num_candidates) return;
  const float threshold = __uint_as_float(*max_bits) * (1.f - 1e-5f);
  if (weighted[c] >= threshold) {
    const int slot = atomicAdd(count, 1);
    if (slot < capacity) finalists[slot] = c;
  }
}


// ---------------------------------------------------------------------------
// Integer bounds (groups of translations, then single candidates) + exact finalists
// ---------------------------------------------------------------------------
// The reference's score of a candidate is mean_p P(cell(T_c p)) summed in f32 in point order
// (:97-114), and P is affine in the stored value: P = kMin + u * kScale with
// u = max(value, 1) - 1 (0 for unknown / outside; real arithmetic -- the f32 table rounds each
// entry by < 1.2e-7).  Integer sums of u are exact and order-free, so most of the search
// needs neither the f32 chain nor the IEEE divisions of GetCellIndex, and most candidates
// need not be scored at all:
//   * the grid becomes a padded uint8 brick q = u >> 7 with a zero halo as wide as twice the
//     reach of the translation window; rotated points are clamped (while they are staged) to
//     one reach outside the stored box, so no lookup needs a bounds test;
//   * GROUP pass: the (2L+1)^3 translations are tiled by 2 x 2 x 2 blocks of lattice steps.
//     The members of a block lie within 0.87 cells (per axis) of its centre, so the cell any
//     member reads is the centre's cell or one of its 26 neighbours: one lookup of the centre
//     in the 3 x 3 x 3-dilated brick bounds all eight members from above (the branch-and-bound
//     idea of the fast matcher, one level deep).  1/6 of the lookups, no exactness needed;
//   * CANDIDATE pass, only for the members of groups whose weighted upper bound reaches the
//     best weighted lower bound known: the cell index is rint(fma(c, RN(1/res), pad - lo))
//     with c = rp + tr exactly the reference's f32 coordinate; it equals the reference's
//     lround(c / res) unless the quotient lies within `1/2 - guard` of a half-integer.  Such
//     lookups are not repaired but COUNTED (per four points): each moves the sum by at most
//     255, which widens the candidate's score interval instead of costing a slow path.
//     Per candidate this yields Q and A with sum(u) in [128 (Q - 1020 A), 128 (Q + 1020 A) +
//     127 N], i.e. an interval for the f32 score once the rounding of the N-term f32 chain
//     ((N + 8) 2^-24, relative) is added;
//   * every candidate whose weighted upper bound reaches the best weighted lower bound is a
//     finalist; Rt3DExactKernel recomputes those with the reference's own arithmetic, and
//     the host applies the libm weight and the first-maximum rule to them exactly as before.
// Returned score and pose are bit-identical to the one-thread-per-candidate kernel above,
// which remains the path for flat score landscapes (more finalists than the list holds) and
// for windows / grids beyond the limits checked in cmx_rt3d_match.
constexpr int kBulk3DThreads = 256;
constexpr int kCand3DThreads = 128;         // candidate pass: work lists are short (a block's idle
                                            // wavefronts only hold wave slots)
constexpr int kBulk3DChunk = 256;          // points staged per round (one per thread)
constexpr int kAmbiguousQuad = 4 * 255;    // what one flagged group of four lookups can move Q by

typedef float v2f __attribute__((ext_vector_type(2)));

struct Rt3DBulkParams {
  const uint8_t* cells;          // [(nz+2p)][(ny+2p)][(nx+2p)] q = u >> 7 (or its dilation)
  unsigned cell_count;
  float pitch_x, pitch_y;        // nx + 2p, ny + 2p as floats (index arithmetic stays < 2^24)
  int tiles_y, pitch_z;          // candidate brick: columns of 8 x 4 cells, see TiledOffset
  float off_x, off_y, off_z;     // pad - lo
  float inv_resolution;
  float guard;                   // a lookup is unambiguous when max |q - rint(q)| <= guard
  float t0x, t0y, t0z;           // init.t: rp + init.t is what gets clamped
  float lo_x, hi_x, lo_y, hi_y, lo_z, hi_z;   // clamp range in metres
  int num_translations;          // entries of `translation` (T, or the number of groups)
  int num_rotations, n;
  const float4* rotation;        // as Rt3DParams
  const float4* translation;     // xyz + distance entering the weight (group: its smallest)
  const float* rotation_angle;
  // Group pass: every translation of rotation blockIdx.y.  Candidate pass: block blockIdx.x
  // takes work descriptor (r, chunk) = blocks[blockIdx.x]: items[r][256 chunk ..] of the
  // counts[r] listed translations.
  const int* counts;
  const int* items;              // [R][num_translations]
  const int2* blocks;
  double wt, wr;
  double min_probability;        // the f32 constants of the value table, widened
  double scale_over_n;           // 128 kScale / N
  double slack_hi;               // 127 kScale + 2e-7 (quantisation + rounding of the table)
  double delta;                  // relative slack of the f32 chain + exp
  float* upper;                  // [R][num_translations] weighted upper bound
  int block_items;               // candidate pass: items per work descriptor (= blockDim.x)
  int list_rotations;            // candidate pass: consecutive rotations sharing one work list;
                                 // an entry is w * num_translations + t (w: rotation in the list)
  uint2* sums;                   // candidate pass: [R][T] (Q, A) accumulated over point slices
  int slice_points;              // points per blockIdx.z (a multiple of the chunk)
  unsigned* max_lower_bits;      // atomicMax of the weighted lower bounds (candidate pass)
  unsigned* max_upper_bits;      // atomicMax of the weighted upper bounds (group pass)
};

__device__ __forceinline__ float ClampStage(float rp, float t0, float lo, float hi) {
  const float b = rp + t0;
  if (b < lo) return lo - t0;
  if (b > hi) return hi - t0;
  return rp;
}

// The candidate pass reads few, scattered translations per rotation: its brick is stored as
// columns of 8 (x) by 4 (y) cells running along z, so that a 128-byte line holds an 8 x 4 x 4
// block of cells -- the eight members of a group of translations (3 x 3 x 3 cells at most)
// touch one or two lines instead of up to nine rows.
//   offset = ((x >> 3) * tiles_y + (y >> 2)) * pitch_z * 32 + z * 32 + (y & 3) * 8 + (x & 7)
// (pitch_z is a multiple of 4, so lines never straddle columns.)
__host__ __device__ __forceinline__ unsigned TiledOffset(unsigned x, unsigned y, unsigned z,
                                                         unsigned tiles_y, unsigned pitch_z) {
  const unsigned column = (x >> 3) * tiles_y + (y >> 2);
  return ((column * pitch_z + z) << 5) | ((y & 3u) << 3) | (x & 7u);
}

// Weighted bounds of the f32 score from the integer sum `acc` and `ambiguous` flagged groups
// of four lookups; rounded outwards: a float strictly below / above the f64 bound.
__device__ __forceinline__ void Bounds3D(const Rt3DBulkParams& P, unsigned acc,
                                         unsigned ambiguous, float distance, int r, float* lower,
                                         float* upper) {
  const double spread = static_cast<double>(ambiguous) * kAmbiguousQuad;
  const double q_lo = fmax(static_cast<double>(acc) - spread, 0.);
  const double q_hi = static_cast<double>(acc) + spread;
  const double mean_lo = P.min_probability + P.scale_over_n * q_lo - 2e-7;
  const double mean_hi = P.min_probability + P.scale_over_n * q_hi + P.slack_hi;
  const double penalty = static_cast<double>(distance) * P.wt +
                         static_cast<double>(P.rotation_angle[r]) * P.wr;
  const double w = exp(-(penalty * penalty));
  *lower = static_cast<float>(mean_lo * w * (1. - P.delta)) * (1.f - 0x1p-22f);
  *upper = static_cast<float>(mean_hi * w * (1. + P.delta)) * (1.f + 0x1p-22f);
}

// Group pass: grid (ceil(R G / 256)); candidate pass: grid (work descriptors, 1, point
// slices).  256 threads: wave w of a block owns work items 256 chunk + 64 w .. + 63.  kGroups: group pass (no ambiguity bookkeeping, upper
// bounds only, all points in one block).  Candidate pass: the work lists are short, so the
// points are split over blockIdx.z as well and (Q, A) are accumulated with atomics (integer
// sums are order-free); Rt3DBoundsKernel turns them into bounds.
template <bool kGroups>
__global__ void __launch_bounds__(kBulk3DThreads)
Rt3DBulkKernel(Rt3DBulkParams P, const float* __restrict__ xyz) {
  // Points 2p, 2p + 1 as {x0, x1, y0, y1, z0, z1}: what the packed f32 instructions read.
  // Group pass: a block's 256 lanes are a window of the flat (rotation, group) sequence, i.e.
  // they belong to one or two consecutive rotations (G groups fill 3.4 wavefronts: a block per
  // rotation left 16 % of the lanes idle); both rotations' points are staged.
  __shared__ v2f stage[2][kGroups ? 2 : 1][3 * kBulk3DChunk / 2];
  const int tid = threadIdx.x;
  int r, t, rotation_a = 0;
  bool valid, wave_active;
  if (kGroups && gridDim.y > 1) {
    // (few groups per rotation, G < 128: a block would span more than two rotations; one
    // rotation per blockIdx.y then, as the candidate pass)
    r = blockIdx.y;
    if (r >= P.num_rotations) return;                // (grid.y is at least 2 to mark this mode)
    t = min(static_cast<int>(blockIdx.x * kBulk3DThreads + tid), P.num_translations - 1);
    valid = static_cast<int>(blockIdx.x * kBulk3DThreads + tid) < P.num_translations;
    wave_active = static_cast<int>(blockIdx.x * kBulk3DThreads + (tid & ~63)) < P.num_translations;
    rotation_a = r;
  } else if (kGroups) {
    // blockDim.x <= G + 1 lanes: they belong to at most two consecutive rotations.
    const long long first_flat = static_cast<long long>(blockIdx.x) * blockDim.x;
    const long long flat = first_flat + tid;
    const long long total = static_cast<long long>(P.num_rotations) * P.num_translations;
    rotation_a = static_cast<int>(first_flat / P.num_translations);
    valid = flat < total;
    const long long f = valid ? flat : total - 1;
    r = static_cast<int>(f / P.num_translations);
    t = static_cast<int>(f - static_cast<long long>(r) * P.num_translations);
    wave_active = first_flat + (tid & ~63) < total;
  } else {
    const int2 work = P.blocks[blockIdx.x];
    r = work.x;                                      // (this kernel: lists of one rotation)
    const int slot = work.y * P.block_items + tid;
    const int count = P.counts[r];
    if (work.y * P.block_items >= count) return;                         // whole block idle
    wave_active = work.y * P.block_items + (tid & ~63) < count;
    valid = slot < count;
    t = P.items[static_cast<size_t>(r) * P.num_translations + (valid ? slot : count - 1)];
    rotation_a = r;
  }
  const int which = r - rotation_a;                  // 0 or 1: which staged rotation a lane reads
  const int rotation_b = min(rotation_a + 1, P.num_rotations - 1);
  const float4 qa4 = P.rotation[rotation_a], qb4 = P.rotation[rotation_b];
  const Quat q{qa4.w, qa4.x, qa4.y, qa4.z}, q_b{qb4.w, qb4.x, qb4.y, qb4.z};
  const float4 tr = P.translation[t];
  const int first = kGroups ? 0 : blockIdx.z * P.slice_points;   // (blockIdx.z: point slice)
  const int n = kGroups ? P.n : min(P.n, first + P.slice_points);
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint8_t*>(P.cells), 0, P.cell_count, 0x00020000);

  const auto stage_chunk = [&](int base, int buf) {
    for (int k = tid; k < kBulk3DChunk; k += blockDim.x) {   // (the group pass may run 192 wide)
      const int i = base + k;
#pragma unroll
      for (int w = 0; w < (kGroups ? 2 : 1); ++w) {
        // Padding points sit one reach below the box: every translation reads the zero halo.
        float4 out = make_float4(P.lo_x - P.t0x, P.lo_y - P.t0y, P.lo_z - P.t0z, 0.f);
        if (i < n) {
          const F3 rp = Rotate(w == 0 ? q : q_b, F3{xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]});
          out.x = ClampStage(rp.x, P.t0x, P.lo_x, P.hi_x);
          out.y = ClampStage(rp.y, P.t0y, P.lo_y, P.hi_y);
          out.z = ClampStage(rp.z, P.t0z, P.lo_z, P.hi_z);
        }
        float* dst = reinterpret_cast<float*>(stage[buf][w]) + 6 * (k >> 1) + (k & 1);
        dst[0] = out.x; dst[2] = out.y; dst[4] = out.z;
      }
    }
  };

  const v2f trx = {tr.x, tr.x}, try_ = {tr.y, tr.y}, trz = {tr.z, tr.z};
  const v2f inv = {P.inv_resolution, P.inv_resolution};
  const v2f ofx = {P.off_x, P.off_x}, ofy = {P.off_y, P.off_y}, ofz = {P.off_z, P.off_z};
  const v2f px2 = {P.pitch_x, P.pitch_x}, py2 = {P.pitch_y, P.pitch_y};
  const float guard = P.guard;
  unsigned acc = 0, ambiguous = 0;

  stage_chunk(first, 0);
  __syncthreads();
  int buf = 0;
  unsigned p0 = 0, p1 = 0, p2 = 0, p3 = 0;     // the previous four lookups, still in flight
  for (int base = first; base < n; base += kBulk3DChunk, buf ^= 1) {
    if (base + kBulk3DChunk < n) stage_chunk(base + kBulk3DChunk, buf ^ 1);
    if (wave_active) {
      const v2f* __restrict__ s = stage[buf][kGroups ? which : 0];
#pragma unroll 2
      for (int j = 0; j < kBulk3DChunk; j += 4) {
        unsigned v[4];
        float g = 0.f;
#pragma unroll
        for (int k = 0; k < 4; k += 2) {
          const v2f* pr = s + 3 * ((j + k) >> 1);                // LDS broadcast reads
          const v2f cx = pr[0] + trx;                            // rigid * point, as the reference
          const v2f cy = pr[1] + try_;
          const v2f cz = pr[2] + trz;
          const v2f qx = __builtin_elementwise_fma(cx, inv, ofx);
          const v2f qy = __builtin_elementwise_fma(cy, inv, ofy);
          const v2f qz = __builtin_elementwise_fma(cz, inv, ofz);
          const v2f nx = {rintf(qx.x), rintf(qx.y)};
          const v2f ny = {rintf(qy.x), rintf(qy.y)};
          const v2f nz = {rintf(qz.x), rintf(qz.y)};
          if (!kGroups) {
            const v2f dx = qx - nx, dy = qy - ny, dz = qz - nz;
            g = fmaxf(fmaxf(g, fabsf(dx.x)), fabsf(dx.y));
            g = fmaxf(fmaxf(g, fabsf(dy.x)), fabsf(dy.y));
            g = fmaxf(fmaxf(g, fabsf(dz.x)), fabsf(dz.y));
          }
          if (kGroups) {
            // (z * pitch_y + y) * pitch_x + x: integers below 2^24, exact in f32.
            const v2f o =
                __builtin_elementwise_fma(__builtin_elementwise_fma(nz, py2, ny), px2, nx);
            v[k] = __builtin_amdgcn_raw_buffer_load_b8(rsrc, static_cast<unsigned>(o.x), 0, 0);
            v[k + 1] = __builtin_amdgcn_raw_buffer_load_b8(rsrc, static_cast<unsigned>(o.y), 0, 0);
          } else {
            const unsigned o0 = TiledOffset(static_cast<unsigned>(nx.x), static_cast<unsigned>(ny.x),
                                            static_cast<unsigned>(nz.x), P.tiles_y, P.pitch_z);
            const unsigned o1 = TiledOffset(static_cast<unsigned>(nx.y), static_cast<unsigned>(ny.y),
                                            static_cast<unsigned>(nz.y), P.tiles_y, P.pitch_z);
            v[k] = __builtin_amdgcn_raw_buffer_load_b8(rsrc, o0, 0, 0);
            v[k + 1] = __builtin_amdgcn_raw_buffer_load_b8(rsrc, o1, 0, 0);
          }
        }
        if (!kGroups) ambiguous += g > guard ? 1u : 0u;
        acc += (p0 + p1) + (p2 + p3);            // consumed one iteration after they were issued
        p0 = v[0]; p1 = v[1]; p2 = v[2]; p3 = v[3];
      }
    }
    __syncthreads();
  }
  acc += (p0 + p1) + (p2 + p3);

  const size_t c = static_cast<size_t>(r) * P.num_translations + t;
  if (!kGroups) {
    if (valid) {
      atomicAdd(&P.sums[c].x, acc);
      if (ambiguous) atomicAdd(&P.sums[c].y, ambiguous);
    }
    return;
  }
  float lower = 0.f, upper = 0.f;
  if (valid) {
    Bounds3D(P, acc, 0u, tr.w, r, &lower, &upper);
    P.upper[c] = upper;
  }
  // Bounds are positive: bit order == value order.
  unsigned bits = __float_as_uint(upper);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) bits = max(bits, __shfl_xor(bits, off, 64));
  if ((tid & 63) == 0 && bits != 0) atomicMax(P.max_upper_bits, bits);
}

// grid (work descriptors): bounds of the candidates on the work lists from their (Q, A).
__global__ void __launch_bounds__(1024)
Rt3DBoundsKernel(Rt3DBulkParams P) {
  const int2 work = P.blocks[blockIdx.x];
  const int list = work.x;
  const int slot = work.y * P.block_items + threadIdx.x;
  const int count = P.counts[list];
  float lower = 0.f, upper = 0.f;
  if (slot < count) {
    const int entry =
        P.items[static_cast<size_t>(list) * P.list_rotations * P.num_translations + slot];
    const int w = entry / P.num_translations, t = entry - w * P.num_translations;
    const int r = list * P.list_rotations + w;
    const size_t c = static_cast<size_t>(r) * P.num_translations + t;
    const uint2 qa = P.sums[c];
    Bounds3D(P, qa.x, qa.y, P.translation[t].w, r, &lower, &upper);
    P.upper[c] = upper;
  }
  unsigned bits = __float_as_uint(lower);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) bits = max(bits, __shfl_xor(bits, off, 64));
  if ((threadIdx.x & 63) == 0 && bits != 0) atomicMax(P.max_lower_bits, bits);
}

// ---- Staged candidate pass --------------------------------------------------------------
// The second candidate round scores every member of every group the best lower bound cannot
// exclude -- a tenth of C4's 1.77 M candidates, each over all 65 536 points, although all but a
// handful lose by a wide margin: the group bound is loose (a 3 x 3 x 3 dilation), the candidates
// themselves are not close.  But the group bound is a sum over POINTS of values that dominate
// the member's own value point by point, so for any part S of the cloud
//     score sum of c  <=  (sum of c over S)  +  (group sum over the rest),
// and the right-hand side tightens as S grows.  The cloud is cut into three segments (a
// quarter, a quarter, a half: Rt3DBinScanKernel), the group pass reports a group's sum by
// segment, and the round runs segment by segment: after the first and after the second,
// candidates whose weighted bound has fallen below the best lower bound are dropped from the
// work lists (they keep `upper` = 0: never finalists, exactly what their full evaluation would
// have concluded).  C4 (tools/prototype numbers in profiles/HISTORY.md 5.4): of 154 k candidates 48 % are
// alive after a quarter of the points, 10 % after half.
//
// grid (work descriptors): flags[r][t] = 1 for the listed candidates that stay.  `stage` = the
// segment just finished (0 or 1).  Verification mode (`stage_ub` != null: CMX_RT3D_VERIFY): the
// lists stay as they are, every candidate is evaluated in full, and this kernel only records the
// decision (`dropped`) and the smallest bound seen (in units of q) for Rt3DStageCheckKernel.
__global__ void __launch_bounds__(1024)
Rt3DStageFilterKernel(Rt3DBulkParams P, const uint2* __restrict__ group_total,
                      const uint2* __restrict__ group_seg, int num_groups, int side,
                      int groups_per_axis, int stage, uint8_t* __restrict__ flags,
                      unsigned long long* __restrict__ stage_ub, uint8_t* __restrict__ dropped) {
  const int2 work = P.blocks[blockIdx.x];
  const int list = work.x;
  const int slot = work.y * P.block_items + threadIdx.x;
  if (slot >= P.counts[list]) return;
  const int entry =
      P.items[static_cast<size_t>(list) * P.list_rotations * P.num_translations + slot];
  const int w = entry / P.num_translations, t = entry - w * P.num_translations;
  const int r = list * P.list_rotations + w;
  const size_t c = static_cast<size_t>(r) * P.num_translations + t;
  const uint2 qa = P.sums[c];                       // over the segments done so far
  const int x = t % side, y = (t / side) % side, z = t / (side * side);
  const int g = ((z >> 1) * groups_per_axis + (y >> 1)) * groups_per_axis + (x >> 1);
  const size_t gi = static_cast<size_t>(r) * num_groups + g;
  const uint2 seg = group_seg[gi];
  const unsigned long long rest = static_cast<unsigned long long>(group_total[gi].x) - seg.x -
                                  (stage >= 1 ? seg.y : 0u);
  const unsigned long long q_hi = static_cast<unsigned long long>(qa.x) +
                                  static_cast<unsigned long long>(qa.y) * kAmbiguousQuad + rest;
  // Bounds3D's upper bound for an integer sum of q_hi (quantisation, table rounding and the f32
  // chain's slack included), rounded outwards
  const double mean_hi = P.min_probability + P.scale_over_n * static_cast<double>(q_hi) + P.slack_hi;
  const double penalty = static_cast<double>(P.translation[t].w) * P.wt +
                         static_cast<double>(P.rotation_angle[r]) * P.wr;
  const double weight = exp(-(penalty * penalty));
  const float upper = static_cast<float>(mean_hi * weight * (1. + P.delta)) * (1.f + 0x1p-22f);
  const bool stays = upper >= __uint_as_float(*P.max_lower_bits);
  if (stage_ub != nullptr) {
    stage_ub[c] = stage == 0 ? q_hi : min(stage_ub[c], q_hi);
    if (stage == 0) dropped[c] = stays ? 0 : 1;
    else if (!stays) dropped[c] = 1;
    return;
  }
  if (stays) flags[c] = 1;
}

// Verification mode of the staged pass, after the round's bounds: every bound recorded on the way
// must dominate the candidate's own full sum (at its lower end), and a candidate the shipped
// path would have dropped gets the `upper` = 0 it would have kept there.  grid (work descriptors).
__global__ void __launch_bounds__(1024)
Rt3DStageCheckKernel(Rt3DBulkParams P, const unsigned long long* __restrict__ stage_ub,
                     const uint8_t* __restrict__ dropped, int* __restrict__ violations) {
  const int2 work = P.blocks[blockIdx.x];
  const int list = work.x;
  const int slot = work.y * P.block_items + threadIdx.x;
  if (slot >= P.counts[list]) return;
  const int entry =
      P.items[static_cast<size_t>(list) * P.list_rotations * P.num_translations + slot];
  const int w = entry / P.num_translations, t = entry - w * P.num_translations;
  const int r = list * P.list_rotations + w;
  const size_t c = static_cast<size_t>(r) * P.num_translations + t;
  const uint2 qa = P.sums[c];                       // the full sum now
  const unsigned long long spread = static_cast<unsigned long long>(qa.y) * kAmbiguousQuad;
  const unsigned long long q_lo = qa.x > spread ? qa.x - spread : 0ull;
  if (stage_ub[c] < q_lo) atomicAdd(violations, 1);
  if (dropped[c]) {
    if (P.upper[c] >= __uint_as_float(*P.max_lower_bits)) atomicAdd(violations, 1);   // a finalist!
    P.upper[c] = 0.f;
  }
}

// CMX_RT3D_VERIFY=1 (tests): a group's upper bound must not lie below the LOWER bound of any of
// its members that was scored -- both bracket the same true score.  (A group pass reading the
// wrong staged rotation once produced garbage bounds that every parity test survived: the
// optimum happened not to be pruned.)  grid (work descriptors).
__global__ void __launch_bounds__(1024)
Rt3DVerifyKernel(Rt3DBulkParams P, const float* __restrict__ group_upper, int num_groups,
                 int side, int groups_per_axis, int* __restrict__ violations) {
  const int2 work = P.blocks[blockIdx.x];
  const int list = work.x;
  const int slot = work.y * P.block_items + threadIdx.x;
  if (slot >= P.counts[list]) return;
  const int entry =
      P.items[static_cast<size_t>(list) * P.list_rotations * P.num_translations + slot];
  const int w = entry / P.num_translations, t = entry - w * P.num_translations;
  const int r = list * P.list_rotations + w;
  const uint2 qa = P.sums[static_cast<size_t>(r) * P.num_translations + t];
  float lower, upper;
  Bounds3D(P, qa.x, qa.y, P.translation[t].w, r, &lower, &upper);
  const int x = t % side, y = (t / side) % side, z = t / (side * side);
  const int g = ((z >> 1) * groups_per_axis + (y >> 1)) * groups_per_axis + (x >> 1);
  if (group_upper[static_cast<size_t>(r) * num_groups + g] < lower) atomicAdd(violations, 1);
}


// ---------------------------------------------------------------------------
// LDS-tiled bulk passes
// ---------------------------------------------------------------------------
// The gathers of the passes above go to memory one byte per lane: a wave's 64 lookups of one
// point touch ~30 cache lines, and the L1's tag rate (~1 line per cycle and CU) is what bounds
// them (1.3e12 lookups/s, profiles/r03_gather_ceiling.txt).  But all lanes of a workgroup read
// the SAME point at the same time, displaced only by their translations (a window of a dozen
// cells) -- so for a spatially compact CHUNK of points everything a workgroup reads lies in a
// box of a few ten cells per axis.  The cloud is therefore sorted into bins of kTileBin^3 cells
// (counting sort on the device, integer sums do not care about the order) and cut into chunks
// of at most kTileChunk points; per chunk a workgroup
//   1. rotates the chunk by its rotation(s) into LDS (as before) and takes the bounding box of
//      the rotated points in cell coordinates (wave reductions + LDS atomics),
//   2. widens the box by the span of the translation table, copies that box of the brick into
//      LDS with aligned dword loads (row-major brick, pitch a multiple of 4),
//   3. runs the same lookup loop against the tile: ds_read_u8 instead of buffer_load_ubyte.
// A box that does not fit the tile capacity (a chunk of far points under a wide rotation
// block) takes the memory gathers for that chunk -- same arithmetic, same sums.  The integer
// sums are accumulated over chunk slices (blockIdx.y) with atomics; bounds come from
// Rt3DSumBoundsKernel / Rt3DBoundsKernel.  Every sum is identical to what Rt3DBulkKernel
// computes (CMX_RT3D_TILES=0 runs that kernel; tests compare the two).
constexpr int kTileChunkGroups = 512;      // points per chunk, group pass (fewer, fuller work units)
constexpr int kTileChunkCandidates = 256;  // ... candidate pass (its f32 stage is 12 B per point and rotation)
constexpr int kTileBin = 24;
constexpr int kTileMaxRotations = 8;

struct Rt3DTileParams {
  const float* sorted_xyz;       // bin-sorted cloud
  const int2* chunks;            // (first point, length)
  const int* num_chunks;
  int pitch_x, pitch_y, pitch_z; // brick dims (pitch_x a multiple of 4)
  float tr_lo[3], tr_hi[3];      // span of translation * inv_resolution over the table
  int rotations_per_block;       // group pass: rotations of a workgroup
  int tile_capacity;             // bytes of dynamic LDS behind the staged points
  int block_items;               // candidate pass: items per work descriptor (= blockDim.x)
  int fixed_point;               // group pass: packed fixed-point cell arithmetic (see kernel)
  const float* boxes;            // Rt3DChunkBoxKernel: [rotation block][chunk][6], or null (boxes
                                 // are then reduced inside the tile kernel)
  // Point segments (Rt3DBinScanKernel): seg_bounds[0], [1] = first chunk of segments 1 and 2 of
  // this pass's list, or null (one segment: every chunk).  The pass takes the chunks of segments
  // seg_first .. seg_last.  Group pass: seg_sums[(r, group)] = (sum over segment 0, sum over
  // segment 1) next to the total in P.sums, or null.
  const int* seg_bounds;
  int seg_first, seg_last;
  uint2* seg_sums;
  unsigned long long* stats;     // CMX_RT3D_REPORT (its atomics cost ~0.5 ms per pass: not for timing): [0] chunks in LDS, [1] on the gather path,
                                 // [2] tile bytes, [3] points (per workgroup and chunk); or null
};

// Bin of a point: its cell at the central candidate (rotation index R/2, translation T/2).
struct Rt3DBinParams {
  float4 rotation, translation;
  float inv_resolution, off_x, off_y, off_z;
  float t0x, t0y, t0z, lo_x, hi_x, lo_y, hi_y, lo_z, hi_z;     // ClampStage
  int bins_x, bins_y, bins_z;
  int n;
};

__device__ __forceinline__ int Rt3DBinOf(const Rt3DBinParams& P, const float* __restrict__ xyz,
                                         int i) {
  const Quat q{P.rotation.w, P.rotation.x, P.rotation.y, P.rotation.z};
  const F3 rp = Rotate(q, F3{xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]});
  const float x = ClampStage(rp.x, P.t0x, P.lo_x, P.hi_x) + P.translation.x;
  const float y = ClampStage(rp.y, P.t0y, P.lo_y, P.hi_y) + P.translation.y;
  const float z = ClampStage(rp.z, P.t0z, P.lo_z, P.hi_z) + P.translation.z;
  // (NaN coordinates never get here: the bulk path requires finite points)
  const int cx = static_cast<int>(rintf(fmaf(x, P.inv_resolution, P.off_x)));
  const int cy = static_cast<int>(rintf(fmaf(y, P.inv_resolution, P.off_y)));
  const int cz = static_cast<int>(rintf(fmaf(z, P.inv_resolution, P.off_z)));
  const int bx = min(max(cx / kTileBin, 0), P.bins_x - 1);
  const int by = min(max(cy / kTileBin, 0), P.bins_y - 1);
  const int bz = min(max(cz / kTileBin, 0), P.bins_z - 1);
  return (bz * P.bins_y + by) * P.bins_x + bx;
}

// counters[bin] += (lanes of this wave with that bin); returns the lane's own slot.  Scan order
// puts neighbouring points into the same bin, so a wave needs one or two atomics.
__device__ __forceinline__ int WaveAggregatedAdd(int* __restrict__ counters, int bin, bool active) {
  const int lane = threadIdx.x & 63;
  int slot = 0;
  for (;;) {
    const unsigned long long todo = __ballot(active);
    if (todo == 0ull) break;
    const int leader = __ffsll(static_cast<long long>(todo)) - 1;
    const int leader_bin = __shfl(bin, leader, 64);
    const bool mine = active && bin == leader_bin;
    const unsigned long long group = __ballot(mine);
    int base = 0;
    if (lane == leader) base = atomicAdd(&counters[leader_bin], __popcll(group));
    base = __shfl(base, leader, 64);
    if (mine) {
      slot = base + __popcll(group & ((1ull << lane) - 1ull));
      active = false;
    }
  }
  return slot;
}

__global__ void Rt3DBinCountKernel(Rt3DBinParams P, const float* __restrict__ xyz,
                                   int* __restrict__ bin_count) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = i < P.n;
  const int bin = active ? Rt3DBinOf(P, xyz, i) : 0;
  (void)WaveAggregatedAdd(bin_count, bin, active);
}

// Point segments of the staged candidate pass (see "Staged candidate pass" below): the cloud is
// cut into three parts -- about a quarter, a quarter and a half of the points -- that BOTH chunk
// lists respect, so that the group pass can report a group's sum over each part and the
// candidate pass can stop after the first or the second.  The unit is a group-pass piece (at
// most kTileChunkGroups points of one bin; the candidate pass's pieces are its halves): a
// piece belongs to the segment its midpoint falls into on the axis of bin-sorted point
// indices, which is cut into windows of `window` points -- first quarter of a window: segment
// 0, second quarter: 1, rest: 2.  Any assignment is correct; this one keeps the parts close to
// their nominal sizes whatever the bins hold and interleaves them through the volume.
__device__ __forceinline__ int Rt3DSegmentOf(int first_point, int length, int window, int end0,
                                             int end1) {
  const int at = (first_point + (length >> 1)) & (window - 1);       // window: a power of two
  return at < end0 ? 0 : at < end1 ? 1 : 2;          // (end0, end1: window / 4, window / 2)
}

// One workgroup: exclusive scan of the bin counts -> first point of every bin (written over the
// counts: the scatter kernel's cursors) and the two chunk lists (pieces of at most
// kTileChunkGroups / kTileChunkCandidates points of one bin), each ordered by segment:
// num_chunks[0 / 1] = pieces of the group / candidate list, [2], [3] = where segments 1 and 2
// of the group list begin, [4], [5] = the same for the candidate list.
__global__ void __launch_bounds__(1024)
Rt3DBinScanKernel(int* __restrict__ bin_count, int num_bins, int2* __restrict__ chunks_groups,
                  int2* __restrict__ chunks_candidates, int* __restrict__ num_chunks,
                  int window, int end0, int end1) {
  static_assert(kTileChunkGroups == 2 * kTileChunkCandidates, "candidate pieces are halves");
  __shared__ int part_points[1024], part_a[3][1024], part_b[3][1024];
  __shared__ int seg_base_a[3], seg_base_b[3];
  const int tid = threadIdx.x;
  const int per = (num_bins + 1023) / 1024;
  const int begin = min(tid * per, num_bins), end = min(begin + per, num_bins);
  // pass 1: points per thread (the segment of a piece depends on its first point)
  int points = 0;
  for (int b = begin; b < end; ++b) points += bin_count[b];
  part_points[tid] = points;
  __syncthreads();
  if (tid == 0) {
    int p = 0;
    for (int k = 0; k < 1024; ++k) {
      const int pp = part_points[k];
      part_points[k] = p;
      p += pp;
    }
  }
  __syncthreads();
  // pass 2: pieces per segment
  int na[3] = {0, 0, 0}, nb[3] = {0, 0, 0};
  {
    int p = part_points[tid];
    for (int b = begin; b < end; ++b) {
      const int count = bin_count[b];
      for (int first = 0; first < count; first += kTileChunkGroups) {
        const int len = min(kTileChunkGroups, count - first);
        const int seg = Rt3DSegmentOf(p + first, len, window, end0, end1);
        const int halves = (len + kTileChunkCandidates - 1) / kTileChunkCandidates;
        if (seg == 0) { na[0] += 1; nb[0] += halves; }
        else if (seg == 1) { na[1] += 1; nb[1] += halves; }
        else { na[2] += 1; nb[2] += halves; }
      }
      p += count;
    }
  }
#pragma unroll
  for (int g = 0; g < 3; ++g) { part_a[g][tid] = na[g]; part_b[g][tid] = nb[g]; }
  __syncthreads();
  if (tid < 6) {                           // six independent serial scans
    int* part = tid < 3 ? part_a[tid] : part_b[tid - 3];
    int run = 0;
    for (int k = 0; k < 1024; ++k) {
      const int v = part[k];
      part[k] = run;
      run += v;
    }
    (tid < 3 ? seg_base_a : seg_base_b)[tid % 3] = run;       // totals, turned into bases below
  }
  __syncthreads();
  if (tid == 0) {
    const int a0 = seg_base_a[0], a1 = seg_base_a[1], a2 = seg_base_a[2];
    const int b0 = seg_base_b[0], b1 = seg_base_b[1], b2 = seg_base_b[2];
    seg_base_a[0] = 0; seg_base_a[1] = a0; seg_base_a[2] = a0 + a1;
    seg_base_b[0] = 0; seg_base_b[1] = b0; seg_base_b[2] = b0 + b1;
    num_chunks[0] = a0 + a1 + a2;
    num_chunks[1] = b0 + b1 + b2;
    num_chunks[2] = a0; num_chunks[3] = a0 + a1;
    num_chunks[4] = b0; num_chunks[5] = b0 + b1;
  }
  __syncthreads();
  // pass 3: the lists
  int ca[3], cb[3];
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    ca[g] = seg_base_a[g] + part_a[g][tid];
    cb[g] = seg_base_b[g] + part_b[g][tid];
  }
  int p = part_points[tid];
  for (int b = begin; b < end; ++b) {
    const int count = bin_count[b];
    bin_count[b] = p;
    for (int first = 0; first < count; first += kTileChunkGroups) {
      const int len = min(kTileChunkGroups, count - first);
      const int seg = Rt3DSegmentOf(p + first, len, window, end0, end1);
      // (static selects: no dynamically indexed private array)
      int at_a = seg == 0 ? ca[0] : seg == 1 ? ca[1] : ca[2];
      int at_b = seg == 0 ? cb[0] : seg == 1 ? cb[1] : cb[2];
      chunks_groups[at_a++] = make_int2(p + first, len);
      for (int half = 0; half < len; half += kTileChunkCandidates)
        chunks_candidates[at_b++] =
            make_int2(p + first + half, min(kTileChunkCandidates, len - half));
      if (seg == 0) { ca[0] = at_a; cb[0] = at_b; }
      else if (seg == 1) { ca[1] = at_a; cb[1] = at_b; }
      else { ca[2] = at_a; cb[2] = at_b; }
    }
    p += count;
  }
}

__global__ void Rt3DBinScatterKernel(Rt3DBinParams P, const float* __restrict__ xyz,
                                     int* __restrict__ cursor, float* __restrict__ sorted) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = i < P.n;
  const int bin = active ? Rt3DBinOf(P, xyz, i) : 0;
  const int slot = WaveAggregatedAdd(cursor, bin, active);
  if (active) {
    sorted[3 * slot] = xyz[3 * i];
    sorted[3 * slot + 1] = xyz[3 * i + 1];
    sorted[3 * slot + 2] = xyz[3 * i + 2];
  }
}

// Bounding boxes of the rotated chunks, once per match: box[(block, chunk)] = min x, y, z, max
// x, y, z over the chunk's points rotated by every rotation of rotation block `block`
// (`rotations_per_block` consecutive rotations), clamped like the staged points, in cells
// (coordinate * inv_resolution; no offset, no translation).  grid (rotation blocks, chunk
// slices), 256 threads: ONE WAVE per chunk (no workgroup barrier, no LDS: a block per chunk
// with an LDS reduction cost 0.5 ms per match).
__global__ void __launch_bounds__(256)
Rt3DChunkBoxKernel(Rt3DBulkParams P, const float* __restrict__ sorted_xyz,
                   const int2* __restrict__ chunks, const int* __restrict__ num_chunks,
                   int rotations_per_block, float* __restrict__ boxes) {
  const int block = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rotation_a = block * rotations_per_block;
  const int num_rot = min(rotations_per_block, P.num_rotations - rotation_a);
  const int total = *num_chunks;
  for (int chunk = blockIdx.y * 4 + wave; chunk < total; chunk += gridDim.y * 4) {
    const int2 span = chunks[chunk];
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int k = lane; k < span.y; k += 64) {
      const int i = span.x + k;
      const F3 p{sorted_xyz[3 * i], sorted_xyz[3 * i + 1], sorted_xyz[3 * i + 2]};
      for (int w = 0; w < num_rot; ++w) {
        const float4 q4 = P.rotation[rotation_a + w];
        const F3 rp = Rotate(Quat{q4.w, q4.x, q4.y, q4.z}, p);
        const float c[3] = {ClampStage(rp.x, P.t0x, P.lo_x, P.hi_x) * P.inv_resolution,
                            ClampStage(rp.y, P.t0y, P.lo_y, P.hi_y) * P.inv_resolution,
                            ClampStage(rp.z, P.t0z, P.lo_z, P.hi_z) * P.inv_resolution};
#pragma unroll
        for (int a = 0; a < 3; ++a) { mn[a] = fminf(mn[a], c[a]); mx[a] = fmaxf(mx[a], c[a]); }
      }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) {
        mn[a] = fminf(mn[a], __shfl_xor(mn[a], off, 64));
        mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off, 64));
      }
    }
    if (lane < 6) {
      const float v = lane == 0 ? mn[0] : lane == 1 ? mn[1] : lane == 2 ? mn[2]
                      : lane == 3 ? mx[0] : lane == 4 ? mx[1] : mx[2];
      boxes[(static_cast<size_t>(block) * total + chunk) * 6 + lane] = v;
    }
  }
}

typedef unsigned uint4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void LdsRead128Asm(uint4v* out, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1" : "=v"(*out) : "v"(addr));
}
__device__ __forceinline__ void LdsReadU8Asm(unsigned* out, unsigned addr) {
  asm volatile("ds_read_u8 %0, %1" : "=v"(*out) : "v"(addr));
}

// kGroups: grid (ceil(R / rotations_per_block), chunk slices), blockDim = the (rotation, group)
// lanes of a block rounded up to whole waves.  Candidate pass: grid (work descriptors, chunk
// slices), blockDim = block_items.  Dynamic LDS: staged points | tile.
// kGroups: the arithmetic of the group pass (centre lookups in a dilated brick: fixed-point
// words, no ambiguity bookkeeping, sums by segment).  kLists: lanes are entries of work lists over
// P.list_rotations rotations (Rt3DCompactKernel) instead of all (rotation, translation) pairs of
// TP.rotations_per_block rotations.  <true, false>: the dense group pass (and the rotation-block
// pass above it); <true, true>: the group pass over the (rotation, group) pairs a rotation-block
// bound could not exclude; <false, true>: the candidate passes.
template <bool kGroups, bool kLists>
__global__ void __launch_bounds__(1024)
Rt3DTileKernel(Rt3DBulkParams P, Rt3DTileParams TP) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tile_smem[];
  static_assert(kGroups || kLists, "candidates always come from work lists");
  constexpr int kChunk = kLists ? kTileChunkCandidates : kTileChunkGroups;
  constexpr int kStageStride = 3 * kChunk / 2 + 2;     // v2f per staged rotation (+16 B: bank shift)
  const int tid = threadIdx.x;
  const int rotations = kLists ? P.list_rotations : TP.rotations_per_block;
  // Dynamic LDS: tile | staged points | (group pass) the points again as packed fixed-point
  // words, see step 2b.  The tile comes first: its LDS address is then a compile-time constant
  // that folds into the gathers' offset field.
  uint8_t* const tile = tile_smem;
  if (static_cast<unsigned>(reinterpret_cast<uintptr_t>(tile)) != 0u) __builtin_trap();
  v2f* const stage = reinterpret_cast<v2f*>(tile_smem + TP.tile_capacity);
  uint32_t* const packed = reinterpret_cast<uint32_t*>(stage + kStageStride * rotations);
  // min x, y, z, max x, y, z of the rotated chunk (cells).  (No static __shared__ in this kernel:
  // the tile then starts at LDS address 0 and the gathers need no base.)
  int* const box = reinterpret_cast<int*>(packed + (kGroups ? kChunk * rotations : 0));

  int r, t, rotation_a, num_rot;
  bool valid;
  if (!kLists) {
    rotation_a = blockIdx.x * rotations;
    num_rot = min(rotations, P.num_rotations - rotation_a);
    const int item = tid;
    valid = item < num_rot * P.num_translations;
    const int which = valid ? item / P.num_translations : 0;
    r = rotation_a + which;
    t = valid ? item - which * P.num_translations : 0;
  } else {
    const int2 work = P.blocks[blockIdx.x];
    const int list = work.x;
    rotation_a = list * P.list_rotations;
    num_rot = min(P.list_rotations, P.num_rotations - rotation_a);
    const int count = P.counts[list];
    const int slot = work.y * TP.block_items + tid;
    valid = slot < count;
    if (work.y * TP.block_items >= count) return;
    const int entry = P.items[static_cast<size_t>(list) * P.list_rotations * P.num_translations +
                              (valid ? slot : count - 1)];
    const int w = entry / P.num_translations;
    t = entry - w * P.num_translations;
    r = rotation_a + w;
  }
  const v2f* const my_stage = stage + (r - rotation_a) * kStageStride;
  const float4 tr = P.translation[t];
  const v2f trx = {tr.x, tr.x}, try_ = {tr.y, tr.y}, trz = {tr.z, tr.z};
  const v2f inv = {P.inv_resolution, P.inv_resolution};
  const v2f ofx = {P.off_x, P.off_x}, ofy = {P.off_y, P.off_y}, ofz = {P.off_z, P.off_z};
  const float guard = P.guard;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint8_t*>(P.cells), 0, P.cell_count, 0x00020000);
  unsigned acc = 0, ambiguous = 0;
  const int num_chunks = *TP.num_chunks;
  // the chunks of this pass: all of them, or the segments seg_first .. seg_last of the list
  int chunk_begin = 0, chunk_end = num_chunks, split0 = num_chunks, split1 = num_chunks;
  if (TP.seg_bounds != nullptr) {
    split0 = TP.seg_bounds[0];
    split1 = TP.seg_bounds[1];
    chunk_begin = TP.seg_first <= 0 ? 0 : TP.seg_first == 1 ? split0 : split1;
    chunk_end = TP.seg_last >= 2 ? num_chunks : TP.seg_last == 1 ? split1 : split0;
  }
  unsigned acc_seen = 0, acc_seg0 = 0, acc_seg1 = 0;     // group pass: the total by segment
  // Fixed-point group pass.  The centre of a group needs no exact cell: its lookup in the
  // 3 x 3 x 3-dilated brick covers the members as long as the cell it reads is the cell of a
  // point within 1 - 0.87 = 0.13 cells (per axis) of the true centre.  So cell coordinates are
  // kept in 1/16 cells: 10 bits per axis, three axes in one word, tile-relative.  The point
  // word (built once per chunk and rotation) carries + 0.5 for the rounding, the lane word
  // (built once) the translation above the table's minimum; ONE integer add then yields all
  // three cell indices -- against 3 + 3 packed f32 operations and six roundings.  Error: both
  // words are rounded to 1/32 cell: 1/16 in total (+ 1e-5 of f32 arithmetic) < 0.13.
  constexpr int kFrac = 4;
  unsigned lane_word = 0;
  if (kGroups) {
    const float s = static_cast<float>(1 << kFrac);
    const unsigned tx = static_cast<unsigned>(rintf((tr.x * P.inv_resolution - TP.tr_lo[0]) * s));
    const unsigned ty = static_cast<unsigned>(rintf((tr.y * P.inv_resolution - TP.tr_lo[1]) * s));
    const unsigned tz = static_cast<unsigned>(rintf((tr.z * P.inv_resolution - TP.tr_lo[2]) * s));
    lane_word = tx | (ty << 10) | (tz << 20);
  }

  for (int chunk = chunk_begin + blockIdx.y; chunk < chunk_end; chunk += gridDim.y) {
    const int2 span = TP.chunks[chunk];
    const int len = span.y;
    if (kGroups) {
      // what the previous chunk added belongs to that chunk's segment (chunks ascend; lanes that
      // skip the lookups add nothing)
      const int previous = chunk - static_cast<int>(gridDim.y);
      const unsigned added = acc - acc_seen;
      acc_seen = acc;
      if (previous < split0) acc_seg0 += added;
      else if (previous < split1) acc_seg1 += added;
    }
    __syncthreads();                                     // the previous chunk is done with
    const bool have_box = TP.boxes != nullptr;
    if (have_box) {
      // The box comes from the pre-pass: no reduction, no barrier, and the points are rotated
      // ONCE, straight into the form the lookups read (step 2c).
      if (tid < 6) {
        const int block_index = kLists ? rotation_a / rotations : static_cast<int>(blockIdx.x);
        const float v = TP.boxes[(static_cast<size_t>(block_index) * num_chunks + chunk) * 6 + tid];
        const float offs[3] = {P.off_x, P.off_y, P.off_z};
        box[tid] = tid < 3 ? static_cast<int>(floorf(v + TP.tr_lo[tid] + offs[tid])) - 1
                           : static_cast<int>(ceilf(v + TP.tr_hi[tid - 3] + offs[tid - 3])) + 1;
      }
    } else {
    if (tid < 3) box[tid] = 0x7fffffff;
    else if (tid < 6) box[tid] = -0x7fffffff;
    __syncthreads();
    // 1. rotate the chunk by every rotation of the block; bounding box in cell coordinates
    {
      float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
      const int len_even = (len + 1) & ~1;               // (an odd chunk's pair partner is defined)
      for (int s = tid; s < num_rot * kChunk; s += blockDim.x) {
        const int w = s / kChunk, k = s % kChunk;      // (a power of two: shifts)
        if (k >= len_even) continue;
        const int i = span.x + min(k, len - 1);
        const float4 q4 = P.rotation[rotation_a + w];
        const F3 rp = Rotate(Quat{q4.w, q4.x, q4.y, q4.z},
                             F3{TP.sorted_xyz[3 * i], TP.sorted_xyz[3 * i + 1],
                                TP.sorted_xyz[3 * i + 2]});
        const float x = ClampStage(rp.x, P.t0x, P.lo_x, P.hi_x);
        const float y = ClampStage(rp.y, P.t0y, P.lo_y, P.hi_y);
        const float z = ClampStage(rp.z, P.t0z, P.lo_z, P.hi_z);
        float* dst = reinterpret_cast<float*>(stage + w * kStageStride) + 6 * (k >> 1) + (k & 1);
        dst[0] = x; dst[2] = y; dst[4] = z;
        const float c[3] = {x * P.inv_resolution, y * P.inv_resolution, z * P.inv_resolution};
#pragma unroll
        for (int a = 0; a < 3; ++a) { mn[a] = fminf(mn[a], c[a]); mx[a] = fmaxf(mx[a], c[a]); }
      }
#pragma unroll
      for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
          mn[a] = fminf(mn[a], __shfl_xor(mn[a], off, 64));
          mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off, 64));
        }
      }
      if ((tid & 63) == 0 && mn[0] <= mx[0]) {
        const float offs[3] = {P.off_x, P.off_y, P.off_z};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          // one cell of slack either side: the lookups round (c + tr) * inv + off, not this sum
          atomicMin(&box[a], static_cast<int>(floorf(mn[a] + TP.tr_lo[a] + offs[a])) - 1);
          atomicMax(&box[3 + a], static_cast<int>(ceilf(mx[a] + TP.tr_hi[a] + offs[a])) + 1);
        }
      }
    }
    }
    __syncthreads();
    // 2. the box, clipped to the brick (ClampStage keeps every lookup inside it), x aligned to 4
    const int lo_x = max(box[0], 0) & ~3, lo_y = max(box[1], 0), lo_z = max(box[2], 0);
    const int hi_x = min(box[3], TP.pitch_x - 1), hi_y = min(box[4], TP.pitch_y - 1),
              hi_z = min(box[5], TP.pitch_z - 1);
    const int dx = (hi_x - lo_x + 4) & ~3, dy = hi_y - lo_y + 1, dz = hi_z - lo_z + 1;
    // (a wave's DMA instruction writes 256 bytes: the last one may run past the box's end)
    const bool in_lds = dx > 0 && dx <= 256 && dy > 0 && dz > 0 &&
                        static_cast<long long>(dx) * dy * dz + 256 <= TP.tile_capacity;
    if (in_lds) {
      // The box enters LDS by LDS-DMA (global_load_lds_dword: lane l of a wave writes base + 4 l,
      // no registers, all loads of a wave in flight at once; ordered by vmcnt(0) + the barrier
      // below): dword e = row * (dx / 4) + xq of the tile, row = z * dy + y; the two divisions
      // through 2^32 reciprocals (exact: e, row < 2^16).  The rotation pass (2c) runs meanwhile.
      const unsigned qx = dx >> 2, total = qx * dy * dz;
      const unsigned rq = static_cast<unsigned>((1ull << 32) / qx) + 1u;
      const unsigned ry = static_cast<unsigned>((1ull << 32) / static_cast<unsigned>(dy)) + 1u;
      const unsigned row_q = TP.pitch_x >> 2, slice_q = TP.pitch_y * row_q;
      const uint32_t* base = reinterpret_cast<const uint32_t*>(P.cells) +
                             (static_cast<size_t>(lo_z) * TP.pitch_y + lo_y) * row_q + (lo_x >> 2);
      const unsigned wave_first = __builtin_amdgcn_readfirstlane(tid & ~63);
      auto* dst = (__attribute__((address_space(3))) unsigned char*)tile;
      for (unsigned first = wave_first; first < total; first += blockDim.x) {
        const unsigned e = min(first + (tid & 63), total - 1);       // (the tail re-reads the last dword)
        const unsigned row = qx == 1 ? e : __umulhi(e, rq);
        const unsigned xq = e - row * qx;
        const unsigned z = dy == 1 ? row : __umulhi(row, ry);
        const unsigned y = row - z * dy;
        const auto* src = (const __attribute__((address_space(1))) unsigned char*)(
            base + (z * slice_q + y * row_q + xq));
        __builtin_amdgcn_global_load_lds(src, dst + 4 * first, 4, 0, 0);
      }
    }
    // 2b. group pass: the staged points as fixed-point words relative to the tile
    const bool fixed = kGroups && TP.fixed_point && in_lds && dx <= 63 && dy <= 63 && dz <= 63;
    if (fixed && !have_box) {
      const float s = static_cast<float>(1 << kFrac);
      const float bx = P.off_x + TP.tr_lo[0] - static_cast<float>(lo_x) + 0.5f;
      const float by = P.off_y + TP.tr_lo[1] - static_cast<float>(lo_y) + 0.5f;
      const float bz = P.off_z + TP.tr_lo[2] - static_cast<float>(lo_z) + 0.5f;
      const int len_even = (len + 1) & ~1;
      for (int e = tid; e < num_rot * kChunk; e += blockDim.x) {
        const int w = e / kChunk, k = e % kChunk;
        if (k >= len_even) continue;
        const float* src = reinterpret_cast<const float*>(stage + w * kStageStride) +
                           6 * (k >> 1) + (k & 1);
        const unsigned px = static_cast<unsigned>(rintf(fmaf(src[0], P.inv_resolution, bx) * s));
        const unsigned py = static_cast<unsigned>(rintf(fmaf(src[2], P.inv_resolution, by) * s));
        const unsigned pz = static_cast<unsigned>(rintf(fmaf(src[4], P.inv_resolution, bz) * s));
        packed[w * kChunk + k] = px | (py << 10) | (pz << 20);
      }
    }
    // 2c. with the box from the pre-pass: ONE rotation pass, into fixed-point words or f32 triples
    if (have_box) {
      const float s = static_cast<float>(1 << kFrac);
      const float bx = P.off_x + TP.tr_lo[0] - static_cast<float>(lo_x) + 0.5f;
      const float by = P.off_y + TP.tr_lo[1] - static_cast<float>(lo_y) + 0.5f;
      const float bz = P.off_z + TP.tr_lo[2] - static_cast<float>(lo_z) + 0.5f;
      const int len_even = (len + 1) & ~1;
      for (int e = tid; e < num_rot * kChunk; e += blockDim.x) {
        const int w = e / kChunk, k = e % kChunk;
        if (k >= len_even) continue;
        const int i = span.x + min(k, len - 1);
        const float4 q4 = P.rotation[rotation_a + w];
        const F3 rp = Rotate(Quat{q4.w, q4.x, q4.y, q4.z},
                             F3{TP.sorted_xyz[3 * i], TP.sorted_xyz[3 * i + 1],
                                TP.sorted_xyz[3 * i + 2]});
        const float x = ClampStage(rp.x, P.t0x, P.lo_x, P.hi_x);
        const float y = ClampStage(rp.y, P.t0y, P.lo_y, P.hi_y);
        const float z = ClampStage(rp.z, P.t0z, P.lo_z, P.hi_z);
        if (fixed) {
          const unsigned px = static_cast<unsigned>(rintf(fmaf(x, P.inv_resolution, bx) * s));
          const unsigned py = static_cast<unsigned>(rintf(fmaf(y, P.inv_resolution, by) * s));
          const unsigned pz = static_cast<unsigned>(rintf(fmaf(z, P.inv_resolution, bz) * s));
          packed[w * kChunk + k] = px | (py << 10) | (pz << 20);
        } else {
          float* dst = reinterpret_cast<float*>(stage + w * kStageStride) + 6 * (k >> 1) + (k & 1);
          dst[0] = x; dst[2] = y; dst[4] = z;
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's LDS-DMA has landed
    __syncthreads();
    if (TP.stats != nullptr && tid == 0) {
      atomicAdd(&TP.stats[in_lds ? 0 : 1], 1ull);
      atomicAdd(&TP.stats[2], static_cast<unsigned long long>(dx) * dy * dz);
      atomicAdd(&TP.stats[3], static_cast<unsigned long long>(len));
    }
    if (!valid) continue;
    // 3. the lookups.  Cell indices are computed exactly as in Rt3DBulkKernel (the candidate
    //    pass's ambiguity argument rests on that expression); only the address differs.
    const v2f lx = {static_cast<float>(lo_x), static_cast<float>(lo_x)};
    const v2f ly = {static_cast<float>(lo_y), static_cast<float>(lo_y)};
    const v2f lz = {static_cast<float>(lo_z), static_cast<float>(lo_z)};
    const v2f dxf = {static_cast<float>(dx), static_cast<float>(dx)};
    const v2f dyf = {static_cast<float>(dy), static_cast<float>(dy)};
    const v2f px2 = {static_cast<float>(TP.pitch_x), static_cast<float>(TP.pitch_x)};
    const v2f py2 = {static_cast<float>(TP.pitch_y), static_cast<float>(TP.pitch_y)};
    // One pair of points (already in registers) -> two lookups.  kLds is a compile-time
    // property of the loop it runs in: the two address paths never share a loop body.
    const auto pair = [&](auto lds_tag, const v2f* pr, unsigned* v0, unsigned* v1, float* g) {
      constexpr bool kLds = decltype(lds_tag)::value;
      const v2f cx = pr[0] + trx, cy = pr[1] + try_, cz = pr[2] + trz;
      const v2f qx = __builtin_elementwise_fma(cx, inv, ofx);
      const v2f qy = __builtin_elementwise_fma(cy, inv, ofy);
      const v2f qz = __builtin_elementwise_fma(cz, inv, ofz);
      const v2f nx = {rintf(qx.x), rintf(qx.y)};
      const v2f ny = {rintf(qy.x), rintf(qy.y)};
      const v2f nz = {rintf(qz.x), rintf(qz.y)};
      if (!kGroups) {
        const v2f ex = qx - nx, ey = qy - ny, ez = qz - nz;
        float m = fmaxf(fmaxf(*g, fabsf(ex.x)), fabsf(ex.y));
        m = fmaxf(fmaxf(m, fabsf(ey.x)), fabsf(ey.y));
        *g = fmaxf(fmaxf(m, fabsf(ez.x)), fabsf(ez.y));
      }
      if (kLds) {
        const v2f o = __builtin_elementwise_fma(
            __builtin_elementwise_fma(nz - lz, dyf, ny - ly), dxf, nx - lx);
        *v0 = tile[static_cast<unsigned>(o.x)];
        *v1 = tile[static_cast<unsigned>(o.y)];
      } else {
        const v2f o = __builtin_elementwise_fma(__builtin_elementwise_fma(nz, py2, ny), px2, nx);
        *v0 = __builtin_amdgcn_raw_buffer_load_b8(rsrc, static_cast<unsigned>(o.x), 0, 0);
        *v1 = __builtin_amdgcn_raw_buffer_load_b8(rsrc, static_cast<unsigned>(o.y), 0, 0);
      }
    };
    const int len4 = len & ~3;
    // LDS returns in issue order: a wait for point data also waits for every gather issued
    // before that read.  So the points of iteration j + 1 are requested BEFORE the gathers of
    // iteration j are issued, and the gathers of iteration j are consumed in iteration j + 1:
    // every wait is for something issued a whole iteration earlier.
    const auto run = [&](auto lds_tag) {
      unsigned p0 = 0, p1 = 0, p2 = 0, p3 = 0;          // the previous four lookups, in flight
      v2f cur[6], nxt[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) cur[k] = my_stage[k];
#pragma unroll 2
      for (int j = 0; j < len4; j += 4) {
        const v2f* ahead = my_stage + 3 * (min(j + 4, kChunk - 4) >> 1);
#pragma unroll
        for (int k = 0; k < 6; ++k) nxt[k] = ahead[k];
        unsigned v[4];
        float g = 0.f;
        pair(lds_tag, cur, &v[0], &v[1], &g);
        pair(lds_tag, cur + 3, &v[2], &v[3], &g);
        if (!kGroups) ambiguous += g > guard ? 1u : 0u;
        acc += (p0 + p1) + (p2 + p3);
        p0 = v[0]; p1 = v[1]; p2 = v[2]; p3 = v[3];
#pragma unroll
        for (int k = 0; k < 6; ++k) cur[k] = nxt[k];
      }
      acc += (p0 + p1) + (p2 + p3);
      // the last 1..3 points: pairs whose second element may lie beyond the chunk (an odd
      // chunk staged its last point twice; the duplicate can only flag an ambiguity its twin
      // flags as well)
      for (int j = len4; j < len; j += 2) {
        unsigned v0, v1;
        float g = 0.f;
        pair(lds_tag, my_stage + 3 * (j >> 1), &v0, &v1, &g);
        acc += v0 + (j + 1 < len ? v1 : 0u);
        if (!kGroups) ambiguous += g > guard ? 1u : 0u;
      }
    };
    if (fixed) {
      const uint32_t* __restrict__ words = packed + (r - rotation_a) * kChunk;
      const unsigned udx = dx, udy = dy;
      const auto cell = [&](unsigned point_word) -> unsigned {
        const unsigned sum = point_word + lane_word;
        const unsigned x = (sum >> kFrac) & 63u, y = (sum >> (10 + kFrac)) & 63u,
                       z = sum >> (20 + kFrac);
        unsigned row, at;
        asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(row) : "v"(z), "s"(udy), "v"(y));
        asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(at) : "v"(row), "s"(udx), "v"(x));
        // (the tile starts at LDS address 0, checked at kernel entry: `at` IS the address)
        typedef __attribute__((address_space(3))) const uint8_t LdsByte;
        return *reinterpret_cast<LdsByte*>(static_cast<uintptr_t>(at));
      };
      // Hand-scheduled (as the window loop of rt_2d.hip): the compiler's version of this loop kept
      // four register copies per four lookups (its unroll-by-two failed) and loaded the next
      // point words right before their use -- an LDS round trip exposed in every iteration of
      // the pass the match spends half its time in.  Eight lookups per iteration in two halves
      // A and B.  LDS operations are issued in the order  A x 4, W0', B x 4, Wb'  (W': the point
      // words of the NEXT iteration) and return in issue order, so every wait names exactly how
      // many later operations may still be pending: A's gathers and the word loads land under
      // B's address arithmetic, B's gathers under the next iteration's A addresses.  Every
      // asynchronous result is waited for before the loop's back edge -- what crosses it (the A
      // addresses, the B words) are ordinary values the compiler may copy as it likes -- and the
      // waits carry the registers they release as in/out operands, which pins the consuming
      // arithmetic behind them.
      const auto address = [&](unsigned point_word) -> unsigned {
        const unsigned sum = point_word + lane_word;
        const unsigned x = (sum >> kFrac) & 63u, y = (sum >> (10 + kFrac)) & 63u,
                       z = sum >> (20 + kFrac);
        unsigned row, at;
        asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(row) : "v"(z), "s"(udy), "v"(y));
        asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(at) : "v"(row), "s"(udx), "v"(x));
        return at;
      };
      const int len8 = len & ~7;
      if (len8 > 0) {
        const unsigned wbase = static_cast<unsigned>(reinterpret_cast<uintptr_t>(
            (const __attribute__((address_space(3))) uint32_t*)words));
        uint4v Wa, Wb;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // nothing of the compiler's pending
        LdsRead128Asm(&Wa, wbase);
        LdsRead128Asm(&Wb, wbase + 16);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(Wa), "+v"(Wb));
        unsigned a0 = address(Wa.x), a1 = address(Wa.y), a2 = address(Wa.z), a3 = address(Wa.w);
        for (int j = 0; j < len8; j += 8) {
          const unsigned next = wbase + 4u * static_cast<unsigned>(min(j + 8, kChunk - 8));
          unsigned A0, A1, A2, A3, B0, B1, B2, B3;
          uint4v W0;
          LdsReadU8Asm(&A0, a0); LdsReadU8Asm(&A1, a1); LdsReadU8Asm(&A2, a2); LdsReadU8Asm(&A3, a3);
          LdsRead128Asm(&W0, next);
          const unsigned b0 = address(Wb.x), b1 = address(Wb.y), b2 = address(Wb.z),
                         b3 = address(Wb.w);
          LdsReadU8Asm(&B0, b0); LdsReadU8Asm(&B1, b1); LdsReadU8Asm(&B2, b2); LdsReadU8Asm(&B3, b3);
          // (Wb's words have been turned into addresses: the next ones go straight into it)
          asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(Wb) : "v"(next));
          asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(A0), "+v"(A1), "+v"(A2), "+v"(A3));   // after A: W0', B x 4, Wb'
          acc += (A0 + A1) + (A2 + A3);
          asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(W0));                                  // after W0': B x 4, Wb'
          a0 = address(W0.x); a1 = address(W0.y); a2 = address(W0.z); a3 = address(W0.w);
          // (the A addresses of the next iteration are operands too: their arithmetic is what
          // B's gathers land under, it must not sink below this wait)
          asm volatile("s_waitcnt lgkmcnt(0)"
                       : "+v"(B0), "+v"(B1), "+v"(B2), "+v"(B3), "+v"(Wb), "+v"(a0), "+v"(a1),
                         "+v"(a2), "+v"(a3));
          acc += (B0 + B1) + (B2 + B3);
        }
      }
      for (int j = len8; j < len; ++j) acc += cell(words[j]);
    } else if (in_lds) {
      run(std::true_type{});
    } else {
      run(std::false_type{});
    }
  }
  if (valid && (acc | ambiguous)) {
    const size_t c = static_cast<size_t>(r) * P.num_translations + t;
    atomicAdd(&P.sums[c].x, acc);
    if (ambiguous) atomicAdd(&P.sums[c].y, ambiguous);
    if (kGroups && TP.seg_sums != nullptr) {
      // the last chunk of this lane's loop: chunk_begin + blockIdx.y + k gridDim.y < chunk_end
      const int steps = (chunk_end - chunk_begin - static_cast<int>(blockIdx.y) +
                         static_cast<int>(gridDim.y) - 1) / static_cast<int>(gridDim.y);
      const int last = chunk_begin + static_cast<int>(blockIdx.y) +
                       (steps - 1) * static_cast<int>(gridDim.y);
      const unsigned added = acc - acc_seen;
      if (last < split0) acc_seg0 += added;
      else if (last < split1) acc_seg1 += added;
      if (acc_seg0) atomicAdd(&TP.seg_sums[c].x, acc_seg0);
      if (acc_seg1) atomicAdd(&TP.seg_sums[c].y, acc_seg1);
    }
  }
}

// CMX_RT3D_CROSSCHECK=1 (tests): the tiled passes against the memory-gather kernels, element by
// element -- group upper bounds (bitwise) and candidate sums Q.
__global__ void Rt3DCompareFloatsKernel(const float* __restrict__ a, const float* __restrict__ b,
                                        long long n, int* __restrict__ mismatches) {
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n && __float_as_uint(a[i]) != __float_as_uint(b[i])) atomicAdd(mismatches, 1);
}
__global__ void Rt3DCompareSumsKernel(const uint2* __restrict__ a, const uint2* __restrict__ b,
                                      long long n, int* __restrict__ mismatches) {
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n && a[i].x != b[i].x) atomicAdd(mismatches, 1);
}

// Group pass of the tiled path: weighted upper bounds from the accumulated sums.
__global__ void Rt3DSumBoundsKernel(Rt3DBulkParams P) {
  const long long c = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  const long long total = static_cast<long long>(P.num_rotations) * P.num_translations;
  float upper = 0.f;
  if (c < total) {
    const int r = static_cast<int>(c / P.num_translations);
    const int t = static_cast<int>(c - static_cast<long long>(r) * P.num_translations);
    float lower;
    Bounds3D(P, P.sums[c].x, 0u, P.translation[t].w, r, &lower, &upper);
    P.upper[c] = upper;
  }
  unsigned bits = __float_as_uint(upper);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) bits = max(bits, __shfl_xor(bits, off, 64));
  if ((threadIdx.x & 63) == 0 && bits != 0) atomicMax(P.max_upper_bits, bits);
}

// ---- the level above the groups: blocks of 2 x 2 x 2 ROTATIONS ----------------------------------
// A rotation block (rb, g) = the rotations of a 2 x 2 x 2 block of the angle-axis lattice x the
// translations of group g, bounded by ONE lookup per point at the block's centre rotation and the
// group's centre translation in the brick dilated TWICE (5 x 5 x 5): a member's rotation vector
// lies within half a step per axis of the centre vector, i.e. within 0.866 steps (x 1.03 for the
// non-commuting part while the window stays below 0.2 rad), a step moves the farthest point by
// one cell (step = 0.999 acos(1 - res^2 / 2 r_max^2)), so the member's rotated point lies within
// 0.89 cells of the centre's, + 0.87 for the translation block + 1/16 for the fixed-point cell:
// 1.82 < 2 cells per axis.  Weight: the smallest member angle and distance.  The (rotation, group)
// pairs of the blocks a threshold cannot exclude then go through the group pass proper
// (Rt3DTileKernel<true, true>, work lists as in the candidate passes); all other pairs keep an
// upper bound of -1: never selected.  As every bound here, these only SELECT what is scored.
//
// pair (r, g) := flagged for the group pass if its block's bound reaches threshold * factor and it
// has not been computed yet.
__global__ void Rt3DSelectPairsKernel(const float* __restrict__ block_upper, int num_groups,
                                      int num_rotations, int side_r, int blocks_per_axis,
                                      const unsigned* __restrict__ threshold_bits, float factor,
                                      uint8_t* __restrict__ computed, uint8_t* __restrict__ flags) {
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= static_cast<long long>(num_groups) * num_rotations) return;
  const int r = static_cast<int>(i / num_groups), g = static_cast<int>(i % num_groups);
  const int rx = r % side_r, ry = (r / side_r) % side_r, rz = r / (side_r * side_r);
  const int rb = ((rz >> 1) * blocks_per_axis + (ry >> 1)) * blocks_per_axis + (rx >> 1);
  const float threshold = __uint_as_float(*threshold_bits) * factor;
  if (computed[i] || !(block_upper[static_cast<size_t>(rb) * num_groups + g] >= threshold)) return;
  computed[i] = 1;
  flags[i] = 1;
}

// Rt3DSumBoundsKernel for a partly computed group table: pairs never computed get -1.
__global__ void Rt3DSumBoundsMaskedKernel(Rt3DBulkParams P, const uint8_t* __restrict__ computed) {
  const long long c = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  const long long total = static_cast<long long>(P.num_rotations) * P.num_translations;
  float upper = 0.f;
  if (c < total) {
    if (computed[c]) {
      const int r = static_cast<int>(c / P.num_translations);
      const int t = static_cast<int>(c - static_cast<long long>(r) * P.num_translations);
      float lower;
      Bounds3D(P, P.sums[c].x, 0u, P.translation[t].w, r, &lower, &upper);
      P.upper[c] = upper;
    } else {
      P.upper[c] = -1.f;
    }
  }
  unsigned bits = __float_as_uint(upper);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) bits = max(bits, __shfl_xor(bits, off, 64));
  if ((threadIdx.x & 63) == 0 && bits != 0) atomicMax(P.max_upper_bits, bits);
}

// Debug switch rt3d_verify: every computed (rotation, group) bound against its block's.
__global__ void Rt3DVerifyBlocksKernel(const float* __restrict__ block_upper,
                                       const float* __restrict__ group_upper,
                                       const uint8_t* __restrict__ computed, int num_groups,
                                       int num_rotations, int side_r, int blocks_per_axis,
                                       int* __restrict__ violations) {
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= static_cast<long long>(num_groups) * num_rotations || !computed[i]) return;
  const int r = static_cast<int>(i / num_groups), g = static_cast<int>(i % num_groups);
  const int rx = r % side_r, ry = (r / side_r) % side_r, rz = r / (side_r * side_r);
  const int rb = ((rz >> 1) * blocks_per_axis + (ry >> 1)) * blocks_per_axis + (rx >> 1);
  if (!(block_upper[static_cast<size_t>(rb) * num_groups + g] >= group_upper[i])) atomicAdd(violations, 1);
}

// 3 x 3 x 3 dilation of the padded brick in two passes (x, then y and z).  The halo is wider
// than one cell, so clamped neighbours at the array border read zeros like the cell itself.
__global__ void DilateXKernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int px,
                              size_t cells) {
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= cells) return;
  const int x = static_cast<int>(i % px);
  unsigned v = in[i];
  if (x > 0) v = max(v, static_cast<unsigned>(in[i - 1]));
  if (x + 1 < px) v = max(v, static_cast<unsigned>(in[i + 1]));
  out[i] = static_cast<uint8_t>(v);
}
__global__ void DilateYZKernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int px,
                               int py, int pz) {
  const size_t cells = static_cast<size_t>(px) * py * pz;
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= cells) return;
  const int x = static_cast<int>(i % px);
  const int y = static_cast<int>((i / px) % py);
  const int z = static_cast<int>(i / (static_cast<size_t>(px) * py));
  unsigned v = 0;
  for (int dz = -1; dz <= 1; ++dz)
    for (int dy = -1; dy <= 1; ++dy) {
      const int yy = min(max(y + dy, 0), py - 1), zz = min(max(z + dz, 0), pz - 1);
      v = max(v, static_cast<unsigned>(in[(static_cast<size_t>(zz) * py + yy) * px + x]));
    }
  out[i] = static_cast<uint8_t>(v);
}

// Groups whose weighted upper bound reaches the threshold and that have not been expanded yet:
// their member translations are flagged for the next candidate pass.
//   threshold = *threshold_bits (as float) * factor.
__global__ void Rt3DSelectGroupsKernel(const float* __restrict__ group_upper, int num_groups,
                                       int num_rotations, int side, int groups_per_axis,
                                       const unsigned* __restrict__ threshold_bits, float factor,
                                       uint8_t* __restrict__ expanded,
                                       uint8_t* __restrict__ flags, int num_translations) {
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= static_cast<long long>(num_groups) * num_rotations) return;
  const float threshold = __uint_as_float(*threshold_bits) * factor;
  if (expanded[i] || !(group_upper[i] >= threshold)) return;
  expanded[i] = 1;
  const int r = static_cast<int>(i / num_groups), g = static_cast<int>(i % num_groups);
  const int gx = g % groups_per_axis, gy = (g / groups_per_axis) % groups_per_axis,
            gz = g / (groups_per_axis * groups_per_axis);
  const int nx = min(2, side - 2 * gx), ny = min(2, side - 2 * gy), nz = min(2, side - 2 * gz);
  for (int c = 0; c < nz; ++c)
    for (int b = 0; b < ny; ++b)
      for (int a = 0; a < nx; ++a)
        flags[static_cast<size_t>(r) * num_translations +
              ((2 * gz + c) * side + (2 * gy + b)) * side + (2 * gx + a)] = 1;
}

// One block per work list (= `list_rotations` consecutive rotations): the flagged translations
// of its rotations, rotation by rotation in ascending order (x offsets fastest, so neighbouring
// lanes of the candidate pass read neighbouring cells), as entries w * T + t; flags cleared for
// the next round; one work descriptor (list, chunk) per `block_items` entries.
__global__ void __launch_bounds__(256)
Rt3DCompactKernel(uint8_t* __restrict__ flags, int num_translations, int num_rotations,
                  int list_rotations, int* __restrict__ counts, int* __restrict__ items,
                  int* __restrict__ total, int2* __restrict__ blocks,
                  int* __restrict__ num_blocks, int block_items) {
  __shared__ int wave_count[4];
  __shared__ int base;
  const int list = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int* out = items + static_cast<size_t>(list) * list_rotations * num_translations;
  if (tid == 0) base = 0;
  __syncthreads();
  for (int w = 0; w < list_rotations; ++w) {
    const int r = list * list_rotations + w;
    if (r >= num_rotations) break;
    uint8_t* f = flags + static_cast<size_t>(r) * num_translations;
    for (int t0 = 0; t0 < num_translations; t0 += 256) {
      const int t = t0 + tid;
      const bool on = t < num_translations && f[t] != 0;
      if (on) f[t] = 0;
      const unsigned long long mask = __ballot(on);
      if (lane == 0) wave_count[wave] = __popcll(mask);
      __syncthreads();
      int offset = base;
      for (int k = 0; k < wave; ++k) offset += wave_count[k];
      if (on) out[offset + __popcll(mask & ((1ull << lane) - 1ull))] = w * num_translations + t;
      __syncthreads();
      if (tid == 0) base += wave_count[0] + wave_count[1] + wave_count[2] + wave_count[3];
      __syncthreads();
    }
  }
  if (tid == 0) {
    counts[list] = base;
    if (base) {
      atomicAdd(total, base);
      const int chunks = (base + block_items - 1) / block_items;
      const int first = atomicAdd(num_blocks, chunks);
      for (int k = 0; k < chunks; ++k) blocks[first + k] = make_int2(list, k);
    }
  }
}

// Candidates whose weighted upper bound reaches the best weighted lower bound; `finalists`
// holds r * T + t (the bulk layout), unordered.  Candidates never scored have upper == 0.
__global__ void Rt3DBulkCollectKernel(const float* __restrict__ upper, long long num_candidates,
                                      const unsigned* __restrict__ max_lower_bits,
                                      int* __restrict__ count, int* __restrict__ finalists,
                                      int capacity) {
  const long long c = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (c >= num_candidates) return;
  if (upper[c] >= __uint_as_float(*max_lower_bits)) {
    const int slot = atomicAdd(count, 1);
    if (slot < capacity) finalists[slot] = static_cast<int>(c);
  }
}

// One block per finalist: the reference's own arithmetic (Rotate, + translation, lround of the
// IEEE quotient, padded f32 probability brick).  The N lookups of a candidate are independent,
// its f32 sum is a chain: waves 1..3 fetch the next 4096 probabilities into one LDS buffer while
// lane 0 of wave 0 runs the chain over the other (ChainSumLds, cmx_device.h).
constexpr int kExact3DChunk = 4096;
__global__ void __launch_bounds__(256)
Rt3DExactKernel(Rt3DParams P, const float* __restrict__ xyz, int n,
                const int* __restrict__ finalists, const int* __restrict__ count, int capacity,
                float* __restrict__ exact) {
  __shared__ __attribute__((aligned(16))) float prob[2][kExact3DChunk];
  const int f = blockIdx.x;
  if (f >= min(*count, capacity)) return;
  const int c = finalists[f];
  const int r = c / P.num_translations, t = c - r * P.num_translations;
  const float4 q4 = P.rotation[r];
  const Quat q{q4.w, q4.x, q4.y, q4.z};
  const float4 tr = P.translation[t];
  const float res = P.resolution;
  const int sx = P.grid.nx + 2, sy = P.grid.ny + 2;
  const int ox = 1 - P.grid.lo_x, oy = 1 - P.grid.lo_y, oz = 1 - P.grid.lo_z;
  const int mx = P.grid.nx + 1, my = P.grid.ny + 1, mz = P.grid.nz + 1;
  const auto fetch = [&](int base, float* out, int first, int stride) {
    const int cnt = min(kExact3DChunk, n - base);
    for (int j = first; j < cnt; j += stride) {
      const int i = base + j;
      const F3 rp = Rotate(q, F3{xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]});
      const int3 idx = CellIndex3(F3{rp.x + tr.x, rp.y + tr.y, rp.z + tr.z}, res);
      const int ix = min(max(idx.x + ox, 0), mx);
      const int iy = min(max(idx.y + oy, 0), my);
      const int iz = min(max(idx.z + oz, 0), mz);
      out[j] = P.grid.cells[(static_cast<size_t>(iz) * sy + iy) * sx + ix];
    }
    // (the chain runs over whole banks of 64: + 0 leaves the non-negative sum as it is)
    for (int j = cnt + first; j < ((cnt + 63) & ~63); j += stride) out[j] = 0.f;
  };
  float acc = 0.f;
  fetch(0, prob[0], threadIdx.x, blockDim.x);
  __syncthreads();
  int b = 0;
  for (int base = 0; base < n; base += kExact3DChunk, b ^= 1) {
    if (threadIdx.x >= 64) {
      if (base + kExact3DChunk < n) fetch(base + kExact3DChunk, prob[b ^ 1], threadIdx.x - 64,
                                          blockDim.x - 64);
    } else if (threadIdx.x == 0) {
      const int cnt = min(kExact3DChunk, n - base);
      acc = ChainSumLds(prob[b], (cnt + 63) & ~63, acc);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) exact[f] = acc / static_cast<float>(n);
}

// uint16 voxel value -> q = (max(value & 32767, 1) - 1) >> 7 in the padded bricks: row-major
// (to be dilated for the group pass) and tiled (candidate pass).
__global__ void ScatterBulkKernel(const cmx_voxel* __restrict__ voxels, long long n, int lo_x,
                                  int lo_y, int lo_z, int pad, int px, int py, int tiles_y,
                                  int pitch_z, uint8_t* __restrict__ rows,
                                  uint8_t* __restrict__ tiled) {
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const cmx_voxel v = voxels[i];
  const unsigned x = v.x - lo_x + pad, y = v.y - lo_y + pad, z = v.z - lo_z + pad;
  const unsigned value = v.value & 32767u;
  const uint8_t q = static_cast<uint8_t>((max(value, 1u) - 1u) >> 7);
  rows[(static_cast<size_t>(z) * py + y) * px + x] = q;
  tiled[TiledOffset(x, y, z, tiles_y, pitch_z)] = q;
}

// The same two conversions from a HybridGrid that already lives in HBM as a dense uint16 brick
// (cmx_grid3d): one thread per cell of that brick.
__global__ void BrickProbabilitiesKernel(const uint16_t* __restrict__ cells, long long n, int nx,
                                         int ny, PaddedBrick b, float* __restrict__ out) {
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned value = cells[i];
  if (value == 0) return;
  const int x = static_cast<int>(i % nx), y = static_cast<int>((i / nx) % ny),
            z = static_cast<int>(i / (static_cast<long long>(nx) * ny));
  out[(static_cast<size_t>(z + 1) * (b.ny + 2) + (y + 1)) * (b.nx + 2) + (x + 1)] =
      ValueToProbabilityDev(value);
}
__global__ void BrickBulkKernel(const uint16_t* __restrict__ cells, long long n, int nx, int ny,
                                int pad, int px, int py, int tiles_y, int pitch_z,
                                uint8_t* __restrict__ rows, uint8_t* __restrict__ tiled) {
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned raw = cells[i];
  if (raw == 0) return;
  const unsigned x = static_cast<unsigned>(i % nx) + pad,
                 y = static_cast<unsigned>((i / nx) % ny) + pad,
                 z = static_cast<unsigned>(i / (static_cast<long long>(nx) * ny)) + pad;
  const unsigned value = raw & 32767u;
  const uint8_t q = static_cast<uint8_t>((max(value, 1u) - 1u) >> 7);
  rows[(static_cast<size_t>(z) * py + y) * px + x] = q;
  tiled[TiledOffset(x, y, z, tiles_y, pitch_z)] = q;
}

// Debug switch rt3d_legacy keeps every candidate on the one-thread-per-candidate kernel,
// rt3d_no_tiles keeps the bulk passes on the memory gathers (Rt3DBulkKernel); parity tests run
// every path.
bool Tiles3DEnabled() { return Debug().rt3d_no_tiles == 0; }
bool Bulk3DEnabled() { return Debug().rt3d_legacy == 0; }
// A tuning override (debug switches of the profiling tools): 0 = the default.
int Override(int value, int fallback) { return value > 0 ? value : fallback; }

}  // namespace

bool VoxelBounds(const cmx_voxel* voxels, int64_t n, int lo[3], int hi[3]) {
  if (n <= 0) return false;
  lo[0] = hi[0] = voxels[0].x; lo[1] = hi[1] = voxels[0].y; lo[2] = hi[2] = voxels[0].z;
  for (int64_t i = 1; i < n; ++i) {
    lo[0] = std::min(lo[0], voxels[i].x); hi[0] = std::max(hi[0], voxels[i].x);
    lo[1] = std::min(lo[1], voxels[i].y); hi[1] = std::max(hi[1], voxels[i].y);
    lo[2] = std::min(lo[2], voxels[i].z); hi[2] = std::max(hi[2], voxels[i].z);
  }
  return true;
}

int GridSizeOf(const cmx_voxel* voxels, int64_t n) {
  int gs = 128;
  int lo[3], hi[3];
  if (!VoxelBounds(voxels, n, lo, hi)) return gs;
  auto fits = [&](int g) {
    const int h = g / 2;
    for (int k = 0; k < 3; ++k)
      if (lo[k] < -h || hi[k] >= h) return false;
    return true;
  };
  while (!fits(gs)) gs *= 2;
  return gs;
}

void BuildBrickFromVoxels(Workspace& ws, const cmx_voxel* voxels, int64_t n, int bytes_per_cell,
                          DeviceBrick* out) {
  int lo[3], hi[3];
  if (!VoxelBounds(voxels, n, lo, hi)) {
    lo[0] = lo[1] = lo[2] = 0;
    hi[0] = hi[1] = hi[2] = 0;
  }
  for (int k = 0; k < 3; ++k) {
    CMX_REQUIRE(lo[k] > -(1 << 20) && hi[k] < (1 << 20), "voxel index out of range");
  }
  Brick b{};
  b.lo_x = lo[0]; b.lo_y = lo[1]; b.lo_z = lo[2];
  b.nx = hi[0] - lo[0] + 1; b.ny = hi[1] - lo[1] + 1; b.nz = hi[2] - lo[2] + 1;
  const size_t cells = static_cast<size_t>(b.nx) * b.ny * b.nz;
  CMX_REQUIRE(cells * bytes_per_cell < (size_t(8) << 30),
              "dense grid of %d x %d x %d cells is too large", b.nx, b.ny, b.nz);
  out->bytes = cells * bytes_per_cell;
  CMX_HIP(hipMalloc(&out->mem, out->bytes + 16));   // (+16: aligned 8-byte reads of the last cells)
  b.cells = out->mem;
  out->desc = b;
  CMX_HIP(hipMemsetAsync(out->mem, 0, out->bytes, ws.stream));
  if (n > 0) {
    cmx_voxel* d_vox = ws.dev[15].ReserveAs<cmx_voxel>(n);
    CMX_HIP(hipMemcpyAsync(d_vox, voxels, n * sizeof(cmx_voxel), hipMemcpyHostToDevice,
                           ws.stream));
    ScatterVoxelsKernel<<<DivUp(n, 256), 256, 0, ws.stream>>>(d_vox, n, b, bytes_per_cell);
    CMX_HIP(hipGetLastError());
  }
  CMX_HIP(hipStreamSynchronize(ws.stream));   // `voxels` is borrowed host memory
}

}  // namespace cmx

namespace cmx {
namespace {
// RealTimeCorrelativeScanMatcher3D::Match.  The HybridGrid is either the voxel list (host
// memory) or, when `resident` is set, the dense uint16 brick cmx_grid3d keeps in HBM
// (`voxels` unused then).
cmx_status Rt3DMatchImpl(const cmx_rt_options* options, float grid_resolution,
                         const cmx_voxel* voxels, int64_t num_voxels, const Brick* resident,
                         const cmx_pose3d* initial_pose_estimate, const float* point_cloud_xyz,
                         int32_t num_points, int32_t device, float* score,
                         cmx_pose3d* pose_estimate, cmx_match_stats* stats) {
  return Guard([&] {
    CMX_REQUIRE(options && initial_pose_estimate && point_cloud_xyz, "null argument");
    CMX_REQUIRE(pose_estimate != nullptr && score != nullptr,
                "pose_estimate must not be null");                       // CHECK at :39
    CMX_REQUIRE(resident != nullptr || num_voxels == 0 || voxels != nullptr, "voxels is null");
    CMX_REQUIRE(num_points >= 1 && num_points <= (1 << 24), "bad point count");
    CMX_REQUIRE(grid_resolution > 0.f, "resolution must be > 0");
    const int n = num_points;
    const float resolution = grid_resolution;

    // GenerateExhaustiveSearchTransforms (:55-95), host side.
    const int L = static_cast<int>(std::lround(options->linear_search_window / resolution));
    float max_scan_range = 3.f * resolution;
    for (int i = 0; i < n; ++i) {
      const h3::V3 p{point_cloud_xyz[3 * i], point_cloud_xyz[3 * i + 1],
                     point_cloud_xyz[3 * i + 2]};
      max_scan_range = std::max(h3::Norm(p), max_scan_range);
    }
    const float kSafetyMargin = 1.f - 1e-3f;
    const float step =
        kSafetyMargin * std::acos(1.f - (resolution * (resolution * 1.f)) /
                                            (2.f * (max_scan_range * (max_scan_range * 1.f))));
    const int A = static_cast<int>(std::lround(options->angular_search_window / step));
    CMX_REQUIRE(L >= 0 && L < 512 && A >= 0 && A < 64, "unsupported search window");
    const int side_t = 2 * L + 1, side_r = 2 * A + 1;
    const long long T = 1ll * side_t * side_t * side_t, R = 1ll * side_r * side_r * side_r;
    const long long num_candidates = T * R;
    CMX_REQUIRE(num_candidates < (1ll << 31), "search window too large");

    const h3::Rigid init = h3::FromPose(*initial_pose_estimate);
    std::vector<float4> rot(R), trans(T);
    std::vector<float> angle(R);
    std::vector<h3::Q> rot_q(R);
    {
      long long k = 0;
      for (int rz = -A; rz <= A; ++rz)
        for (int ry = -A; ry <= A; ++ry)
          for (int rx = -A; rx <= A; ++rx, ++k) {
            h3::Rigid tf;
            tf.q = h3::FromAngleAxisVector({rx * step, ry * step, rz * step});
            rot_q[k] = tf.q;
            angle[k] = h3::GetAngle(tf);
            const h3::Q q = h3::Normalized(h3::Mul(init.q, tf.q));
            rot[k] = make_float4(q.x, q.y, q.z, q.w);
          }
      k = 0;
      for (int z = -L; z <= L; ++z)
        for (int y = -L; y <= L; ++y)
          for (int x = -L; x <= L; ++x, ++k) {
            const h3::V3 tc{x * resolution, y * resolution, z * resolution};
            const h3::V3 r = h3::Rotate(init.q, tc);
            trans[k] = make_float4(r.x + init.t.x, r.y + init.t.y, r.z + init.t.z, h3::Norm(tc));
          }
    }

    // Rotation blocks (2 x 2 x 2 of the angle-axis lattice, see Rt3DSelectPairsKernel): the
    // centre rotation and the smallest member angle of each.
    const int Ab = (side_r + 1) / 2;
    const long long Rb = 1ll * Ab * Ab * Ab;
    std::vector<float4> rot_b(Rb);
    std::vector<float> angle_b(Rb);
    {
      long long k = 0;
      for (int bz = 0; bz < Ab; ++bz)
        for (int by = 0; by < Ab; ++by)
          for (int bx = 0; bx < Ab; ++bx, ++k) {
            const int nx = std::min(2, side_r - 2 * bx), ny = std::min(2, side_r - 2 * by),
                      nz = std::min(2, side_r - 2 * bz);
            h3::Rigid tf;
            tf.q = h3::FromAngleAxisVector({(2 * bx + 0.5f * (nx - 1) - A) * step,
                                            (2 * by + 0.5f * (ny - 1) - A) * step,
                                            (2 * bz + 0.5f * (nz - 1) - A) * step});
            const h3::Q q = h3::Normalized(h3::Mul(init.q, tf.q));
            rot_b[k] = make_float4(q.x, q.y, q.z, q.w);
            float smallest = INFINITY;
            for (int c = 0; c < nz; ++c)
              for (int b = 0; b < ny; ++b)
                for (int a = 0; a < nx; ++a)
                  smallest = std::min(smallest, angle[((2 * bz + c) * side_r + (2 * by + b)) * side_r + (2 * bx + a)]);
            angle_b[k] = smallest;
          }
    }

    WorkspaceLease ws(device);
    // Padded f32 probability brick over the voxels' bounding box.
    PaddedBrick brick{};
    {
      int lo[3], hi[3];
      if (resident != nullptr) {             // the brick's own box (cells never written hold 0)
        lo[0] = resident->lo_x; lo[1] = resident->lo_y; lo[2] = resident->lo_z;
        hi[0] = lo[0] + resident->nx - 1; hi[1] = lo[1] + resident->ny - 1;
        hi[2] = lo[2] + resident->nz - 1;
      } else if (!VoxelBounds(voxels, num_voxels, lo, hi)) {
        lo[0] = lo[1] = lo[2] = 0;
        hi[0] = hi[1] = hi[2] = 0;
      }
      for (int k = 0; k < 3; ++k)
        CMX_REQUIRE(lo[k] > -(1 << 20) && hi[k] < (1 << 20), "voxel index out of range");
      brick.lo_x = lo[0]; brick.lo_y = lo[1]; brick.lo_z = lo[2];
      brick.nx = hi[0] - lo[0] + 1; brick.ny = hi[1] - lo[1] + 1; brick.nz = hi[2] - lo[2] + 1;
      const size_t cells = static_cast<size_t>(brick.nx + 2) * (brick.ny + 2) * (brick.nz + 2);
      CMX_REQUIRE(cells < (size_t(1) << 29), "dense grid of %d x %d x %d cells is too large",
                  brick.nx, brick.ny, brick.nz);
      float* d_cells = ws->dev[7].ReserveAs<float>(cells);
      brick.cells = d_cells;
      FillFloatKernel<<<DivUp(cells, 256), 256, 0, ws->stream>>>(d_cells, cells, 0.1f);
      if (resident != nullptr) {
        const long long count = static_cast<long long>(resident->nx) * resident->ny * resident->nz;
        BrickProbabilitiesKernel<<<DivUp(count, 256), 256, 0, ws->stream>>>(
            static_cast<const uint16_t*>(resident->cells), count, resident->nx, resident->ny, brick,
            d_cells);
      } else if (num_voxels > 0) {
        cmx_voxel* d_vox = ws->dev[15].ReserveAs<cmx_voxel>(num_voxels);
        CMX_HIP(hipMemcpyAsync(d_vox, voxels, num_voxels * sizeof(cmx_voxel),
                               hipMemcpyHostToDevice, ws->stream));
        ScatterProbabilitiesKernel<<<DivUp(num_voxels, 256), 256, 0, ws->stream>>>(
            d_vox, num_voxels, brick, d_cells);
      }
      CMX_HIP(hipGetLastError());
    }

    float* d_xyz = ws->dev[0].ReserveAs<float>(3 * static_cast<size_t>(n));
    float4* d_rot = ws->dev[1].ReserveAs<float4>(R);
    float4* d_trans = ws->dev[2].ReserveAs<float4>(T);
    float* d_angle = ws->dev[3].ReserveAs<float>(R);
    float* d_unweighted = ws->dev[4].ReserveAs<float>(num_candidates);
    float* d_weighted = ws->dev[5].ReserveAs<float>(num_candidates);
    const int kFinalistCap = 4096;
    int num_finalists = 0;
    char* d_misc = static_cast<char*>(ws->dev[6].Reserve(16 + sizeof(long long) * kFinalistCap));
    unsigned* d_max = reinterpret_cast<unsigned*>(d_misc);
    int* d_count = reinterpret_cast<int*>(d_misc + 4);
    long long* d_finalists = reinterpret_cast<long long*>(d_misc + 16);
    char* h_misc = static_cast<char*>(ws->pinned[1].Reserve(16 + sizeof(long long) * kFinalistCap));

    CMX_HIP(hipMemcpyAsync(d_xyz, point_cloud_xyz, 3 * sizeof(float) * n, hipMemcpyHostToDevice,
                           ws->stream));
    CMX_HIP(hipMemcpyAsync(d_rot, rot.data(), R * sizeof(float4), hipMemcpyHostToDevice,
                           ws->stream));
    CMX_HIP(hipMemcpyAsync(d_trans, trans.data(), T * sizeof(float4), hipMemcpyHostToDevice,
                           ws->stream));
    CMX_HIP(hipMemcpyAsync(d_angle, angle.data(), R * sizeof(float), hipMemcpyHostToDevice,
                           ws->stream));
    float4* d_rot_b = ws->dev[24].ReserveAs<float4>(Rb);
    float* d_angle_b = ws->dev[25].ReserveAs<float>(Rb);
    CMX_HIP(hipMemcpyAsync(d_rot_b, rot_b.data(), Rb * sizeof(float4), hipMemcpyHostToDevice,
                           ws->stream));
    CMX_HIP(hipMemcpyAsync(d_angle_b, angle_b.data(), Rb * sizeof(float), hipMemcpyHostToDevice,
                           ws->stream));
    CMX_HIP(hipMemsetAsync(d_misc, 0, 16, ws->stream));
    // pageable sources above: make sure the copies are done before they go away
    CMX_HIP(hipStreamSynchronize(ws->stream));

    Rt3DParams P;
    P.grid = brick;
    P.resolution = resolution;
    P.inv_resolution = 1.f / resolution;
    P.num_translations = static_cast<int>(T);
    P.num_rotations = static_cast<int>(R);
    P.side_t = side_t; P.side_r = side_r;
    P.rotation = d_rot; P.translation = d_trans; P.rotation_angle = d_angle;
    P.wt = options->translation_delta_cost_weight;
    P.wr = options->rotation_delta_cost_weight;

    std::vector<long long> finalists;     // reference candidate index t * R + r, ascending
    std::vector<float> acc;               // their exact unweighted scores
    bool done = false;
    RecordEvent(ws->ev_begin, ws->stream);

    // ---- integer bounds (groups, then candidates) + exact finalists --------------------
    // Limits: finite points (a NaN would read an unchecked cell), N small enough for the
    // rounding bound of the f32 chain to mean something, index arithmetic exact in f32,
    // weights that decrease with distance (a group is weighted by its nearest member).
    const int reach = static_cast<int>(std::ceil(L * 1.7320508075688772)) + 1;   // |init.q * t_c|
    const int pad = 2 * reach + 2;
    // (rows a multiple of 4 cells: the tiled passes copy boxes of the brick with aligned dwords)
    const long long bx = (brick.nx + 2ll * pad + 3) & ~3ll, by = brick.ny + 2ll * pad,
                    bz = brick.nz + 2ll * pad;
    bool use_bulk = Bulk3DEnabled() && n <= (1 << 21) && R <= 65535 &&
                    bx * by * bz < (1ll << 24) &&
                    options->translation_delta_cost_weight >= 0. &&
                    options->rotation_delta_cost_weight >= 0.;
    for (int i = 0; i < 3 * n && use_bulk; ++i) use_bulk = std::isfinite(point_cloud_xyz[i]);
    long long bounds_evaluated = 0;
    long long group_bounds = 0;          // bounds above the candidates: rotation blocks + (rotation, group) pairs
    bool second_pairs_pass = false;      // (its span: ev_x0 .. ev_x1)
    if (use_bulk) {
      const size_t cells = static_cast<size_t>(bx * by * bz);
      const int tiles_x = DivUp(bx, 8), tiles_y = DivUp(by, 4);
      const int pitch_z = DivUp(bz, 4) * 4;
      const size_t tiled_cells = static_cast<size_t>(tiles_x) * tiles_y * pitch_z * 32;
      // [row-major q (only the input of the dilation) | tiled q]
      uint8_t* d_bulk = ws->dev[8].ReserveAs<uint8_t>(cells + 128 + tiled_cells);
      uint8_t* d_tiled = d_bulk + (cells + 127) / 128 * 128;
      uint8_t* d_dilated = ws->dev[9].ReserveAs<uint8_t>(cells);
      // (dilation scratch first, then the (Q, A) pairs of the candidate pass)
      uint8_t* d_tmp = ws->dev[10].ReserveAs<uint8_t>(
          std::max<size_t>(cells, sizeof(uint2) * static_cast<size_t>(num_candidates)));
      CMX_HIP(hipMemsetAsync(d_bulk, 0, (d_tiled - d_bulk) + tiled_cells, ws->stream));
      if (resident != nullptr) {
        const long long count = static_cast<long long>(resident->nx) * resident->ny * resident->nz;
        BrickBulkKernel<<<DivUp(count, 256), 256, 0, ws->stream>>>(
            static_cast<const uint16_t*>(resident->cells), count, resident->nx, resident->ny, pad,
            static_cast<int>(bx), static_cast<int>(by), tiles_y, pitch_z, d_bulk, d_tiled);
      } else if (num_voxels > 0) {
        const cmx_voxel* d_vox = static_cast<const cmx_voxel*>(ws->dev[15].get());
        ScatterBulkKernel<<<DivUp(num_voxels, 256), 256, 0, ws->stream>>>(
            d_vox, num_voxels, brick.lo_x, brick.lo_y, brick.lo_z, pad, static_cast<int>(bx),
            static_cast<int>(by), tiles_y, pitch_z, d_bulk, d_tiled);
      }
      DilateXKernel<<<DivUp(cells, 256), 256, 0, ws->stream>>>(d_bulk, d_tmp,
                                                               static_cast<int>(bx), cells);
      DilateYZKernel<<<DivUp(cells, 256), 256, 0, ws->stream>>>(
          d_tmp, d_dilated, static_cast<int>(bx), static_cast<int>(by), static_cast<int>(bz));

      // 2 x 2 x 2 blocks of lattice steps: centre translation + the smallest member distance.
      const int gpa = (side_t + 1) / 2;
      const int G = gpa * gpa * gpa;
      std::vector<float4> group(G);
      for (int gz = 0, g = 0; gz < gpa; ++gz)
        for (int gy = 0; gy < gpa; ++gy)
          for (int gx = 0; gx < gpa; ++gx, ++g) {
            const int nx = std::min(2, side_t - 2 * gx), ny = std::min(2, side_t - 2 * gy),
                      nz = std::min(2, side_t - 2 * gz);
            const h3::V3 tc{(2 * gx + 0.5f * (nx - 1) - L) * resolution,
                            (2 * gy + 0.5f * (ny - 1) - L) * resolution,
                            (2 * gz + 0.5f * (nz - 1) - L) * resolution};
            const h3::V3 rc = h3::Rotate(init.q, tc);
            float nearest = INFINITY;
            for (int c = 0; c < nz; ++c)
              for (int b = 0; b < ny; ++b)
                for (int a = 0; a < nx; ++a)
                  nearest = std::min(
                      nearest,
                      trans[((2 * gz + c) * side_t + (2 * gy + b)) * side_t + (2 * gx + a)].w);
            group[g] = make_float4(rc.x + init.t.x, rc.y + init.t.y, rc.z + init.t.z, nearest);
          }

      const int kBulkFinalistCap = 4096;
      const long long RG = R * G;
      float4* d_group = ws->dev[14].ReserveAs<float4>(G);
      float* d_group_upper = ws->dev[11].ReserveAs<float>(RG);
      uint8_t* d_expanded = ws->dev[12].ReserveAs<uint8_t>(RG + num_candidates);
      uint8_t* d_flags = d_expanded + RG;                             // [R][T]
      float* d_upper = d_weighted;                                   // [R][T], reused
      int* d_items = reinterpret_cast<int*>(d_unweighted);           // [R][T], reused
      const size_t head_bytes = 16 + (sizeof(int) + sizeof(float)) * kBulkFinalistCap;   // 16 | 32768
      // work descriptors of a round: at most one per 256 translations and rotation, + count
      const int max_blocks = static_cast<int>(R) * DivUp(T, kCand3DThreads);   // (>= any block size used)
      const size_t counts_bytes = (sizeof(int) * R + 15) / 16 * 16;
      char* d_bmisc = static_cast<char*>(ws->dev[13].Reserve(
          head_bytes + counts_bytes + sizeof(int2) * (max_blocks + 2)));
      unsigned* d_max_lower = reinterpret_cast<unsigned*>(d_bmisc);
      int* d_bcount = reinterpret_cast<int*>(d_bmisc + 4);
      unsigned* d_max_upper = reinterpret_cast<unsigned*>(d_bmisc + 8);
      int* d_total = reinterpret_cast<int*>(d_bmisc + 12);
      int* d_bfinalists = reinterpret_cast<int*>(d_bmisc + 16);
      float* d_exact = reinterpret_cast<float*>(d_bmisc + 16 + sizeof(int) * kBulkFinalistCap);
      int* d_counts = reinterpret_cast<int*>(d_bmisc + head_bytes);
      int2* d_blocks = reinterpret_cast<int2*>(d_bmisc + head_bytes + counts_bytes);
      int* d_num_blocks = reinterpret_cast<int*>(d_blocks + max_blocks);
      int* d_violations = d_num_blocks + 1;          // (the second word of that int2 slot)
      int* d_stage_scratch = d_num_blocks + 2;       // item total of the staged re-compactions (unused)
      const bool verify = Debug().rt3d_verify != 0;
      int* h_num_blocks = ws->pinned[1].ReserveAs<int>(16);
      char* h_bmisc = static_cast<char*>(ws->pinned[2].Reserve(head_bytes));
      float4* h_group = ws->pinned[3].ReserveAs<float4>(G);
      std::memcpy(h_group, group.data(), sizeof(float4) * G);
      CMX_HIP(hipMemcpyAsync(d_group, h_group, sizeof(float4) * G, hipMemcpyHostToDevice,
                             ws->stream));
      CMX_HIP(hipMemsetAsync(d_bmisc, 0, 16, ws->stream));
      CMX_HIP(hipMemsetAsync(d_num_blocks, 0, sizeof(int) * 4, ws->stream));   // blocks, violations, totals
      CMX_HIP(hipMemsetAsync(d_counts, 0, sizeof(int) * R, ws->stream));
      CMX_HIP(hipMemsetAsync(d_expanded, 0, RG + num_candidates, ws->stream));
      CMX_HIP(hipMemsetAsync(d_upper, 0, sizeof(float) * num_candidates, ws->stream));

      Rt3DBulkParams B{};
      B.cell_count = static_cast<unsigned>(cells);
      B.pitch_x = static_cast<float>(bx); B.pitch_y = static_cast<float>(by);
      B.off_x = static_cast<float>(pad - brick.lo_x);
      B.off_y = static_cast<float>(pad - brick.lo_y);
      B.off_z = static_cast<float>(pad - brick.lo_z);
      B.inv_resolution = 1.f / resolution;
      double q_abs = 0.;
      const int los[3] = {brick.lo_x, brick.lo_y, brick.lo_z};
      const int dims[3] = {brick.nx, brick.ny, brick.nz};
      for (int k = 0; k < 3; ++k)
        q_abs = std::max<double>(q_abs, std::max(std::abs(los[k] - pad),
                                                 std::abs(los[k] + dims[k] - 1 + pad)));
      const double q_pad = static_cast<double>(std::max(bx, std::max(by, bz)));
      // |q - (qe + off)| <= (2 |c / res| + |q|) 2^-24 (rounded 1/res, one fma rounding, the
      // reference's rounded quotient qe); 25 % on top.
      const double e = (2. * q_abs + q_pad + 4.) * 0x1p-24 * 1.25;
      B.guard = std::nextafter(static_cast<float>(0.5 - e), 0.f);
      B.t0x = init.t.x; B.t0y = init.t.y; B.t0z = init.t.z;
      B.lo_x = (brick.lo_x - reach - 1) * resolution;
      B.hi_x = (brick.lo_x + brick.nx + reach) * resolution;
      B.lo_y = (brick.lo_y - reach - 1) * resolution;
      B.hi_y = (brick.lo_y + brick.ny + reach) * resolution;
      B.lo_z = (brick.lo_z - reach - 1) * resolution;
      B.hi_z = (brick.lo_z + brick.nz + reach) * resolution;
      B.num_rotations = static_cast<int>(R);
      B.n = n;
      B.rotation = d_rot; B.rotation_angle = d_angle;
      B.wt = options->translation_delta_cost_weight;
      B.wr = options->rotation_delta_cost_weight;
      const float kMinP = 0.1f, kMaxP = 1.f - kMinP;
      const float scale = (kMaxP - kMinP) / (32768 - 2.f);
      B.min_probability = kMinP;
      B.scale_over_n = 128. * static_cast<double>(scale) / n;
      B.slack_hi = 127. * static_cast<double>(scale) + 2e-7;
      const double ku = (n + 8.) * 0x1p-24;
      B.delta = ku / (1. - ku) * 1.01 + 2e-6;
      B.max_lower_bits = d_max_lower; B.max_upper_bits = d_max_upper;

      // Group pass: every rotation, every block of translations, on the dilated brick.
      Rt3DBulkParams BG = B;
      BG.cells = d_dilated;
      BG.translation = d_group; BG.num_translations = G;
      BG.upper = d_group_upper;
      StageTrace trace(ws->stream);
      trace.Mark("bricks");
      int cus = 256;
      (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
      // Tiled passes: the cloud sorted into bins of kTileBin^3 cells at the central candidate.
      // Rotations of a group-pass workgroup: as many as fit 512 lanes, so that TWO workgroups
      // share a CU and one computes while the other stages its next chunk (C4, 216 groups: two
      // rotations = 448 lanes, group pass 5.51 -> 4.91 ms against four rotations in one 896-lane
      // workgroup per CU; debug switch rt3d_group_rotations overrides, experiments).
      const int rot_per_block = std::max(
          1, std::min({kTileMaxRotations, std::max(1, 512 / std::max(G, 1)),
                       Override(Debug().rt3d_group_rotations, kTileMaxRotations)}));
      const bool use_tiles = Tiles3DEnabled() && G <= 1024;
      const bool crosscheck = use_tiles && Debug().rt3d_crosscheck != 0;
      // (boxes of the rotated chunks from a pre-pass kernel instead of a reduction per chunk
      // inside the tile kernel; debug switch rt3d_no_boxes for the A/B)
      const bool use_boxes = Debug().rt3d_no_boxes == 0;
      Rt3DTileParams TG{};
      int max_chunks = 0;
      int* d_segment_counts = nullptr;
      // Staged second candidate round (see Rt3DStageFilterKernel); CMX_RT3D_STAGED=0 for the A/B.
      // The cross-check compares complete sums and the expand-all mode has no second round.
      const bool staged = use_tiles && !crosscheck && Debug().rt3d_unstaged == 0 &&
                          Debug().rt3d_expand_all == 0;
      // The rotation-block level above the group pass (Rt3DSelectPairsKernel): needs a window of
      // more than one rotation per axis, small enough for the 1.03 of its bound, and the lists of
      // the tiled passes.
      const bool rotblocks = use_tiles && !crosscheck && Debug().rt3d_no_rotblocks == 0 && A >= 1 &&
                             (A + 1) * static_cast<double>(step) * 1.7320508 <= 0.2;
      // (work lists of the tiled list passes: several rotations each, so that a tile serves some
      // hundred lanes even when a rotation keeps only a few dozen entries)
      const int list_rotations =
          use_tiles && !crosscheck ? std::max(1, std::min(8, Override(Debug().rt3d_cand_rotations, 8))) : 1;
      const int num_lists = DivUp(R, list_rotations);
      uint8_t* d_computed = nullptr;       // rotblocks: [R][G] pair computed | [R][G] pair flagged
      float* d_block_upper = nullptr;      // rotblocks: [Rb][G]
      unsigned* d_block_max = nullptr;
      float* d_list_boxes = nullptr;       // chunk boxes of the candidate chunk list, per work list
      std::function<void(const unsigned*, float)> run_pairs;   // one round of the sparse group pass
      if (use_tiles) {
        Rt3DBinParams BP{};
        BP.rotation = rot[R / 2];
        BP.translation = trans[T / 2];
        BP.inv_resolution = B.inv_resolution;
        BP.off_x = B.off_x; BP.off_y = B.off_y; BP.off_z = B.off_z;
        BP.t0x = B.t0x; BP.t0y = B.t0y; BP.t0z = B.t0z;
        BP.lo_x = B.lo_x; BP.hi_x = B.hi_x; BP.lo_y = B.lo_y; BP.hi_y = B.hi_y;
        BP.lo_z = B.lo_z; BP.hi_z = B.hi_z;
        BP.bins_x = DivUp(bx, kTileBin); BP.bins_y = DivUp(by, kTileBin);
        BP.bins_z = DivUp(bz, kTileBin);
        BP.n = n;
        const int num_bins = BP.bins_x * BP.bins_y * BP.bins_z;
        max_chunks = n / kTileChunkCandidates + num_bins + 1;      // (either list)
        float* d_sorted = ws->dev[16].ReserveAs<float>(3 * static_cast<size_t>(n));
        char* d_bins = static_cast<char*>(ws->dev[17].Reserve(
            sizeof(int) * (num_bins + 8) + 2 * sizeof(int2) * static_cast<size_t>(max_chunks)));
        int* d_bin_count = reinterpret_cast<int*>(d_bins);
        // [0] group list, [1] candidate list, [2..3] / [4..5] their segment boundaries
        int* d_chunk_count = d_bin_count + num_bins;
        int2* d_chunks = reinterpret_cast<int2*>(d_bin_count + num_bins + 8);
        int2* d_chunks_candidates = d_chunks + max_chunks;
        CMX_HIP(hipMemsetAsync(d_bin_count, 0, sizeof(int) * (num_bins + 8), ws->stream));
        Rt3DBinCountKernel<<<DivUp(n, 256), 256, 0, ws->stream>>>(BP, d_xyz, d_bin_count);
        // (segment windows: sixteen or more periods over a large cloud, never shorter than four
        // group-pass pieces)
        const int segment_window = n >= 32768 ? 4096 : 2048;
        // (debug switch rt3d_segments = a | b << 8: the first segment ends at a / 16, the second
        // at b / 16 of a window; default a quarter and a half.  Experiments.)
        int sixteenths0 = 4, sixteenths1 = 8;
        if (const int seg = Debug().rt3d_segments; seg > 0) {
          sixteenths0 = std::max(1, std::min(15, seg & 0xff));
          sixteenths1 = std::max(sixteenths0, std::min(15, (seg >> 8) & 0xff));
        }
        Rt3DBinScanKernel<<<1, 1024, 0, ws->stream>>>(
            d_bin_count, num_bins, d_chunks, d_chunks_candidates, d_chunk_count, segment_window,
            segment_window / 16 * sixteenths0, segment_window / 16 * sixteenths1);
        d_segment_counts = d_chunk_count;
        Rt3DBinScatterKernel<<<DivUp(n, 256), 256, 0, ws->stream>>>(BP, d_xyz, d_bin_count,
                                                                    d_sorted);
        CMX_HIP(hipGetLastError());
        if (Debug().rt3d_report) {
          TG.stats = reinterpret_cast<unsigned long long*>(ws->dev[19].Reserve(64));
          CMX_HIP(hipMemsetAsync(TG.stats, 0, 64, ws->stream));
        }
        TG.sorted_xyz = d_sorted;
        TG.chunks = d_chunks;
        TG.num_chunks = d_chunk_count;
        TG.pitch_x = static_cast<int>(bx); TG.pitch_y = static_cast<int>(by);
        TG.pitch_z = static_cast<int>(bz);
        trace.Mark("bins");
      }
      // Span of a translation table in cells (what widens a chunk's box into its tile).
      const auto span_of = [&](const std::vector<float4>& table, Rt3DTileParams* tp) {
        for (int a = 0; a < 3; ++a) { tp->tr_lo[a] = INFINITY; tp->tr_hi[a] = -INFINITY; }
        for (const float4& v : table) {
          const float c[3] = {v.x * B.inv_resolution, v.y * B.inv_resolution,
                              v.z * B.inv_resolution};
          for (int a = 0; a < 3; ++a) {
            tp->tr_lo[a] = std::min(tp->tr_lo[a], c[a]);
            tp->tr_hi[a] = std::max(tp->tr_hi[a], c[a]);
          }
        }
      };
      RecordEvent(ws->ev_k0, ws->stream);
      if (use_tiles) {
        Rt3DTileParams TP = TG;
        span_of(group, &TP);
        TP.rotations_per_block = rot_per_block;
        TP.tile_capacity = Override(Debug().rt3d_group_tile_kb, 44) * 1024;
        const int threads = std::min(1024, DivUp(rot_per_block * G, 64) * 64);
        TP.fixed_point = Debug().rt3d_group_float ? 0 : 1;
        if (crosscheck) TP.fixed_point = 0;       // (the float path reproduces the gather kernel's sums)
        const size_t lds = (sizeof(v2f) * (3 * kTileChunkGroups / 2 + 2) +
                            sizeof(uint32_t) * kTileChunkGroups) * rot_per_block +
                           TP.tile_capacity + 32;
        const int blocks = DivUp(R, rot_per_block);
        const int slices = std::max(1, std::min(max_chunks, DivUp(8 * cus, blocks)));
        BG.cell_count = static_cast<unsigned>(cells);
        BG.sums = reinterpret_cast<uint2*>(ws->dev[18].Reserve(sizeof(uint2) * RG * (staged ? 2 : 1)));
        CMX_HIP(hipMemsetAsync(BG.sums, 0, sizeof(uint2) * RG * (staged ? 2 : 1), ws->stream));
        if (staged) {
          TP.seg_bounds = d_segment_counts + 2;
          TP.seg_first = 0; TP.seg_last = 2;
          TP.seg_sums = BG.sums + RG;
        }
        OptInLds(reinterpret_cast<const void*>(Rt3DTileKernel<true, false>), ws->device, lds);
        if (use_boxes) {
          // (the chunk boxes of the list passes: the candidate chunk list under list_rotations
          // rotations -- the sparse group pass and the candidate passes share them)
          d_list_boxes = ws->dev[21].ReserveAs<float>(static_cast<size_t>(num_lists) * max_chunks * 6);
          Rt3DChunkBoxKernel<<<dim3(num_lists, 8), 256, 0, ws->stream>>>(
              BG, TG.sorted_xyz, TG.chunks + max_chunks, TG.num_chunks + 1, list_rotations, d_list_boxes);
        }
        if (rotblocks) {
          // ---- blocks of 2 x 2 x 2 rotations x groups: the same kernel over the centre rotations
          // and the twice-dilated brick --------------------------------------------------------
          uint8_t* d_dilated2 = ws->dev[26].ReserveAs<uint8_t>(cells);
          DilateXKernel<<<DivUp(cells, 256), 256, 0, ws->stream>>>(d_dilated, d_tmp,
                                                                   static_cast<int>(bx), cells);
          DilateYZKernel<<<DivUp(cells, 256), 256, 0, ws->stream>>>(
              d_tmp, d_dilated2, static_cast<int>(bx), static_cast<int>(by), static_cast<int>(bz));
          const long long RbG = Rb * G;
          char* d_block = static_cast<char*>(ws->dev[27].Reserve((sizeof(uint2) + sizeof(float)) * RbG + 16));
          uint2* d_block_sums = reinterpret_cast<uint2*>(d_block);
          d_block_upper = reinterpret_cast<float*>(d_block + sizeof(uint2) * RbG);
          d_block_max = reinterpret_cast<unsigned*>(d_block + (sizeof(uint2) + sizeof(float)) * RbG);
          CMX_HIP(hipMemsetAsync(d_block, 0, (sizeof(uint2) + sizeof(float)) * RbG + 16, ws->stream));
          d_computed = ws->dev[28].ReserveAs<uint8_t>(2 * RG);
          CMX_HIP(hipMemsetAsync(d_computed, 0, 2 * RG, ws->stream));
          Rt3DBulkParams BS = BG;
          BS.cells = d_dilated2;
          BS.num_rotations = static_cast<int>(Rb);
          BS.rotation = d_rot_b; BS.rotation_angle = d_angle_b;
          BS.upper = d_block_upper; BS.sums = d_block_sums; BS.max_upper_bits = d_block_max;
          Rt3DTileParams TB = TP;
          TB.seg_bounds = nullptr; TB.seg_sums = nullptr; TB.seg_first = 0; TB.seg_last = 2;
          const int blocks_b = DivUp(Rb, rot_per_block);
          const int slices_b = std::max(1, std::min(max_chunks, DivUp(8 * cus, blocks_b)));
          if (use_boxes) {
            float* d_boxes = ws->dev[20].ReserveAs<float>(static_cast<size_t>(blocks_b) * max_chunks * 6);
            Rt3DChunkBoxKernel<<<dim3(blocks_b, 8), 256, 0, ws->stream>>>(
                BS, TG.sorted_xyz, TG.chunks, TG.num_chunks, rot_per_block, d_boxes);
            TB.boxes = d_boxes;
          }
          Rt3DTileKernel<true, false><<<dim3(blocks_b, slices_b), threads, lds, ws->stream>>>(BS, TB);
          Rt3DSumBoundsKernel<<<DivUp(RbG, 256), 256, 0, ws->stream>>>(BS);
          trace.Mark("rotation blocks");
          // ---- the group pass proper over the pairs a threshold cannot exclude -----------------
          const int pair_items = 512;
          Rt3DBulkParams BQ = BG;                // (sums, upper, group table, once-dilated brick)
          BQ.counts = d_counts; BQ.items = d_items; BQ.blocks = d_blocks;
          BQ.list_rotations = list_rotations; BQ.block_items = pair_items;
          Rt3DTileParams TQ = TP;
          TQ.chunks = TG.chunks + max_chunks;    // the candidate pass's chunk list (256 points)
          TQ.num_chunks = TG.num_chunks + 1;
          TQ.block_items = pair_items;
          TQ.boxes = d_list_boxes;
          if (staged) TQ.seg_bounds = d_segment_counts + 4;      // (seg_first .. seg_last, seg_sums: TP's)
          const size_t lds_q = (sizeof(v2f) * (3 * kTileChunkCandidates / 2 + 2) +
                                sizeof(uint32_t) * kTileChunkCandidates) * list_rotations +
                               TQ.tile_capacity + 32;
          OptInLds(reinterpret_cast<const void*>(Rt3DTileKernel<true, true>), ws->device, lds_q);
          int* d_pair_total = d_num_blocks + 3;
          uint8_t* d_pair_flags = d_computed + RG;
          const bool check_blocks = verify;
          run_pairs = [=, &ws](const unsigned* threshold_bits, float factor) {
            CMX_HIP(hipMemsetAsync(d_num_blocks, 0, sizeof(int), ws->stream));
            Rt3DSelectPairsKernel<<<DivUp(RG, 256), 256, 0, ws->stream>>>(
                d_block_upper, G, static_cast<int>(R), side_r, Ab, threshold_bits, factor, d_computed,
                d_pair_flags);
            Rt3DCompactKernel<<<num_lists, 256, 0, ws->stream>>>(
                d_pair_flags, G, static_cast<int>(R), list_rotations, d_counts, d_items, d_pair_total,
                d_blocks, d_num_blocks, pair_items);
            CMX_HIP(hipMemcpyAsync(h_num_blocks, d_num_blocks, sizeof(int), hipMemcpyDeviceToHost,
                                   ws->stream));
            CMX_HIP(hipMemcpyAsync(h_num_blocks + 4, d_segment_counts, sizeof(int) * 8,
                                   hipMemcpyDeviceToHost, ws->stream));
            CMX_HIP(hipStreamSynchronize(ws->stream));
            const int nbq = *h_num_blocks;
            const int list_chunks = h_num_blocks[4 + 1];         // chunks of the candidate list
            if (nbq > 0 && list_chunks > 0) {
              const int slices_q = std::max(1, std::min(list_chunks, DivUp(8 * cus, nbq)));
              Rt3DTileKernel<true, true><<<dim3(nbq, slices_q), pair_items, lds_q, ws->stream>>>(BQ, TQ);
            }
            Rt3DSumBoundsMaskedKernel<<<DivUp(RG, 256), 256, 0, ws->stream>>>(BG, d_computed);
            if (check_blocks)
              Rt3DVerifyBlocksKernel<<<DivUp(RG, 256), 256, 0, ws->stream>>>(
                  d_block_upper, d_group_upper, d_computed, G, static_cast<int>(R), side_r, Ab,
                  d_violations);
          };
          // first round: the pairs of the blocks next to the best block bound (all of them when
          // every group is to be expanded: rt3d_expand_all)
          const int permille = Debug().rt3d_rotblock_permille;
          run_pairs(d_block_max, Debug().rt3d_expand_all ? 0.f : permille > 0 ? std::min(permille, 1000) * 1e-3f : 0.97f);
        } else {
          if (use_boxes) {
            float* d_boxes = ws->dev[20].ReserveAs<float>(static_cast<size_t>(blocks) * max_chunks * 6);
            Rt3DChunkBoxKernel<<<dim3(blocks, 8), 256, 0, ws->stream>>>(
                BG, TG.sorted_xyz, TG.chunks, TG.num_chunks, rot_per_block, d_boxes);
            TP.boxes = d_boxes;
          }
          Rt3DTileKernel<true, false><<<dim3(blocks, slices), threads, lds, ws->stream>>>(BG, TP);
          Rt3DSumBoundsKernel<<<DivUp(RG, 256), 256, 0, ws->stream>>>(BG);
        }
        if (TG.stats) {
          unsigned long long h[4];
          int chunks_made = 0;
          CMX_HIP(hipMemcpyAsync(h, TG.stats, sizeof(h), hipMemcpyDeviceToHost, ws->stream));
          CMX_HIP(hipMemcpyAsync(&chunks_made, TG.num_chunks, sizeof(int), hipMemcpyDeviceToHost,
                                 ws->stream));
          CMX_HIP(hipMemsetAsync(TG.stats, 0, 64, ws->stream));
          CMX_HIP(hipStreamSynchronize(ws->stream));
          const double units = static_cast<double>(h[0] + h[1]);
          fprintf(stderr,
                  "[cmx] rt3d tiles (groups): %d chunks, %d rotations x %d lanes per block, %d "
                  "slices; %.0f (block, chunk) units: %.1f %% in LDS, mean tile %.1f KB, mean "
                  "chunk %.1f points\n",
                  chunks_made, rot_per_block, threads, slices, units, 100. * h[0] / std::max(units, 1.),
                  h[2] / std::max(units, 1.) / 1024., h[3] / std::max(units, 1.));
        }
        if (crosscheck) {
          Rt3DBulkParams B2 = BG;
          float* d_upper2 = ws->dev[19].ReserveAs<float>(RG + 4);
          B2.upper = d_upper2;
          B2.max_upper_bits = reinterpret_cast<unsigned*>(d_upper2 + RG);
          int* d_mismatch = reinterpret_cast<int*>(d_upper2 + RG + 1);
          CMX_HIP(hipMemsetAsync(d_upper2 + RG, 0, 16, ws->stream));
          const int flat_threads = std::min(kBulk3DThreads, (G + 1) / 64 * 64);
          if (flat_threads >= 128 && R >= 2)
            Rt3DBulkKernel<true><<<DivUp(RG, flat_threads), flat_threads, 0, ws->stream>>>(B2, d_xyz);
          else
            Rt3DBulkKernel<true><<<dim3(DivUp(G, kBulk3DThreads), std::max<unsigned>(2u, R)),
                                   kBulk3DThreads, 0, ws->stream>>>(B2, d_xyz);
          Rt3DCompareFloatsKernel<<<DivUp(RG, 256), 256, 0, ws->stream>>>(d_group_upper, d_upper2,
                                                                         RG, d_mismatch);
          int mismatches = -1;
          CMX_HIP(hipMemcpyAsync(&mismatches, d_mismatch, sizeof(int), hipMemcpyDeviceToHost,
                                 ws->stream));
          CMX_HIP(hipStreamSynchronize(ws->stream));
          CMX_REQUIRE(mismatches == 0,
                      "internal error: %d of %lld tiled group bounds differ from the gather kernel's",
                      mismatches, RG);
        }
      } else {
      // Flat (rotation, group) lanes: as many whole wavefronts per block as fit G + 1 lanes.
      const int flat_threads = std::min(kBulk3DThreads, (G + 1) / 64 * 64);
      if (flat_threads >= 128 && R >= 2)
        Rt3DBulkKernel<true><<<DivUp(RG, flat_threads), flat_threads, 0, ws->stream>>>(BG, d_xyz);
      else
        Rt3DBulkKernel<true><<<dim3(DivUp(G, kBulk3DThreads), std::max<unsigned>(2u, R)),
                               kBulk3DThreads, 0, ws->stream>>>(BG, d_xyz);
      }
      RecordEvent(ws->ev_k1, ws->stream);
      // Candidate pass, twice: the members of the groups next to the best upper bound yield a
      // lower bound; then everything that lower bound cannot exclude.  (The threshold only
      // rises afterwards, so no third round can add a group.)
      Rt3DBulkParams BC = B;
      BC.cells = d_tiled;
      BC.cell_count = static_cast<unsigned>(tiled_cells);
      BC.tiles_y = tiles_y; BC.pitch_z = pitch_z;
      BC.translation = d_trans; BC.num_translations = static_cast<int>(T);
      BC.counts = d_counts; BC.items = d_items; BC.blocks = d_blocks;
      BC.upper = d_upper;
      BC.sums = reinterpret_cast<uint2*>(d_tmp);
      CMX_HIP(hipMemsetAsync(d_tmp, 0, sizeof(uint2) * static_cast<size_t>(num_candidates),
                             ws->stream));
      trace.Mark("group pass");
      Rt3DTileParams TC = TG;
      const int block_items =
          !use_tiles ? kCand3DThreads : crosscheck ? 256 : Override(Debug().rt3d_cand_threads, 512);
      BC.block_items = block_items;
      BC.list_rotations = list_rotations;
      if (use_tiles) {
        span_of(trans, &TC);
        TC.chunks = TG.chunks + max_chunks;                        // the candidate pass's own list
        TC.num_chunks = TG.num_chunks + 1;
        TC.rotations_per_block = 1;
        TC.boxes = d_list_boxes;                        // (computed before the group pass)
        TC.tile_capacity = Override(Debug().rt3d_cand_tile_kb, 48) * 1024;
        TC.block_items = block_items;
        BC.cells = d_bulk;                              // the row-major q brick
        BC.cell_count = static_cast<unsigned>(cells);
      }
      int round_blocks[2] = {0, 0};
      int stage_blocks[2] = {-1, -1};            // work blocks left after segments 0 and 1
      int* h_segments = h_num_blocks + 4;        // pinned: the chunk counts by segment
      int* d_stage_total = d_stage_scratch;      // (re-compactions must not count items twice)
      // Debug switch rt3d_expand_all (tests, with rt3d_verify): every group is expanded in the
      // first round, so every group bound is checked against every one of its members.
      const float first_round_factor = Debug().rt3d_expand_all ? 0.f : 0.97f;
      for (int round = 0; round < 2; ++round) {
        CMX_HIP(hipMemsetAsync(d_num_blocks, 0, sizeof(int), ws->stream));
        if (round == 1 && rotblocks) {
          // the pairs of every rotation block the lower bound of round 0 cannot exclude
          RecordEvent(ws->ev_x0, ws->stream);
          run_pairs(d_max_lower, 1.f);
          RecordEvent(ws->ev_x1, ws->stream);
          second_pairs_pass = true;
          CMX_HIP(hipMemsetAsync(d_num_blocks, 0, sizeof(int), ws->stream));
          trace.Mark("group pass 2 (pairs)");
        }
        Rt3DSelectGroupsKernel<<<DivUp(RG, 256), 256, 0, ws->stream>>>(
            d_group_upper, G, static_cast<int>(R), side_t, gpa,
            round == 0 ? d_max_upper : d_max_lower, round == 0 ? first_round_factor : 1.f, d_expanded,
            d_flags, static_cast<int>(T));
        Rt3DCompactKernel<<<num_lists, 256, 0, ws->stream>>>(
            d_flags, static_cast<int>(T), static_cast<int>(R), list_rotations, d_counts, d_items,
            d_total, d_blocks, d_num_blocks, block_items);
        // The launch is sized by what survived: one sync per round (tens of microseconds)
        // instead of hundreds of thousands of blocks that find nothing to do.
        CMX_HIP(hipMemcpyAsync(h_num_blocks, d_num_blocks, sizeof(int), hipMemcpyDeviceToHost,
                               ws->stream));
        if (staged && round == 0)
          CMX_HIP(hipMemcpyAsync(h_segments, d_segment_counts, sizeof(int) * 8,
                                 hipMemcpyDeviceToHost, ws->stream));
        CMX_HIP(hipStreamSynchronize(ws->stream));
        int nb = round_blocks[round] = *h_num_blocks;
        if (nb == 0) continue;
        if (use_tiles) {
          const size_t lds = sizeof(v2f) * (3 * kTileChunkCandidates / 2 + 2) * list_rotations +
                             TC.tile_capacity + 32;
          OptInLds(reinterpret_cast<const void*>(Rt3DTileKernel<false, true>), ws->device, lds);
          if (staged && round == 1) {
            // Segment by segment; after the first and the second, the candidates whose bound
            // (own sum so far + the group's sum over the rest) has fallen below the best lower
            // bound leave the work lists.  Verification mode keeps the lists and records the
            // decisions instead (Rt3DStageCheckKernel).
            unsigned long long* d_stage_ub = nullptr;
            uint8_t* d_dropped = nullptr;
            if (verify) {
              d_stage_ub = reinterpret_cast<unsigned long long*>(
                  ws->dev[22].Reserve(sizeof(unsigned long long) * static_cast<size_t>(num_candidates)));
              d_dropped = ws->dev[23].ReserveAs<uint8_t>(static_cast<size_t>(num_candidates));
            }
            int live = nb;
            for (int stage = 0; stage < 3 && live > 0; ++stage) {
              Rt3DTileParams TS = TC;
              TS.seg_bounds = d_segment_counts + 4;
              TS.seg_first = TS.seg_last = stage;
              const int seg_chunks =
                  stage == 0 ? h_segments[4] : stage == 1 ? h_segments[5] - h_segments[4]
                                                          : h_segments[1] - h_segments[5];
              if (seg_chunks > 0) {
                const int slices = std::max(1, std::min(seg_chunks, DivUp(8 * cus, live)));
                Rt3DTileKernel<false, true><<<dim3(live, slices), block_items, lds, ws->stream>>>(BC, TS);
              }
              if (stage == 2) break;
              Rt3DStageFilterKernel<<<live, block_items, 0, ws->stream>>>(
                  BC, BG.sums, BG.sums + RG, G, side_t, gpa, stage, d_flags, d_stage_ub, d_dropped);
              if (verify) continue;
              CMX_HIP(hipMemsetAsync(d_num_blocks, 0, sizeof(int), ws->stream));
              Rt3DCompactKernel<<<num_lists, 256, 0, ws->stream>>>(
                  d_flags, static_cast<int>(T), static_cast<int>(R), list_rotations, d_counts,
                  d_items, d_stage_total, d_blocks, d_num_blocks, block_items);
              CMX_HIP(hipMemcpyAsync(h_num_blocks, d_num_blocks, sizeof(int), hipMemcpyDeviceToHost,
                                     ws->stream));
              CMX_HIP(hipStreamSynchronize(ws->stream));
              live = *h_num_blocks;
              stage_blocks[stage] = live;
            }
            nb = live;
            if (nb > 0) {
              Rt3DBoundsKernel<<<nb, block_items, 0, ws->stream>>>(BC);
              if (verify) {
                Rt3DStageCheckKernel<<<nb, block_items, 0, ws->stream>>>(BC, d_stage_ub, d_dropped,
                                                                          d_violations);
                Rt3DVerifyKernel<<<nb, block_items, 0, ws->stream>>>(BC, d_group_upper, G, side_t,
                                                                        gpa, d_violations);
              }
            }
            trace.Mark("candidate pass 2 (staged)");
            continue;
          }
          const int slices = std::max(1, std::min(max_chunks, DivUp(8 * cus, nb)));
          Rt3DTileKernel<false, true><<<dim3(nb, slices), block_items, lds, ws->stream>>>(BC, TC);
          if (crosscheck) {
            // the same work lists on the gather kernel, into a second (Q, A) array
            Rt3DBulkParams B2 = BC;
            uint2* d_sums2 = reinterpret_cast<uint2*>(
                ws->dev[19].Reserve(sizeof(uint2) * static_cast<size_t>(num_candidates) + 16));
            int* d_mismatch = reinterpret_cast<int*>(d_sums2 + num_candidates);
            B2.sums = d_sums2;
            B2.cells = d_tiled;
            B2.cell_count = static_cast<unsigned>(tiled_cells);
            if (round == 0)
              CMX_HIP(hipMemsetAsync(d_sums2, 0,
                                     sizeof(uint2) * static_cast<size_t>(num_candidates) + 16,
                                     ws->stream));
            B2.slice_points = DivUp(n, kBulk3DChunk) * kBulk3DChunk;
            Rt3DBulkKernel<false><<<dim3(nb, 1, 1), block_items, 0, ws->stream>>>(B2, d_xyz);
            Rt3DCompareSumsKernel<<<DivUp(num_candidates, 256), 256, 0, ws->stream>>>(
                BC.sums, d_sums2, num_candidates, d_mismatch);
            int mismatches = -1;
            CMX_HIP(hipMemcpyAsync(&mismatches, d_mismatch, sizeof(int), hipMemcpyDeviceToHost,
                                   ws->stream));
            CMX_HIP(hipStreamSynchronize(ws->stream));
            CMX_REQUIRE(mismatches == 0,
                        "internal error: %d tiled candidate sums differ from the gather kernel's",
                        mismatches);
          }
        } else {
        // Points are split over blockIdx.z until the launch is ~32 blocks per CU.
        // (blocks of two wavefronts: sixteen fit a CU)
        const int want = std::max(1, DivUp(32 * cus, nb));
        BC.slice_points = std::max(1, DivUp(DivUp(n, want), kBulk3DChunk)) * kBulk3DChunk;
        const dim3 cand_grid(nb, 1, DivUp(n, BC.slice_points));
        Rt3DBulkKernel<false><<<cand_grid, kCand3DThreads, 0, ws->stream>>>(BC, d_xyz);
        }
        Rt3DBoundsKernel<<<nb, block_items, 0, ws->stream>>>(BC);
        if (verify)
          Rt3DVerifyKernel<<<nb, block_items, 0, ws->stream>>>(BC, d_group_upper, G, side_t,
                                                                  gpa, d_violations);
        trace.Mark(round == 0 ? "candidate pass 1" : "candidate pass 2");
      }
      Rt3DBulkCollectKernel<<<DivUp(num_candidates, 256), 256, 0, ws->stream>>>(
          d_upper, num_candidates, d_max_lower, d_bcount, d_bfinalists, kBulkFinalistCap);
      Rt3DExactKernel<<<kBulkFinalistCap, 256, 0, ws->stream>>>(P, d_xyz, n, d_bfinalists,
                                                                d_bcount, kBulkFinalistCap,
                                                                d_exact);
      trace.Mark("finalists");
      CMX_HIP(hipGetLastError());
      RecordEvent(ws->ev_end, ws->stream);
      CMX_HIP(hipMemcpyAsync(h_bmisc, d_bmisc, head_bytes, hipMemcpyDeviceToHost, ws->stream));
      CMX_HIP(hipMemcpyAsync(h_num_blocks + 12, d_num_blocks + 3, sizeof(int), hipMemcpyDeviceToHost,
                             ws->stream));          // (rotation blocks: pairs that went through the group pass)
      CMX_HIP(hipStreamSynchronize(ws->stream));
      trace.Report();
      const int count = *reinterpret_cast<int*>(h_bmisc + 4);
      const int total_items = *reinterpret_cast<int*>(h_bmisc + 12);
      if (verify) {
        int violations = 0;
        CMX_HIP(hipMemcpy(&violations, d_violations, sizeof(int), hipMemcpyDeviceToHost));
        CMX_REQUIRE(violations == 0,
                    "internal error: %d bound violations (a group bound below a member's lower "
                    "bound, or a staged bound below the candidate's final sum)", violations);
      }
      if (Debug().rt3d_report) {
        float ms = 0.f, all = 0.f;
        ms = ElapsedMs(ws->ev_k0, ws->ev_k1);
        all = ElapsedMs(ws->ev_begin, ws->ev_end);
        float lo_f, up_f;
        std::memcpy(&lo_f, h_bmisc, 4);
        std::memcpy(&up_f, h_bmisc + 8, 4);
        fprintf(stderr,
                "[cmx] rt3d: group pass %.3f ms (%lld bounds), %d of %lld candidates scored, "
                "%d finalists, best lower %.6f, best group upper %.6f, work blocks %d + %d "
                "(staged: %d after a quarter of the points, %d after half), device %.3f ms\n",
                ms, rotblocks ? Rb * G + h_num_blocks[12] : RG, total_items, num_candidates, count, lo_f, up_f, round_blocks[0],
                round_blocks[1], stage_blocks[0], stage_blocks[1], all);
      }
      if (count >= 1 && count <= kBulkFinalistCap) {
        const int* fin = reinterpret_cast<const int*>(h_bmisc + 16);
        const float* exact =
            reinterpret_cast<const float*>(h_bmisc + 16 + sizeof(int) * kBulkFinalistCap);
        std::vector<std::pair<long long, float>> pairs(count);
        for (int i = 0; i < count; ++i) {
          const long long r = fin[i] / T, t = fin[i] % T;
          pairs[i] = {t * R + r, exact[i]};
        }
        std::sort(pairs.begin(), pairs.end());
        finalists.resize(count);
        acc.resize(count);
        for (int i = 0; i < count; ++i) { finalists[i] = pairs[i].first; acc[i] = pairs[i].second; }
        num_finalists = count;
        group_bounds = rotblocks ? Rb * G + h_num_blocks[12] : RG;
        bounds_evaluated = group_bounds + total_items;
        done = true;
      }
      // else: a flat score landscape -- every candidate on the per-candidate kernel below.
    }

    if (!done) {
    RecordEvent(ws->ev_k0, ws->stream);
    Rt3DScoreKernel<<<dim3(DivUp(T, 64), static_cast<unsigned>(R)), 64, 0, ws->stream>>>(
        P, d_xyz, n, d_unweighted, d_weighted, d_max);
    RecordEvent(ws->ev_k1, ws->stream);
    Rt3DCollectKernel<<<DivUp(num_candidates, 256), 256, 0, ws->stream>>>(
        d_weighted, num_candidates, d_max, d_count, d_finalists, kFinalistCap);
    CMX_HIP(hipGetLastError());
    RecordEvent(ws->ev_end, ws->stream);
    CMX_HIP(hipMemcpyAsync(h_misc, d_misc, 16 + sizeof(long long) * kFinalistCap,
                           hipMemcpyDeviceToHost, ws->stream));
    CMX_HIP(hipStreamSynchronize(ws->stream));

    const int count = *reinterpret_cast<int*>(h_misc + 4);
    const long long* h_finalists = reinterpret_cast<long long*>(h_misc + 16);
    if (count <= kFinalistCap) {
      finalists.assign(h_finalists, h_finalists + count);
      std::sort(finalists.begin(), finalists.end());
      acc.resize(count);
      for (int i = 0; i < count; ++i)
        CMX_HIP(hipMemcpy(&acc[i], d_unweighted + finalists[i], sizeof(float),
                          hipMemcpyDeviceToHost));
    } else {
      finalists.resize(num_candidates);
      for (long long c = 0; c < num_candidates; ++c) finalists[c] = c;
      acc.resize(num_candidates);
      CMX_HIP(hipMemcpy(acc.data(), d_unweighted, sizeof(float) * num_candidates,
                        hipMemcpyDeviceToHost));
    }
    }
    CMX_REQUIRE(!finalists.empty(), "internal error: no candidate collected");
    // Exact weighting and the strict '>' running maximum of :44-50.
    float best_score = -1.f;
    long long best = -1;
    for (size_t i = 0; i < finalists.size(); ++i) {
      const long long c = finalists[i];
      const long long t = c / R, r = c % R;
      float sc = acc[i];
      const double penalty = static_cast<double>(trans[t].w) * P.wt +
                             static_cast<double>(angle[r]) * P.wr;
      sc *= std::exp(-(penalty * (penalty * 1.)));
      if (sc > best_score) { best_score = sc; best = c; }
    }
    {
      const long long t = best / R, r = best % R;
      h3::Rigid candidate;
      candidate.t = {trans[t].x, trans[t].y, trans[t].z};
      candidate.q = {rot[r].w, rot[r].x, rot[r].y, rot[r].z};
      *pose_estimate = h3::ToPose(candidate);
      *score = best_score;
    }
    if (stats) {
      cmx_match_stats st{};
      // Bulk path: the search space is covered by bounds -- R * G group bounds plus the
      // candidates scored one by one; `candidates_scored` stays the size of the search space
      // (what the reference scores), `coarse_candidates` says how many bounds that took.
      st.candidates_scored = num_candidates;
      st.coarse_candidates = bounds_evaluated ? bounds_evaluated : num_candidates;
      st.num_scans = static_cast<int>(R);
      st.nodes_expanded = num_finalists;   // bulk path: candidates re-scored exactly
      float ms = 0.f;
      ms = ElapsedMs(ws->ev_begin, ws->ev_end);
      st.device_ms = ms;
      ms = ElapsedMs(ws->ev_k0, ws->ev_k1);
      st.dominant_kernel_ms = ms;
      // the bounds above the candidates (rotation blocks, then the (rotation, group) pairs they
      // leave): how many, their lookups, and the time of ALL their passes
      st.expansion_nodes = group_bounds;
      st.expansion_lookups = group_bounds * n;
      st.expansion_ms = ms;
      if (second_pairs_pass) {
        ms = ElapsedMs(ws->ev_x0, ws->ev_x1);
        st.expansion_ms += ms;
      }
      *stats = st;
    }
  });
}
}  // namespace
}  // namespace cmx

extern "C" cmx_status cmx_rt3d_match(const cmx_rt_options* options, float grid_resolution,
                                     const cmx_voxel* voxels, int64_t num_voxels,
                                     const cmx_pose3d* initial_pose_estimate,
                                     const float* point_cloud_xyz, int32_t num_points,
                                     int32_t device, float* score, cmx_pose3d* pose_estimate,
                                     cmx_match_stats* stats) {
  return cmx::Rt3DMatchImpl(options, grid_resolution, voxels, num_voxels, nullptr,
                            initial_pose_estimate, point_cloud_xyz, num_points, device, score,
                            pose_estimate, stats);
}

// The same search on the ACTIVE submap's HybridGrid where cmx_grid3d keeps it in HBM
// (LocalTrajectoryBuilder3D's per-scan Match -> InsertRangeData pair, mapping/internal/3d/
// local_trajectory_builder_3d.cc:96-108): only the scan crosses PCIe.
extern "C" cmx_status cmx_rt3d_match_grid(const cmx_rt_options* options, const cmx_grid3d* grid,
                                          const cmx_pose3d* initial_pose_estimate,
                                          const float* point_cloud_xyz, int32_t num_points,
                                          float* score, cmx_pose3d* pose_estimate,
                                          cmx_match_stats* stats) {
  using namespace cmx;
  if (grid == nullptr) {
    return Guard([] { CMX_REQUIRE(false, "null argument"); });
  }
  Brick brick{};
  float resolution = 0.f;
  int device = 0;
  if (!Grid3DBrick(grid, &brick, &resolution, &device)) {
    // nothing inserted yet: an empty voxel list (every candidate scores kMinProbability)
    return Rt3DMatchImpl(options, resolution, nullptr, 0, nullptr, initial_pose_estimate,
                         point_cloud_xyz, num_points, device, score, pose_estimate, stats);
  }
  return Rt3DMatchImpl(options, resolution, nullptr, 0, &brick, initial_pose_estimate,
                       point_cloud_xyz, num_points, device, score, pose_estimate, stats);
}

