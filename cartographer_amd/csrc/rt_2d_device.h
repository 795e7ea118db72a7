// Device helpers shared by the two translation units of the real-time 2D matcher: rt_2d.hip
// (entry points, one-thread-per-candidate kernels for TSDFs and flat score landscapes) and
// rt_2d_tiles.hip (probability grids: integer bulk pass out of LDS tiles + exact finalists).
#ifndef CMX_RT_2D_DEVICE_H_
#define CMX_RT_2D_DEVICE_H_

#include "cmx_device.h"

namespace cmx {

// Integer bulk pass (rt_2d_tiles.hip): cells enter LDS as q = u >> kQShift, u = 32767 - value.
constexpr int kQShift = 5;
constexpr int kQ8Shift = 7;                  // the byte image of the bound kernel's tail: q8 = u >> 7 <= 255
constexpr int kRt2DMaxPoints = 8192;        // points per scan the tile path takes
// Finalists of a match: (candidate index, f32 score bits) pairs -- the first kFinalistHead next
// to the counters (they travel back with them), the rest in the overflow region.
constexpr int kFinalistCap = 4096;
constexpr int kFinalistHead = 61;           // 2 + 2 * 61 words, then two bound and two stage counters: 512 bytes per match
constexpr int kMaxRowsPerLane = 8;
#ifndef CMX_RT2D_TASK_ITERS
#define CMX_RT2D_TASK_ITERS 64
#endif
constexpr int kPairTaskIters = CMX_RT2D_TASK_ITERS;          // iterations (entries per stream) of one task

// Slack of a score bound for the rounding of the reference's N-term f32 chain: the chain's
// result differs from the real sum of the N probabilities (each <= 0.9, each rounded by the
// table to < 1e-7) by at most (N - 1) 2^-24 relative in first order; twice that plus 1e-4.
__host__ __device__ inline float Rt2DBoundSlack(int n) {
  return 1e-4f + 1.2e-7f * static_cast<float>(n);
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

// The reference's discretisation of one point: pre-rotation by the initial yaw (q0), rotation of
// the scan (qs), translation, MapLimits::GetCellIndex -- two yaw rotations without the exactly
// zero terms (RotateZ, cmx_device.h: bit-identical x / y for finite coordinates), the cell index
// from an f32 estimate when provably equal (CellIndexFast).  The result is clamped to one cell
// further outside the grid than any offset of the window can reach back in.
struct Rt2DFrame {
  double res, inv_res, max_x, max_y;
  float tx, ty, q0w, q0z;
  int nx, ny, nl;
};
__device__ __forceinline__ void Rt2DCellOf(const Rt2DFrame& F, float qsw, float qsz, float px,
                                           float py, int* ix, int* iy) {
  float ax, ay, bx, by;
  RotateZ(F.q0w, F.q0z, px, py, &ax, &ay);
  RotateZ(qsw, qsz, ax, ay, &bx, &by);
  const float x = bx + F.tx;
  const float y = by + F.ty;
  const int cx = CellIndexFast(F.max_y, y, F.res, F.inv_res);
  const int cy = CellIndexFast(F.max_x, x, F.res, F.inv_res);
  *ix = min(max(cx, -(F.nl + 1)), F.nx + F.nl);
  *iy = min(max(cy, -(F.nl + 1)), F.ny + F.nl);
}

// The same from a point already rotated by the initial yaw (RotateZ(F.q0w, F.q0z, px, py): the
// intermediate of Rt2DCellOf, so the cell is the same bit for bit).
__device__ __forceinline__ void Rt2DCellOfPrerotated(const Rt2DFrame& F, float qsw, float qsz,
                                                     float ax, float ay, int* ix, int* iy) {
  float bx, by;
  RotateZ(qsw, qsz, ax, ay, &bx, &by);
  const float x = bx + F.tx;
  const float y = by + F.ty;
  const int cx = CellIndexFast(F.max_y, y, F.res, F.inv_res);
  const int cy = CellIndexFast(F.max_x, x, F.res, F.inv_res);
  *ix = min(max(cx, -(F.nl + 1)), F.nx + F.nl);
  *iy = min(max(cy, -(F.nl + 1)), F.ny + F.nl);
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

// ---------------------------------------------------------------------------------------------
// The window update of the bulk pass.  A HALF-wavefront is one stream of points of one phase
// (window start mod 4): lane = (row r < H, block b < B), H B <= 32, and a lane owns the window
// rows r, r + H, ... (RPL of them) of the aligned 4-cell block b.  `addrs` holds, in every row of
// 16 lanes, the block addresses of 16 consecutive entries of the stream: `v_add_u32_dpp ...
// row_newbcast:k` adds lane k of the row to the lane's own (row, block) offset in ONE
// instruction, and the further rows of the lane ride in the read's immediate offset.
// ---------------------------------------------------------------------------------------------
typedef unsigned uint2v __attribute__((ext_vector_type(2)));

template <int K>
__device__ __forceinline__ int RowBcastAdd(int addrs, int lane_off) {
  int out;   // (asm: the compiler splits the intrinsic form into v_mov_b32_dpp + v_add_u32)
  asm("v_add_u32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
      : "=v"(out) : "v"(addrs), "v"(lane_off), "i"(K));
  return out;
}
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
// Two entries per instruction: 64 entries never carry out of a 16-bit field (64 * 1023 < 65536),
// so the packed sums are plain 32-bit additions and v_add3_u32 adds two entries' cells at once.
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
    static_assert(RPL <= 4, "compile-time row strides are instantiated up to four rows per lane");
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
// `null_addr` = byte address (relative to the image) of an all-zero block of RPL x H rows: what
// the slots of a stream beyond its length read.  acc32[j][c]: sum of cell c of the lane's block
// in its j-th row.
template <int RPL, int kRowStride>
__device__ __forceinline__ void RowPairAccumulate(const uint16_t* my_list, int my_len, int iters,
                                                  int lane, int lane_off, int row_stride_rt,
                                                  int null_addr, uint32_t (&acc32)[RPL][4]) {
  uint32_t lo[RPL], hi[RPL];
#pragma unroll
  for (int j = 0; j < RPL; ++j) lo[j] = hi[j] = 0;
  const int groups = iters >> 4;
  constexpr int kBank = 2 * RPL;             // reads of one bank
  // (lgkmcnt is a 4-bit counter: with eight rows per lane "at most 15 pending" stands in for 16 --
  // stricter, still exactly the older bank)
  constexpr int kWait = kBank < 15 ? kBank : 15;
  int e = lane & 15;
  // (unconditional read of a slot inside the padded list, selected afterwards: a branch around
  // the read makes the compiler wait for it -- and for everything else -- at once)
  const int raw0 = my_list[min(e, iters - 1)];
  int addrs = e < my_len ? raw0 << 3 : null_addr;
  for (int g = 0; g < groups; ++g) {
    // the next group's entries are fetched under this group's reads
    const int e_next = e + 16;
    const int raw_next = my_list[min(e_next, iters - 1)];
    const int addrs_next = e_next < my_len ? raw_next << 3 : null_addr;
    uint2v a[2][RPL], b[2][RPL];
    PairLoad<RPL, kRowStride, 0>(a, addrs, lane_off, row_stride_rt);
    PairLoad<RPL, kRowStride, 2>(b, addrs, lane_off, row_stride_rt);
    WaitLds<kWait>(); PairAdd<RPL>(a, lo, hi); PairLoad<RPL, kRowStride, 4>(a, addrs, lane_off, row_stride_rt);
    WaitLds<kWait>(); PairAdd<RPL>(b, lo, hi); PairLoad<RPL, kRowStride, 6>(b, addrs, lane_off, row_stride_rt);
    WaitLds<kWait>(); PairAdd<RPL>(a, lo, hi); PairLoad<RPL, kRowStride, 8>(a, addrs, lane_off, row_stride_rt);
    WaitLds<kWait>(); PairAdd<RPL>(b, lo, hi); PairLoad<RPL, kRowStride, 10>(b, addrs, lane_off, row_stride_rt);
    WaitLds<kWait>(); PairAdd<RPL>(a, lo, hi); PairLoad<RPL, kRowStride, 12>(a, addrs, lane_off, row_stride_rt);
    WaitLds<kWait>(); PairAdd<RPL>(b, lo, hi); PairLoad<RPL, kRowStride, 14>(b, addrs, lane_off, row_stride_rt);
    WaitLds<kWait>(); PairAdd<RPL>(a, lo, hi);
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

}  // namespace cmx

#endif  // CMX_RT_2D_DEVICE_H_
