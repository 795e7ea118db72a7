// Device-side arithmetic shared by the scan-matching kernels.
//
// Everything here is compiled with -ffp-contract=off: the reference computes
// these expressions without FMA contraction, and a one-ulp difference ahead of
// an lround moves a point into a neighbouring cell.  IEEE +,-,*,/ on gfx950
// are correctly rounded (f32 denormals enabled by default), so with identical
// operation order the results are bit-identical to the x86 reference.
// Transcendentals (cosf/sinf/exp) are NOT evaluated on the device where a
// rounding difference could change a result; the host passes them in.
#ifndef CMX_DEVICE_H_
#define CMX_DEVICE_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cmx {

constexpr int kMaxDepth = 12;   // branch_and_bound_depth upper bound
constexpr int kWave = 64;

// std::lround semantics (round half away from zero) without libm.
__device__ __forceinline__ int LRoundF64(double x) {
  double t = trunc(x);
  const double frac = x - t;  // exact
  if (frac >= 0.5) t += 1.0;
  if (frac <= -0.5) t -= 1.0;
  return static_cast<int>(t);
}
__device__ __forceinline__ int LRoundF32(float x) {
  float t = truncf(x);
  const float frac = x - t;  // exact
  if (frac >= 0.5f) t += 1.0f;
  if (frac <= -0.5f) t -= 1.0f;
  return static_cast<int>(t);
}

// Pointers read from descriptor structs in memory are "generic" to the compiler, which
// then emits flat_load (LDS-aperture check, vmcnt and lgkmcnt both tied up, no partial
// waits).  All such pointers are HBM addresses: say so, and the loads become global_load.
// (Keep the result in an `auto` variable: converting back to a plain pointer drops it.)
#define CMX_GLOBAL __attribute__((address_space(1)))
template <typename T>
__device__ __forceinline__ CMX_GLOBAL T* AsGlobal(T* p) {
  return (CMX_GLOBAL T*)p;
}

// A wavefront-uniform 64-bit value (pointers and sizes that come out of LDS or out of a
// descriptor in memory: the compiler does not know they are uniform and would wrap every
// buffer load that uses them in a waterfall loop).  Through `unsigned`: readfirstlane returns an
// int, and a low word with its top bit set must not sign-extend over the high word.
__device__ __forceinline__ unsigned long long UniformU64(unsigned long long v) {
  const unsigned lo = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(v)));
  const unsigned hi =
      static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(v >> 32)));
  return static_cast<unsigned long long>(lo) | (static_cast<unsigned long long>(hi) << 32);
}
// Buffer resource over `bytes` (< 2 GB) at `base`: raw buffer loads through it are never
// predicated (an offset >= bytes -- kOutOfBuffer -- reads 0 without touching memory), so the
// gathers of an unrolled loop can be issued back to back and consumed at vmcnt(k).
constexpr unsigned kOutOfBuffer = 0xfffffff0u;
constexpr unsigned long long kMaxBufferBytes = 1ull << 31;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t UniformBuffer(const void* base,
                                                                unsigned long long bytes) {
  const unsigned long long b = UniformU64(bytes);
  return __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<void*>(UniformU64(reinterpret_cast<unsigned long long>(base))), 0,
      b < kMaxBufferBytes ? static_cast<int>(b) : 0, 0x00020000);
}

// lround(t / res - 0.5) -- MapLimits::GetCellIndex (mapping/2d/map_limits.h:69-76) --
// without the f64 division in the common case.  With inv = RN(1 / res), v0 = t * inv - 0.5
// differs from the reference's value by less than |t / res| * 2^-50 + 2^-52; when v0 is
// further than 16x that from every half-integer the rounded results agree and rint(v0) is
// returned, otherwise (practically never, NaN / inf included) the exact expression runs.
__device__ __forceinline__ int CellIndexF64(double t, double res, double inv_res) {
  const double q0 = t * inv_res;
  const double v0 = q0 - 0.5;
  const double n = rint(v0);
  const double margin = 0.5 - fabs(v0 - n);            // exact
  if (margin > fabs(q0) * 0x1p-46 + 0x1p-46) return static_cast<int>(n);
  return LRoundF64(t / res - 0.5);
}

// The same cell index from an f32 estimate when that is provably enough.  t32 = (RN32(max)
// - v) * RN32(1 / res) - 0.5 differs from the reference's f64 value by less than
// (|max| + |max - v|) / res * 2^-24 + |q| * 2^-22 (three roundings and the two rounded
// constants); when t32 is further than twice that from every half-integer, rint(t32) is the
// reference's lround.  Otherwise (about one point in a thousand, NaN / inf / huge values
// included) the f64 expression above decides.  The f64 path costs ~160 cycles per wavefront
// and coordinate, most of a scan's preparation; this one ~20.
__device__ __forceinline__ int CellIndexFast(double max_d, float v, double res, double inv_res) {
  const float maxf = static_cast<float>(max_d);
  const float inv_resf = static_cast<float>(inv_res);
  const float a = maxf - v;
  const float b = a * inv_resf;
  const float t = b - 0.5f;
  const float r = rintf(t);
  const float margin = 0.5f - fabsf(t - r);
  const float bound = (fabsf(maxf) + fabsf(a)) * inv_resf * 0x1p-23f + fabsf(b) * 0x1p-21f + 0x1p-20f;
  if (margin > bound && fabsf(t) < 1e6f) return static_cast<int>(r);
  return CellIndexF64(max_d - static_cast<double>(v), res, inv_res);
}

struct F3 { float x, y, z; };
struct Quat { float w, x, y, z; };

__device__ __forceinline__ F3 Cross(const F3& a, const F3& b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
// Eigen 3.3 Quaternion::_transformVector (used by
// sensor/rangefinder_point.h:43-48 through Rigid3::operator*).
__device__ __forceinline__ F3 Rotate(const Quat& q, const F3& v) {
  const F3 qv{q.x, q.y, q.z};
  F3 uv = Cross(qv, v);
  uv.x += uv.x; uv.y += uv.y; uv.z += uv.z;
  const F3 c = Cross(qv, uv);
  return {(v.x + q.w * uv.x) + c.x, (v.y + q.w * uv.y) + c.y, (v.z + q.w * uv.z) + c.z};
}
// Rotate(q, v) for q = (w, 0, 0, z) -- a yaw quaternion -- without the products by the zero
// components.  For finite inputs the x and y results are bit-identical to Rotate's:
//   uv = 2 ((0,0,z) x v)   = 2 (0 v.z - z v.y, z v.x - 0 v.z, .) = 2 (-(z v.y), z v.x, .)
//   c  = (0,0,z) x uv      = (0 uv.z - z uv.y, z uv.x - 0 uv.z, .) = (-(z uv.y), z uv.x, .)
//   out = (v + w uv) + c
// (0 a is a signed zero; b - 0 and 0 - b differ from b and -b at most in the sign of a zero
// result, and a zero's sign disappears in the next addition or in `max - coordinate`).  The z
// result is not computed: the 2D cell index never reads it.  Non-finite coordinates (which
// Rotate would turn into NaNs everywhere) are outside the contract.
__device__ __forceinline__ void RotateZ(float w, float z, float vx, float vy, float* ox,
                                        float* oy) {
  float uvx = -(z * vy), uvy = z * vx;
  uvx += uvx; uvy += uvy;
  const float cx = -(z * uvy), cy = z * uvx;
  *ox = (vx + w * uvx) + cx;
  *oy = (vy + w * uvy) + cy;
}

__device__ __forceinline__ Quat QuatMul(const Quat& a, const Quat& b) {
  return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z,
          a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
          a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
          a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x};
}

// Wave-wide integer sum (all 64 lanes receive the total).  DPP row shifts and
// row broadcasts keep the reduction in the VALU (a __shfl_xor butterfly goes
// through the LDS crossbar six dependent times).
__device__ __forceinline__ int WaveSum(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);   // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);   // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);   // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);   // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, true);   // row_bcast:15
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, true);   // row_bcast:31
  return __builtin_amdgcn_readlane(v, 63);
}
// Wave-wide min / max in the VALU (same DPP ladder as WaveSum; lanes without a source keep
// their own value).  All lanes receive the result.
__device__ __forceinline__ int WaveMinDpp(int v) {
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x111, 0xf, 0xf, false));   // row_shr:1
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x112, 0xf, 0xf, false));   // row_shr:2
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x114, 0xf, 0xf, false));   // row_shr:4
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x118, 0xf, 0xf, false));   // row_shr:8
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x142, 0xa, 0xf, false));   // row_bcast:15
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x143, 0xc, 0xf, false));   // row_bcast:31
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int WaveMaxDpp(int v) {
  v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x111, 0xf, 0xf, false));
  v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x112, 0xf, 0xf, false));
  v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x114, 0xf, 0xf, false));
  v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x118, 0xf, 0xf, false));
  v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x142, 0xa, 0xf, false));
  v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x143, 0xc, 0xf, false));
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int WaveMin(int v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = min(v, __shfl_xor(v, off, 64));
  return v;
}
__device__ __forceinline__ int WaveMax(int v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
  return v;
}
__device__ __forceinline__ unsigned long long WaveMaxU64(unsigned long long v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const unsigned long long o = __shfl_xor(v, off, 64);
    v = o > v ? o : v;
  }
  return v;
}

// The reference's sequential f32 sum of `count` floats in LDS (count a multiple of 64, `p` 16-byte
// aligned), continued from `acc`: one lane, N dependent additions in index order.  The values
// arrive 32 at a time (eight ds_read_b128) in two register banks: a bank is requested BEFORE the
// other bank's 32 additions start, so its LDS latency runs under them (the compiler's own
// schedule requested it behind them and then copied it across: 16 cycles per addition instead of
// the adder's latency).  Loads and waits are asm -- a wait names its bank as an in/out operand, so
// that the additions stay behind it; LDS returns in order: "at most 8 pending" = the older bank.
typedef float ChainF4 __attribute__((ext_vector_type(4)));
template <int kImm>
__device__ __forceinline__ void ChainRead(ChainF4* out, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(*out) : "v"(addr), "i"(kImm));
}
__device__ __forceinline__ void ChainReadBank(ChainF4 (&b)[8], unsigned addr) {
  ChainRead<0>(&b[0], addr);   ChainRead<16>(&b[1], addr);  ChainRead<32>(&b[2], addr);
  ChainRead<48>(&b[3], addr);  ChainRead<64>(&b[4], addr);  ChainRead<80>(&b[5], addr);
  ChainRead<96>(&b[6], addr);  ChainRead<112>(&b[7], addr);
}
template <int kPending>
__device__ __forceinline__ void ChainWaitBank(ChainF4 (&b)[8]) {
  asm volatile("s_waitcnt lgkmcnt(%8)"
               : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7])
               : "i"(kPending));
}
__device__ __forceinline__ float ChainAddBank(float acc, const ChainF4 (&b)[8]) {
#pragma unroll
  for (int k = 0; k < 8; ++k) { acc += b[k].x; acc += b[k].y; acc += b[k].z; acc += b[k].w; }
  return acc;
}
__device__ __forceinline__ float ChainSumLds(const float* p, int count, float acc) {
  const unsigned base = static_cast<unsigned>(reinterpret_cast<uintptr_t>(
      (const __attribute__((address_space(3))) float*)p));
  const int last = count - 32;                     // first float of the last bank
  ChainF4 a[8], b[8];
  ChainReadBank(a, base);
  for (int j = 0; j < count; j += 64) {
    ChainReadBank(b, base + 4u * static_cast<unsigned>(j + 32));
    ChainWaitBank<8>(a);
    acc = ChainAddBank(acc, a);
    ChainReadBank(a, base + 4u * static_cast<unsigned>(min(j + 64, last)));   // (past the end: re-read, unused)
    ChainWaitBank<8>(b);
    acc = ChainAddBank(acc, b);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  return acc;
}

// Optional in-kernel timeline (environment CMX_TIMELINE=1, tools only): thread 0 of a block
// stores the 100 MHz wall clock at phase boundaries, 16 stamps per block.
constexpr int kTimelineStamps = 16;
__device__ __forceinline__ void Stamp(unsigned long long* timeline, int block, int k) {
  if (timeline && threadIdx.x == 0)
    timeline[static_cast<size_t>(block) * kTimelineStamps + k] = wall_clock64();
}

}  // namespace cmx

#endif  // CMX_DEVICE_H_
