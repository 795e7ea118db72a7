// f32 atan / atan2 bit for bit as the libm the reference links against computes them (glibc 2.35,
// x86-64: sysdeps/ieee754/flt-32/{s_atanf,e_atan2f}.c, the fdlibm single-precision algorithm --
// not part of /root/reference: a third-party dependency, restated from its published form):
// argument reduction to |t| < 7/16 by one of four identities, odd / even split of an eleven-term
// polynomial, hi / lo words of atan(0.5), atan(1), atan(1.5), atan(inf); atan2 = atan of the
// correctly rounded quotient plus the quadrant correction with a two-word pi.  Plain f32
// operations in a fixed order (the library is built with -ffp-contract=off; hipcc's f32 division
// is correctly rounded by default): the same bits on the host and on the device.
// The constants are the bit patterns the C compiler makes of glibc's DECIMAL literals (the first
// polynomial coefficient, 3.3333334327e-01, is 0x3eaaaaab, not the 0x3eaaaaaa of its comment).
// Pinned on the host by tests/test_atan2f.py: 0 mismatches against libm's atanf over every 7th
// float bit pattern and against atan2f over 4 x 10^8 pairs (any bits, scan-sized, sub-metre,
// equal-exponent) when this header was written.  rotational_scan_matcher.cc:57-58,:82 (the two
// common::atan2 calls of the histogram) are the only users.
#ifndef CMX_ATAN2F_GLIBC_H_
#define CMX_ATAN2F_GLIBC_H_
#include <stdint.h>
#include <string.h>
#ifndef CMX_HD
#ifdef __HIPCC__
#define CMX_HD __host__ __device__
#else
#define CMX_HD
#endif
#endif
namespace cmx {
CMX_HD inline float BitsToFloat(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
CMX_HD inline uint32_t FloatToBits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

CMX_HD inline float AtanfGlibc(float x) {
  const uint32_t hx = FloatToBits(x), ix = hx & 0x7fffffffu;
  const bool negative = (hx >> 31) != 0;
  if (ix >= 0x4c000000u) {                        // |x| >= 2^25 (or NaN)
    if (ix > 0x7f800000u) return x + x;
    const float r = BitsToFloat(0x3fc90fdau) + BitsToFloat(0x33a22168u);
    return negative ? -r : r;
  }
  int id;
  if (ix < 0x3ee00000u) {                         // |x| < 7/16
    if (ix < 0x31000000u) return x;               // |x| < 2^-29
    id = -1;
  } else {
    x = BitsToFloat(ix);
    if (ix < 0x3f980000u) {                       // |x| < 19/16
      if (ix < 0x3f300000u) { id = 0; x = (2.0f * x - 1.0f) / (2.0f + x); }
      else                  { id = 1; x = (x - 1.0f) / (x + 1.0f); }
    } else {
      if (ix < 0x401c0000u) { id = 2; x = (x - 1.5f) / (1.0f + 1.5f * x); }
      else                  { id = 3; x = -1.0f / x; }
    }
  }
  const float a0 = BitsToFloat(0x3eaaaaabu), a1 = BitsToFloat(0xbe4ccccdu),
              a2 = BitsToFloat(0x3e124925u), a3 = BitsToFloat(0xbde38e38u),
              a4 = BitsToFloat(0x3dba2e6eu), a5 = BitsToFloat(0xbd9d8795u),
              a6 = BitsToFloat(0x3d886b35u), a7 = BitsToFloat(0xbd6ef16bu),
              a8 = BitsToFloat(0x3d4bda59u), a9 = BitsToFloat(0xbd15a221u),
              a10 = BitsToFloat(0x3c8569d7u);
  const float z = x * x, w = z * z;
  const float s1 = z * (a0 + w * (a2 + w * (a4 + w * (a6 + w * (a8 + w * a10)))));
  const float s2 = w * (a1 + w * (a3 + w * (a5 + w * (a7 + w * a9))));
  if (id < 0) return x - x * (s1 + s2);
  const uint32_t hi[4] = {0x3eed6338u, 0x3f490fdau, 0x3f7b985eu, 0x3fc90fdau};
  const uint32_t lo[4] = {0x31ac3769u, 0x33222168u, 0x33140fb4u, 0x33a22168u};
  const float r = BitsToFloat(hi[id]) - ((x * (s1 + s2) - BitsToFloat(lo[id])) - x);
  return negative ? -r : r;
}

CMX_HD inline float Atan2fGlibc(float y, float x) {
  const float pi = BitsToFloat(0x40490fdbu), pi_lo = BitsToFloat(0xb3bbbd2eu),
              pi_o_2 = BitsToFloat(0x3fc90fdbu), pi_o_4 = BitsToFloat(0x3f490fdbu);
  const float tiny = 1.0e-30f;
  const int32_t hx = static_cast<int32_t>(FloatToBits(x)), hy = static_cast<int32_t>(FloatToBits(y));
  const int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
  if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
  if (hx == 0x3f800000) return AtanfGlibc(y);
  const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
  if (iy == 0) {
    switch (m) {
      case 0: case 1: return y;
      case 2: return pi + tiny;
      default: return -pi - tiny;
    }
  }
  if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
  if (ix == 0x7f800000) {
    if (iy == 0x7f800000) {
      switch (m) {
        case 0: return pi_o_4 + tiny;
        case 1: return -pi_o_4 - tiny;
        case 2: return 3.0f * pi_o_4 + tiny;
        default: return -3.0f * pi_o_4 - tiny;
      }
    }
    switch (m) {
      case 0: return 0.0f;
      case 1: return -0.0f;
      case 2: return pi + tiny;
      default: return -pi - tiny;
    }
  }
  if (iy == 0x7f800000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
  const int32_t k = (iy - ix) >> 23;
  float z;
  if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
  else if (hx < 0 && k < -60) z = 0.0f;
  else {
    const float q = y / x;
    z = AtanfGlibc(BitsToFloat(FloatToBits(q) & 0x7fffffffu));
  }
  switch (m) {
    case 0: return z;
    case 1: return BitsToFloat(FloatToBits(z) ^ 0x80000000u);
    case 2: return pi - (z - pi_lo);
    default: return (z - pi_lo) - pi;
  }
}
}  // namespace cmx
#endif
