// Host-side pin of cartographer_amd/csrc/cmx_atan2f.h against this machine's libm (the one the
// reference's rotational_scan_matcher.cc is linked against here).  Built and run by
// tests/test_atan2f.py:   atan2f_check <pairs>   -> prints the two mismatch counts.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>

#include "cmx_atan2f.h"

int main(int argc, char** argv) {
  const long long pairs = argc > 1 ? atoll(argv[1]) : 20000000;
  const unsigned stride = argc > 2 ? static_cast<unsigned>(atoi(argv[2])) : 97;
  long long bad_atan = 0, bad_atan2 = 0;
  for (uint64_t u = 0; u < (1ull << 32); u += stride) {
    const float x = cmx::BitsToFloat(static_cast<uint32_t>(u));
    const float a = atanf(x), b = cmx::AtanfGlibc(x);
    if (cmx::FloatToBits(a) != cmx::FloatToBits(b) && !(a != a && b != b)) {
      if (bad_atan < 5) printf("atanf(%a) = %a, header %a\n", x, a, b);
      ++bad_atan;
    }
  }
  const auto check = [&](float y, float x) {
    const float a = atan2f(y, x), b = cmx::Atan2fGlibc(y, x);
    if (cmx::FloatToBits(a) != cmx::FloatToBits(b) && !(a != a && b != b)) {
      if (bad_atan2 < 5) printf("atan2f(%a, %a) = %a, header %a\n", y, x, a, b);
      ++bad_atan2;
    }
  };
  const float special[] = {0.f, -0.f, 1.f, -1.f, INFINITY, -INFINITY, NAN, 1e-40f, -1e-40f,
                           3e38f, -3e38f, 0.5f, 2.f, 1e-30f, 1e30f, 0.4375f, 0.6875f, 1.1875f,
                           2.4375f, 33554432.f};
  for (float y : special) for (float x : special) check(y, x);
  std::mt19937_64 rng(12345);
  const float unit = 1.0f / 2147483648.0f;
  for (long long i = 0; i < pairs; ++i) {
    const uint64_t r = rng();
    const uint32_t lo = static_cast<uint32_t>(r), hi = static_cast<uint32_t>(r >> 32);
    float y, x;
    switch (i & 3) {
      case 0: y = cmx::BitsToFloat(lo); x = cmx::BitsToFloat(hi); break;            // any bits
      case 1: y = static_cast<int32_t>(lo) * unit * 30.f;                            // scan-sized
              x = static_cast<int32_t>(hi) * unit * 30.f; break;
      case 2: y = static_cast<int32_t>(lo) * unit * 0.9f;                            // the walk's deltas
              x = static_cast<int32_t>(hi) * unit * 0.9f; break;
      default: {                                                                     // nearby exponents
        const uint32_t e = 100 + (r & 63);
        y = cmx::BitsToFloat((e << 23) | ((lo >> 8) & 0x007fffffu) | (lo & 0x80000000u));
        x = cmx::BitsToFloat(((e + ((r >> 40) & 7) - 3) << 23) | (hi & 0x807fffffu));
      }
    }
    check(y, x);
  }
  printf("atanf_mismatches %lld\natan2f_mismatches %lld\n", bad_atan, bad_atan2);
  return bad_atan || bad_atan2 ? 1 : 0;
}
