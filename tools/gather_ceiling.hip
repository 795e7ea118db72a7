// Re-runnable gather-issue ceiling of the chip (replaces the transcribed
// profiles/r02_rt3d_gather_ceiling.txt): how many wave-wide gather instructions per second
// an MI355X issues when the addresses cost (almost) nothing to compute and the table is
// cache-resident -- by payload width and address pattern.  The branch-and-bound expansions and
// the RT-3D passes are priced against these numbers next to the guide's L2 / LDS peaks.
//   hipcc --offload-arch=gfx950 -O3 tools/gather_ceiling.hip -o /tmp/gather_ceiling && /tmp/gather_ceiling
// Output: one line per (width, pattern, table size): G lookups/s (lanes x instructions), cycles
// per wave-instruction per CU at the measured clock (2.4 GHz nominal), useful GB/s.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { \
  fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int kIters = 512;      // gathers per thread
constexpr int kUnroll = 8;

// pattern 0: coherent (lane i reads element base + i); 1: 64 distinct 128-byte lines per
// instruction (lane i reads base + 32 i elements ...); 2: pseudo-random inside the table.
template <typename T, int kPattern>
__global__ void __launch_bounds__(256) Gather(const T* __restrict__ table, unsigned mask,
                                              unsigned long long* out) {
  const unsigned lane = threadIdx.x & 63;
  unsigned state = (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
  unsigned long long acc = 0;
  unsigned base = blockIdx.x * 977u;
  for (int it = 0; it < kIters; it += kUnroll) {
    T v[kUnroll];
#pragma unroll
    for (int k = 0; k < kUnroll; ++k) {
      unsigned idx;
      if (kPattern == 0) idx = base + lane;
      else if (kPattern == 1) idx = base + lane * (128 / sizeof(T));
      else { state = state * 1664525u + 1013904223u; idx = state >> 8; }
      v[k] = table[idx & mask];
      base += 4099u;
    }
#pragma unroll
    for (int k = 0; k < kUnroll; ++k) acc += static_cast<unsigned long long>(v[k]);
  }
  if (acc == 0x123456789abcdefull) out[0] = acc;     // keeps the loads alive
}

template <typename T, int kPattern>
int Run(const char* width, const char* pattern, size_t table_bytes, int blocks) {
  T* table = nullptr;
  unsigned long long* out = nullptr;
  CHECK(hipMalloc(&table, table_bytes));
  CHECK(hipMalloc(&out, 8));
  CHECK(hipMemset(table, 1, table_bytes));
  const unsigned mask = static_cast<unsigned>(table_bytes / sizeof(T)) - 1;
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  Gather<T, kPattern><<<blocks, 256>>>(table, mask, out);      // warm
  CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    CHECK(hipEventRecord(a));
    Gather<T, kPattern><<<blocks, 256>>>(table, mask, out);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, a, b));
    best = ms < best ? ms : best;
  }
  const double lookups = static_cast<double>(blocks) * 256 * kIters;
  const double per_s = lookups / (best * 1e-3);
  const double wave_instr_per_s = per_s / 64;
  printf("%-6s %-22s table %8zu KiB  %8.1f Glookup/s  %6.2f cycles/wave-instr/CU @2.4GHz  %8.1f GB/s useful\n",
         width, pattern, table_bytes >> 10, per_s / 1e9, 2.4e9 * 256 / wave_instr_per_s,
         per_s * sizeof(T) / 1e9);
  (void)hipFree(table);
  (void)hipFree(out);
  return 0;
}

int main() {
  int dev = 0;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, dev));
  printf("# %s, %d CUs, clock %d MHz; %d gathers per thread, 16 waves per CU resident\n", prop.name,
         prop.multiProcessorCount, prop.clockRate / 1000, kIters);
  const int blocks = prop.multiProcessorCount * 32;          // 8 waves per SIMD, several rounds
  for (size_t kib : {256u, 4096u, 65536u}) {
    const size_t bytes = kib << 10;
    if (Run<uint8_t, 0>("u8", "coherent", bytes, blocks)) return 1;
    if (Run<uint8_t, 1>("u8", "64 lines/instr", bytes, blocks)) return 1;
    if (Run<uint8_t, 2>("u8", "random", bytes, blocks)) return 1;
    if (Run<uint32_t, 0>("u32", "coherent", bytes, blocks)) return 1;
    if (Run<uint32_t, 2>("u32", "random", bytes, blocks)) return 1;
    if (Run<unsigned long long, 0>("u64", "coherent", bytes, blocks)) return 1;
    if (Run<unsigned long long, 1>("u64", "64 lines/instr", bytes, blocks)) return 1;
    if (Run<unsigned long long, 2>("u64", "random", bytes, blocks)) return 1;
  }
  return 0;
}
