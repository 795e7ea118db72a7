// How fast does a CU take the fast-2D front end's access pattern -- per wave-instruction FOUR
// 64-byte rows (lane group g = lane / 16 reads dword `lane % 16` of its own row) at random
// 64-byte-aligned places of a 256 KB table -- from memory (L1 / L2) and from LDS (the half of
// the table that fits: 128 KB)?  The numbers behind DESIGN.md 5.1 / 8: the memory figure is what
// PrepScoreFusedKernel's scoring loop runs at, the LDS figure what a plane-resident front end
// could run at.
//   hipcc --offload-arch=gfx950 -O3 tools/row_gather_ceiling.hip -o tools/bin/row_gather_ceiling
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { \
  fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int kIters = 1024;     // gather instructions per wavefront
constexpr int kUnroll = 16;      // in flight

template <bool kLds, int kRowsPerInstr>
__global__ void __launch_bounds__(kLds ? 1024 : 256) RowGather(const uint32_t* __restrict__ table, int table_rows,
                                                 unsigned long long* out) {
  extern __shared__ uint32_t lds[];
  const unsigned lane = threadIdx.x & 63;
  const int rows = kLds ? min(table_rows, 2048) : table_rows;       // 2048 rows = 128 KB
  if (kLds) {
    for (int i = threadIdx.x; i < rows * 16; i += blockDim.x) lds[i] = table[i];
    __syncthreads();
  }
  // a lane group's row: pseudo-random per (wave, instruction, group)
  const unsigned lanes_per_row = 64 / kRowsPerInstr;
  const unsigned group = lane / lanes_per_row, sub = lane % 16;
  unsigned state = ((blockIdx.x * 4u + (threadIdx.x >> 6)) * 64u + group) * 2654435761u + 12345u;
  unsigned long long acc = 0;
  for (int it = 0; it < kIters; it += kUnroll) {
    uint32_t v[kUnroll];
#pragma unroll
    for (int k = 0; k < kUnroll; ++k) {
      state = state * 1664525u + 1013904223u;
      const unsigned row = (state >> 10) & static_cast<unsigned>(rows - 1);     // (rows: a power of two)
      v[k] = kLds ? lds[row * 16 + sub] : table[row * 16 + sub];
    }
#pragma unroll
    for (int k = 0; k < kUnroll; ++k) acc += v[k];
  }
  if (acc == 0x123456789abcdefull) out[0] = acc;      // (keeps the loads alive)
}

template <bool kLds, int kRowsPerInstr>
int Run(const uint32_t* d_table, int rows, unsigned long long* d_out, int cus, const char* what) {
  const int waves_per_cu = kLds ? 16 : 16;
  const int blocks = cus * waves_per_cu / 4;
  const size_t lds_bytes = kLds ? 128 * 1024 : 0;
  if (kLds)
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(RowGather<kLds, kRowsPerInstr>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_bytes)));
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  // (LDS: one workgroup of 1024 threads per CU so that the 128 KB table exists once per CU)
  const dim3 grid(kLds ? cus : blocks), block(kLds ? 1024 : 256);
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipEventRecord(a));
    RowGather<kLds, kRowsPerInstr><<<grid, block, lds_bytes>>>(d_table, rows, d_out);
    CHECK(hipGetLastError());
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
  }
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, a, b));
  const double waves = static_cast<double>(grid.x) * (block.x / 64);
  const double instr = waves * kIters;
  const double per_cu_instr = instr / cus;
  const double cycles = ms * 1e-3 * 2.4e9 / per_cu_instr;
  printf("%-44s %8.3f ms  %7.2f cycles per wave-instruction and CU  = %6.2f per 64-byte row  (%5.1f B/clk/CU)%s\n",
         what, ms, cycles, cycles / kRowsPerInstr, 64.0 * kRowsPerInstr / cycles,
         kLds ? "  [includes the 128 KB fill]" : "");
  return 0;
}

int main() {
  int cus = 0;
  CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  const int rows = 4096;                                  // 256 KB of 64-byte rows
  std::vector<uint32_t> h(rows * 16);
  for (size_t i = 0; i < h.size(); ++i) h[i] = static_cast<uint32_t>(i * 2654435761u);
  uint32_t* d_table;
  unsigned long long* d_out;
  CHECK(hipMalloc(&d_table, h.size() * 4));
  CHECK(hipMalloc(&d_out, 8));
  CHECK(hipMemcpy(d_table, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  printf("# %d CUs, 2.4 GHz assumed; %d gather instructions per wavefront, 16 wavefronts per CU\n", cus, kIters);
  if (Run<false, 4>(d_table, rows, d_out, cus, "memory, 4 random rows per instruction")) return 1;
  if (Run<false, 2>(d_table, rows, d_out, cus, "memory, 2 random rows per instruction")) return 1;
  if (Run<false, 1>(d_table, rows, d_out, cus, "memory, 1 row per instruction (coherent)")) return 1;
  if (Run<true, 4>(d_table, rows, d_out, cus, "LDS (128 KB), 4 random rows per instruction")) return 1;
  if (Run<true, 2>(d_table, rows, d_out, cus, "LDS (128 KB), 2 random rows per instruction")) return 1;
  if (Run<true, 1>(d_table, rows, d_out, cus, "LDS (128 KB), 1 row per instruction")) return 1;
  return 0;
}
