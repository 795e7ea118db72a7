// Micro-benchmark (round 5): the cost of ds_read_b64 at addresses of different alignment, as the
// bound kernel of RT-2D issues them (rt_2d_bounds.h: a block row is 8 bytes at ANY byte address).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/lds_unaligned.hip -o tools/bin/lds_unaligned
// Prints cycles per wave-instruction for 64 lanes reading pseudo-random rows of a 60 KB LDS image.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned uint2v __attribute__((ext_vector_type(2)));
struct __attribute__((packed)) Row { uint2v v; };

template <int kMode>
__global__ void __launch_bounds__(512) Probe(unsigned long long* out, unsigned* sink, int iters) {
  extern __shared__ unsigned char lds[];
  const int tid = threadIdx.x;
  for (int i = tid; i < 15360; i += 512) reinterpret_cast<unsigned*>(lds)[i] = i * 2654435761u;
  __syncthreads();
  unsigned h = tid * 747796405u + 2891336453u;
  unsigned acc0 = 0, acc1 = 0;
  const unsigned long long t0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
    h = h * 1664525u + 1013904223u;
    int at = (h >> 8) % 59000;
    if (kMode == 0) at &= ~7;            // 8-byte aligned
    if (kMode == 1) at &= ~3;            // 4-byte aligned
    if (kMode == 2) at &= ~1;            // 2-byte aligned
                                         // 3: any byte; 4: any byte, two aligned dwords + third, shifted
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      if (kMode <= 3) {
        const uint2v d = reinterpret_cast<const Row*>(lds + at + j * 128)->v;
        acc0 += d.x; acc1 += d.y;
      } else {
        const int base = (at + j * 128) & ~3, sh = ((at + j * 128) & 3) * 8;
        const unsigned a = *reinterpret_cast<const unsigned*>(lds + base);
        const unsigned b = *reinterpret_cast<const unsigned*>(lds + base + 4);
        const unsigned c = *reinterpret_cast<const unsigned*>(lds + base + 8);
        acc0 += __builtin_amdgcn_alignbit(b, a, sh);
        acc1 += __builtin_amdgcn_alignbit(c, b, sh);
      }
    }
  }
  const unsigned long long t1 = wall_clock64();
  if (tid == 0) out[blockIdx.x] = t1 - t0;
  sink[blockIdx.x * 512 + tid] = acc0 ^ acc1;
}

int main() {
  unsigned long long* d_out; unsigned* d_sink;
  hipMalloc(&d_out, 512 * 8); hipMalloc(&d_sink, 512 * 512 * 4);
  const int iters = 2000;
  const char* names[] = {"8-byte aligned", "4-byte aligned", "2-byte aligned", "any byte", "any byte, 3 x b32 + alignbit"};
  for (int mode = 0; mode < 5; ++mode) {
    for (int blocks : {1, 512}) {
      auto launch = [&](auto k) { hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 61440, 0, d_out, d_sink, iters); };
      for (int rep = 0; rep < 2; ++rep) {
        if (mode == 0) launch(Probe<0>); else if (mode == 1) launch(Probe<1>); else if (mode == 2) launch(Probe<2>);
        else if (mode == 3) launch(Probe<3>); else launch(Probe<4>);
        hipDeviceSynchronize();
      }
      unsigned long long t[512];
      hipMemcpy(t, d_out, blocks * 8, hipMemcpyDeviceToHost);
      double ticks = 0; for (int b = 0; b < blocks; ++b) ticks += t[b]; ticks /= blocks;
      // wall_clock64: 100 MHz; 8 wavefronts per workgroup each issue 7 reads per iteration
      printf("%-32s %3d workgroup(s) of 512: %.1f ns per iteration of 7 row reads per wave (%.0f us)\n", names[mode], blocks,
             ticks * 10.0 / iters, ticks * 0.01);
    }
  }
  return 0;
}
