// How many workgroups of a given size / LDS footprint the runtime keeps resident per CU.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/occupancy_probe.hip -o /tmp/occ && /tmp/occ
#include <hip/hip_runtime.h>
#include <cstdio>
template <int kThreads>
__global__ void __launch_bounds__(kThreads) Probe(int* out) {
  extern __shared__ int smem[];
  smem[threadIdx.x] = threadIdx.x;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = smem[7];
}
template <int kThreads>
void Report() {
  for (size_t lds : {size_t{16} << 10, size_t{40} << 10, size_t{52} << 10, size_t{64} << 10, size_t{76} << 10, size_t{80} << 10}) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(Probe<kThreads>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
    int blocks = -1;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, Probe<kThreads>, kThreads, lds);
    printf("threads %4d lds %3zu KB: %d resident per CU (%s)\n", kThreads, lds >> 10, blocks, hipGetErrorString(e));
  }
}
int main() {
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  printf("%s: %d CUs, maxThreadsPerMultiProcessor %d, sharedMemPerMultiprocessor %zu, sharedMemPerBlock %zu, maxSharedMemoryPerBlockOptin %zu, regsPerMultiprocessor %d\n",
         p.name, p.multiProcessorCount, p.maxThreadsPerMultiProcessor, p.sharedMemPerMultiprocessor, p.sharedMemPerBlock, p.sharedMemPerBlockOptin, p.regsPerMultiprocessor);
  Report<1024>(); Report<512>(); Report<256>();
  return 0;
}
