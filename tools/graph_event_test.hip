// Does hipEventElapsedTime work on events recorded inside a captured graph? (ROCm 7.2 probe)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void Spin(int* p, int iters) {
  int v = 0;
  for (int i = 0; i < iters; ++i) v += __builtin_amdgcn_readfirstlane(i) ^ v;
  if (threadIdx.x == 0 && blockIdx.x == 0) *p = v;
}
int main() {
  hipStream_t s; CK(hipStreamCreate(&s));
  int* d; CK(hipMalloc(&d, 4));
  hipEvent_t e0, e1, e2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  CK(hipEventRecord(e0, s));
  for (int k = 0; k < 6; ++k) Spin<<<256, 256, 0, s>>>(d, 2000);
  CK(hipEventRecord(e1, s));
  for (int k = 0; k < 6; ++k) Spin<<<256, 256, 0, s>>>(d, 2000);
  CK(hipEventRecord(e2, s));
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int rep = 0; rep < 3; ++rep) {
    auto t0 = std::chrono::steady_clock::now();
    CK(hipGraphLaunch(ge, s));
    auto t1 = std::chrono::steady_clock::now();
    CK(hipStreamSynchronize(s));
    auto t2 = std::chrono::steady_clock::now();
    float a = -1, b = -1;
    hipError_t ea = hipEventElapsedTime(&a, e0, e1), eb = hipEventElapsedTime(&b, e0, e2);
    printf("graph: launch %.1f us total %.1f us; elapsed e0-e1 %.3f ms (%s) e0-e2 %.3f ms (%s)\n",
           std::chrono::duration<double, std::micro>(t1 - t0).count(),
           std::chrono::duration<double, std::micro>(t2 - t0).count(), a, hipGetErrorString(ea), b,
           hipGetErrorString(eb));
  }
  for (int rep = 0; rep < 3; ++rep) {
    auto t0 = std::chrono::steady_clock::now();
    CK(hipEventRecord(e0, s));
    for (int k = 0; k < 6; ++k) Spin<<<256, 256, 0, s>>>(d, 2000);
    CK(hipEventRecord(e1, s));
    for (int k = 0; k < 6; ++k) Spin<<<256, 256, 0, s>>>(d, 2000);
    CK(hipEventRecord(e2, s));
    auto t1 = std::chrono::steady_clock::now();
    CK(hipStreamSynchronize(s));
    auto t2 = std::chrono::steady_clock::now();
    float a = -1, b = -1;
    CK(hipEventElapsedTime(&a, e0, e1)); CK(hipEventElapsedTime(&b, e0, e2));
    printf("stream: launch %.1f us total %.1f us; elapsed e0-e1 %.3f ms e0-e2 %.3f ms\n",
           std::chrono::duration<double, std::micro>(t1 - t0).count(),
           std::chrono::duration<double, std::micro>(t2 - t0).count(), a, b);
  }
  return 0;
}
