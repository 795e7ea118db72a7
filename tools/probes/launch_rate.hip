// How many dependent kernel chains per second can T host threads push through T streams of ONE
// device -- the shape of the 8-thread bench line (a search = one small upload + seven dependent
// launches + a synchronisation)?  Empty kernels: what is measured is the runtime's launch path.
// Three forms: seven launches, the same seven as ONE captured graph, one launch.
//   built by cartographer_amd/build.py (build_tools) into tools/bin/launch_rate
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); std::exit(1); } } while (0)
__global__ void Tiny(int* p) { if (threadIdx.x == 0 && blockIdx.x == 0 && p == nullptr) *p = 0; }

double Run(int threads, int chains, int mode, int kernels) {
  std::vector<std::thread> pool;
  std::atomic<int> ready{0};
  std::atomic<bool> go{false};
  const auto body = [&](int) {
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    int* d; CK(hipMalloc(&d, 64));
    int* h; CK(hipHostMalloc(&h, 64));
    hipGraphExec_t ge = nullptr;
    if (mode == 1) {
      hipGraph_t g;
      CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      for (int k = 0; k < kernels; ++k) Tiny<<<64, 64, 0, s>>>(d);
      CK(hipStreamEndCapture(s, &g));
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    }
    ready++;
    while (!go.load()) {}
    for (int c = 0; c < chains; ++c) {
      CK(hipMemcpyAsync(d, h, 64, hipMemcpyHostToDevice, s));
      if (mode == 1) CK(hipGraphLaunch(ge, s));
      else for (int k = 0; k < (mode == 2 ? 1 : kernels); ++k) Tiny<<<64, 64, 0, s>>>(d);
      CK(hipStreamSynchronize(s));
    }
  };
  for (int t = 0; t < threads; ++t) pool.emplace_back(body, t);
  while (ready.load() != threads) {}
  const auto t0 = std::chrono::steady_clock::now();
  go = true;
  for (auto& t : pool) t.join();
  const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  return us / (static_cast<double>(threads) * chains);
}

int main() {
  CK(hipSetDevice(0));
  Run(1, 50, 0, 7);
  const char* names[3] = {"7 launches", "1 graph of 7", "1 launch"};
  for (int mode = 0; mode < 3; ++mode)
    for (int threads : {1, 2, 4, 8, 16})
      printf("%-14s %2d threads: %.1f us per chain (whole process)\n", names[mode], threads,
             Run(threads, 400, mode, 7));
  return 0;
}
