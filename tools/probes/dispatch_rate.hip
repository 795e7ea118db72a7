// How fast does one device take chains of K dependent launches of B workgroups x 256 threads from T
// host threads on T streams -- empty kernels, so what is measured is the dispatch of workgroups,
// not their work?  (round 6: the sixteen-thread C2 line completes a search every 35 us whatever the
// kernels contain; a search is five launches and ~2 000 workgroups.)
//   built by cartographer_amd/build.py (build_tools) into tools/bin/dispatch_rate
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <unistd.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); std::exit(1); } } while (0)
__global__ void __launch_bounds__(256) Tiny(int* p, int spin) {
  // (`spin` dependent shifts: a workgroup that lives for a while without touching memory)
  int v = threadIdx.x;
  for (int i = 0; i < spin; ++i) v = (v * 1664525 + 1013904223) >> 1;
  if (v == 0x7fffffff && p == nullptr) *p = v;
}

double Run(int threads, int chains, int kernels, int blocks, int spin) {
  std::vector<std::thread> pool;
  std::atomic<int> ready{0};
  std::atomic<bool> go{false};
  const auto body = [&](int) {
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    int* d; CK(hipMalloc(&d, 64));
    ready++;
    while (!go.load()) {}
    for (int c = 0; c < chains; ++c) {
      for (int k = 0; k < kernels; ++k) Tiny<<<blocks, 256, 0, s>>>(d, spin);
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

int main(int argc, char** argv) {
  CK(hipSetDevice(0));
  if (argc >= 3) {
    // dispatch_rate <threads> <chains>: one configuration (several PROCESSES side by side: is it
    // the runtime -- one per process -- or the device that takes 3 us per launch?)
    const int threads = std::atoi(argv[1]), chains = std::atoi(argv[2]);
    Run(threads, 50, 5, 64, 0);
    const auto t0 = std::chrono::steady_clock::now();
    const double per = Run(threads, chains, 5, 64, 0);
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("process %d: %d threads x %d chains of 5 launches: %.1f us per chain, %.0f chains/s over %.2f s\n",
           static_cast<int>(getpid()), threads, chains, per, threads * chains / s, s);
    return 0;
  }
  Run(1, 50, 5, 64, 0);
  for (int spin : {0, 2000})
    for (int blocks : {1, 64, 256, 512, 1024, 2048})
      for (int threads : {1, 16})
        printf("5 launches x %4d workgroups (spin %4d), %2d threads: %.1f us per chain (whole process)\n",
               blocks, spin, threads, Run(threads, 300, 5, blocks, spin));
  return 0;
}
