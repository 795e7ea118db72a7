// A C++ caller's thread pool in miniature (bench and probe tooling, host only): T std::threads
// each issue `calls` full-submap searches through the C ABI, each with its own result buffers --
// how the workers of the reference's common::ThreadPool call the matcher from
// ConstraintBuilder2D::ComputeConstraint (constraints/constraint_builder_2d.cc:194-236).  The
// entry point is handed over as a pointer (the product library is not linked here); the timed
// region is inside, from the moment all threads are released to the last join, so a Python
// caller's interpreter lock is not part of what is measured.
#include <atomic>
#include <chrono>
#include <cstdint>
#include <thread>
#include <vector>

#include "../../../include/cartographer_mi355x.h"

extern "C" {

typedef cmx_status (*cmx_full_submap_batch_fn)(const cmx_fast2d* const*, int32_t, const cmx_cloud*,
                                               float, int32_t*, float*, cmx_pose2d*,
                                               cmx_match_stats*);

// clouds[k % num_clouds] is the scan of a thread's k-th call.  Returns the wall seconds of the
// region (< 0: a call failed, its status negated); *candidates / *found are summed over all calls.
double cmx_thread_driver_fast2d(void* entry, const cmx_fast2d* const* matchers, int32_t num_matchers,
                                const cmx_cloud* const* clouds, int32_t num_clouds, float min_score,
                                int32_t threads, int32_t calls, int64_t* candidates, int64_t* found) {
  const auto fn = reinterpret_cast<cmx_full_submap_batch_fn>(entry);
  std::atomic<int> ready{0};
  std::atomic<bool> go{false};
  std::atomic<int> failed{0};
  std::atomic<long long> cand{0}, hits{0};
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t) {
    pool.emplace_back([&, t] {
      std::vector<int32_t> f(num_matchers);
      std::vector<float> s(num_matchers);
      std::vector<cmx_pose2d> p(num_matchers);
      long long my_cand = 0, my_hits = 0;
      ready.fetch_add(1);
      while (!go.load(std::memory_order_acquire)) std::this_thread::yield();
      for (int k = 0; k < calls; ++k) {
        cmx_match_stats st{};
        const cmx_status status = fn(matchers, num_matchers, clouds[(t * calls + k) % num_clouds],
                                     min_score, f.data(), s.data(), p.data(), &st);
        if (status != CMX_OK) { failed.store(static_cast<int>(status)); break; }
        my_cand += st.candidates_scored;
        for (int i = 0; i < num_matchers; ++i) my_hits += f[i] != 0;
      }
      cand.fetch_add(my_cand);
      hits.fetch_add(my_hits);
    });
  }
  while (ready.load() < threads) std::this_thread::yield();
  const auto t0 = std::chrono::steady_clock::now();
  go.store(true, std::memory_order_release);
  for (std::thread& th : pool) th.join();
  const double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (candidates) *candidates = cand.load();
  if (found) *found = hits.load();
  const int bad = failed.load();
  return bad ? -static_cast<double>(bad) : seconds;
}

}  // extern "C"
