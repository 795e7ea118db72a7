// Host plumbing: error text, device selection, workspace pool, library entry
// points that are not tied to one matcher.
#include "cmx_common.h"

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <exception>
#include <thread>

#include <execinfo.h>
#include <signal.h>
#include <unistd.h>

#include <algorithm>
#include <cstdlib>
#include <map>
#include <memory>

namespace cmx {
namespace {
// Debugging aid (CMX_SYNC=1): print a native backtrace on SIGSEGV / SIGABRT.
void CrashHandler(int sig) {
  void* frames[64];
  const int n = backtrace(frames, 64);
  const char msg[] = "[cmx] fatal signal, native backtrace:\n";
  (void)!write(2, msg, sizeof(msg) - 1);
  backtrace_symbols_fd(frames, n, 2);
  signal(sig, SIG_DFL);
  raise(sig);
}
// The HIP runtime maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4);
// kernels of different streams that share a queue wait for each other, so a thread pool of callers
// (the reference's: constraint_builder_2d.cc:97-111) gets four searches in flight whatever its
// size.  Sixteen unless the deployment says otherwise -- effective when this library is loaded
// before the runtime initialises (the variable is read once, at its first call).
struct HardwareQueueDefault {
  HardwareQueueDefault() { setenv("GPU_MAX_HW_QUEUES", "16", /*overwrite=*/0); }
} g_hardware_queue_default;

struct alignas(64) TurnCounter { std::atomic<unsigned> value{0}; };
TurnCounter g_turn_next, g_turn_serving;

struct CrashHandlerInstaller {
  CrashHandlerInstaller() {
    const char* env = getenv("CMX_SYNC");
    if (env && env[0] == '1') {
      signal(SIGSEGV, CrashHandler);
      signal(SIGABRT, CrashHandler);
    }
  }
} g_crash_handler_installer;

thread_local std::string g_last_error;
thread_local std::map<int, hipStream_t> g_stream_override;

struct Pool {
  std::mutex mu;
  std::map<int, std::vector<Workspace*>> free_list;
  std::map<int, std::vector<StreamSetLease::Set*>> free_stream_sets;
  ~Pool() {
    // Intentionally leak at process exit: the HIP runtime may already be
    // torn down when static destructors run.
  }
};
Pool& ThePool() {
  static Pool* pool = new Pool;
  return *pool;
}
}  // namespace

LaunchTurn::LaunchTurn() : held_(Debug().launch_serial == 0) {
  if (!held_) return;
  const unsigned mine = g_turn_next.value.fetch_add(1, std::memory_order_relaxed);
  int spins = 0;
  while (g_turn_serving.value.load(std::memory_order_acquire) != mine) {
    __builtin_ia32_pause();
    // (a holder that was descheduled: let it run)
    if (++spins > 20000) { std::this_thread::yield(); spins = 0; }
  }
}
LaunchTurn::~LaunchTurn() {
  if (held_) g_turn_serving.value.fetch_add(1, std::memory_order_release);
}

namespace {
}  // namespace

// ---------------------------------------------------------------------------
// ParallelFor: persistent host workers
// ---------------------------------------------------------------------------
namespace {
class HostPool {
 public:
  static HostPool& Get() {
    static HostPool* pool = new HostPool();     // leaked on purpose: workers outlive static dtors
    return *pool;
  }
  int workers() const { return num_workers_; }

  void Run(int n, const std::function<void(int)>& fn) {
    // One job at a time -- and a caller that finds the pool busy does NOT wait for it: it runs
    // its items itself, on its own thread.  (A job may be long: the two halves of an RT-2D batch
    // each run to their stream synchronisation inside one.  Waiting here serialised whole calls
    // of different host threads: four threads issuing 128-match batches measured exactly the
    // single-thread rate.)
    std::unique_lock<std::mutex> one_job(job_mutex_, std::try_to_lock);
    if (!one_job.owns_lock()) {
      for (int i = 0; i < n; ++i) fn(i);
      return;
    }
    Job job{&fn, n};
    job.chunk = std::max(1, n / (8 * (num_workers_ + 1)));
    job_.store(&job, std::memory_order_seq_cst);
    generation_.fetch_add(1, std::memory_order_seq_cst);
    {
      std::lock_guard<std::mutex> lk(sleep_mutex_);
      if (sleepers_ > 0) sleep_cv_.notify_all();
    }
    tls_inside_ = true;          // a ParallelFor inside fn runs inline on this thread as well
    Work(&job);
    tls_inside_ = false;
    // (items are microseconds of host work: a short spin, then the core is handed back between
    // polls instead of burning it for as long as the slowest worker takes)
    const auto wait_until = [](const auto& ready) {
      for (int spins = 0; !ready(); ++spins) {
        if (spins < 4096) __builtin_ia32_pause();
        else std::this_thread::yield();
      }
    };
    wait_until([&] { return job.done.load(std::memory_order_acquire) >= n; });
    // `job` lives on this stack: no worker may still hold it when we return.  A worker
    // announces itself (active_) BEFORE it reads job_, so either it reads null below or we
    // see it here (sequentially consistent on both sides).
    job_.store(nullptr, std::memory_order_seq_cst);
    wait_until([&] { return active_.load(std::memory_order_seq_cst) == 0; });
    if (job.error) std::rethrow_exception(job.error);
  }

  static thread_local bool tls_inside_;

 private:
  struct Job {
    const std::function<void(int)>* fn;
    int n;
    std::atomic<int> next{0}, done{0};
    std::mutex error_mutex;
    std::exception_ptr error;
    int chunk = 1;
    Job(const std::function<void(int)>* f, int count) : fn(f), n(count) {}
  };
  HostPool() {
    int want = 0;
    if (const char* e = getenv("CMX_HOST_THREADS")) want = atoi(e);
    if (want <= 0) want = std::min(16u, std::max(1u, std::thread::hardware_concurrency() / 2));
    num_workers_ = want - 1;
    if (const char* e = getenv("CMX_HOST_SPIN_US")) spin_us_ = std::max(0, atoi(e));
    for (int t = 0; t < num_workers_; ++t) std::thread([this] { WorkerLoop(); }).detach();
  }
  static void Work(Job* job) {
    // (indices are drawn `chunk` at a time: a thousand sub-microsecond items drawn one by one
    // spend their time on the two shared counters)
    const int chunk = job->chunk;
    for (;;) {
      const int begin = job->next.fetch_add(chunk, std::memory_order_relaxed);
      if (begin >= job->n) break;
      const int end = std::min(job->n, begin + chunk);
      for (int i = begin; i < end; ++i) {
        try {
          (*job->fn)(i);
        } catch (...) {
          std::lock_guard<std::mutex> lk(job->error_mutex);
          if (!job->error) job->error = std::current_exception();
        }
      }
      job->done.fetch_add(end - begin, std::memory_order_release);
    }
  }
  void WorkerLoop() {
    tls_inside_ = true;
    unsigned long long seen = 0;
    for (;;) {
      // spin for a while, then sleep until the next job
      const auto t0 = std::chrono::steady_clock::now();
      bool have = false;
      for (int spins = 0;; ++spins) {
        if (generation_.load(std::memory_order_acquire) != seen) { have = true; break; }
        __builtin_ia32_pause();
        if ((spins & 1023) == 1023 &&
            std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(spin_us_))
          break;
      }
      if (!have) {
        std::unique_lock<std::mutex> lk(sleep_mutex_);
        ++sleepers_;
        sleep_cv_.wait(lk, [&] { return generation_.load(std::memory_order_acquire) != seen; });
        --sleepers_;
      }
      seen = generation_.load(std::memory_order_acquire);
      active_.fetch_add(1, std::memory_order_seq_cst);
      if (Job* job = job_.load(std::memory_order_seq_cst)) Work(job);
      active_.fetch_sub(1, std::memory_order_seq_cst);
    }
  }

  int num_workers_ = 0;
  int spin_us_ = 500;          // how long an idle worker spins before it sleeps
  std::mutex job_mutex_, sleep_mutex_;
  std::condition_variable sleep_cv_;
  int sleepers_ = 0;
  std::atomic<unsigned long long> generation_{0};
  std::atomic<Job*> job_{nullptr};
  std::atomic<int> active_{0};
};
thread_local bool HostPool::tls_inside_ = false;
}  // namespace

void ParallelFor(int n, int serial_below, const std::function<void(int)>& fn) {
  if (n <= 0) return;
  if (n < serial_below || HostPool::tls_inside_) {
    for (int i = 0; i < n; ++i) fn(i);
    return;
  }
  HostPool& pool = HostPool::Get();
  if (pool.workers() == 0) {
    for (int i = 0; i < n; ++i) fn(i);
    return;
  }
  pool.Run(n, fn);
}

// ---------------------------------------------------------------------------
// HostLane: one helper thread
// ---------------------------------------------------------------------------
namespace {
struct LaneState {
  std::mutex lease;                          // who may post jobs
  std::mutex sleep_mutex;
  std::condition_variable sleep_cv;
  bool sleeping = false;
  std::atomic<int> state{0};                 // 0: idle, 1: a job is posted, 2: it is done
  std::function<void()> job;
  bool started = false;
  void Loop() {
    for (;;) {
      const auto t0 = std::chrono::steady_clock::now();
      bool have = false;
      for (int spins = 0;; ++spins) {
        if (state.load(std::memory_order_acquire) == 1) { have = true; break; }
        __builtin_ia32_pause();
        if ((spins & 1023) == 1023 &&
            std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(300))
          break;
      }
      if (!have) {
        std::unique_lock<std::mutex> lk(sleep_mutex);
        sleeping = true;
        sleep_cv.wait(lk, [&] { return state.load(std::memory_order_acquire) == 1; });
        sleeping = false;
      }
      job();
      job = nullptr;
      state.store(2, std::memory_order_release);
    }
  }
};
LaneState& TheLane() {
  static LaneState* lane = new LaneState;    // leaked on purpose, like the pool
  return *lane;
}
}  // namespace

HostLane::HostLane() {
  LaneState& L = TheLane();
  own_ = L.lease.try_lock();
  if (own_ && !L.started) {
    L.started = true;
    std::thread([&L] { L.Loop(); }).detach();
  }
}
HostLane::~HostLane() {
  if (!own_) return;
  Wait();
  TheLane().lease.unlock();
}
void HostLane::Run(std::function<void()> fn) {
  if (!own_) { fn(); return; }
  Wait();
  LaneState& L = TheLane();
  L.job = std::move(fn);
  pending_ = true;
  L.state.store(1, std::memory_order_release);
  std::lock_guard<std::mutex> lk(L.sleep_mutex);
  if (L.sleeping) L.sleep_cv.notify_one();
}
void HostLane::Wait() {
  if (!own_ || !pending_) return;
  LaneState& L = TheLane();
  for (int spins = 0; L.state.load(std::memory_order_acquire) != 2; ++spins) {
    if (spins < 4096) __builtin_ia32_pause();
    else std::this_thread::yield();
  }
  L.state.store(0, std::memory_order_relaxed);
  pending_ = false;
}

// ---------------------------------------------------------------------------
// Debug switches (cmx_debug_set)
// ---------------------------------------------------------------------------
namespace {
DebugOptions g_debug;       // (plain ints written by cmx_debug_set before the calls they affect)
}
const DebugOptions& Debug() { return g_debug; }
bool DebugSet(const char* name, int value) {
  if (name == nullptr) return false;
#define CMX_DEBUG_SET(field) \
  if (std::strcmp(name, #field) == 0) { g_debug.field = value; return true; }
  CMX_DEBUG_OPTIONS(CMX_DEBUG_SET)
#undef CMX_DEBUG_SET
  return false;
}

void SetLastError(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
}
const char* LastError() { return g_last_error.c_str(); }

void UseDevice(int device) {
  int count = 0;
  hipError_t err = hipGetDeviceCount(&count);
  if (err != hipSuccess || count <= 0) {
    SetLastError("no HIP device available (%s); this library has no CPU fallback",
                 err == hipSuccess ? "device count is 0" : hipGetErrorString(err));
    throw HipError{CMX_DEVICE_ERROR};
  }
  CMX_REQUIRE(device >= 0 && device < count, "device %d out of range [0,%d)", device, count);
  CMX_HIP(hipSetDevice(device));
}

hipStream_t OverrideStream(int device) {
  auto it = g_stream_override.find(device);
  return it == g_stream_override.end() ? nullptr : it->second;
}

namespace {
__global__ void __launch_bounds__(1024)
SmallCopyKernel(uint4* __restrict__ dst, const uint4* __restrict__ src, size_t words16,
                unsigned char* __restrict__ dst_tail, const unsigned char* __restrict__ src_tail,
                int tail_bytes, uint4* __restrict__ zero, size_t zero_words16) {
  // (several workgroups for transfers beyond 64 KB: each takes a contiguous share)
  const size_t per_block = (words16 + gridDim.x - 1) / gridDim.x;
  const size_t begin = blockIdx.x * per_block, end = min(words16, begin + per_block);
  for (size_t i = begin + threadIdx.x; i < end; i += blockDim.x) dst[i] = src[i];
  // (a region the caller wants zeroed on the device: zeros that do not cross the bus)
  const size_t zper = (zero_words16 + gridDim.x - 1) / gridDim.x;
  const size_t zbegin = blockIdx.x * zper, zend = min(zero_words16, zbegin + zper);
  for (size_t i = zbegin + threadIdx.x; i < zend; i += blockDim.x) zero[i] = make_uint4(0, 0, 0, 0);
  if (blockIdx.x == 0 && static_cast<int>(threadIdx.x) < tail_bytes)
    dst_tail[threadIdx.x] = src_tail[threadIdx.x];
}
}  // namespace

void SmallCopyAsync(void* dst, const void* src, size_t bytes, bool to_device, hipStream_t stream,
                    void* zero, size_t zero_bytes) {
  const bool enabled = Debug().no_copy_kernels == 0;
  const bool aligned = (reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src) |
                        reinterpret_cast<uintptr_t>(zero) | zero_bytes) % 16 == 0;
  if (!enabled || !aligned || bytes == 0 || bytes > kCopyKernelMaxBytes) {
    CMX_HIP(hipMemcpyAsync(dst, src, bytes, to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost,
                           stream));
    if (zero_bytes) CMX_HIP(hipMemsetAsync(zero, 0, zero_bytes, stream));
    return;
  }
  const size_t words16 = bytes / 16;
  const int tail = static_cast<int>(bytes - words16 * 16);
  SmallCopyKernel<<<static_cast<unsigned>((bytes + 65535) / 65536), 1024, 0, stream>>>(
      static_cast<uint4*>(dst), static_cast<const uint4*>(src), words16,
      static_cast<unsigned char*>(dst) + words16 * 16,
      static_cast<const unsigned char*>(src) + words16 * 16, tail, static_cast<uint4*>(zero), zero_bytes / 16);
  CMX_HIP(hipGetLastError());
}

// Opt-in to more than 64 KB of dynamic LDS, once per (device, kernel): HIP keeps function
// attributes per device.
void OptInLds(const void* fn, int device, size_t bytes) {
  struct Seen { const void* fn; int device; size_t bytes; };
  static std::mutex mu;
  static std::vector<Seen>* seen = new std::vector<Seen>;
  std::lock_guard<std::mutex> lock(mu);
  for (Seen& s : *seen) {
    if (s.fn == fn && s.device == device) {
      if (s.bytes >= bytes) return;
      CMX_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes)));
      s.bytes = bytes;
      return;
    }
  }
  CMX_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes)));
  seen->push_back(Seen{fn, device, bytes});
}

Workspace::~Workspace() {
  if (ev_begin) (void)hipEventDestroy(ev_begin);
  if (ev_end) (void)hipEventDestroy(ev_end);
  if (ev_k0) (void)hipEventDestroy(ev_k0);
  if (ev_k1) (void)hipEventDestroy(ev_k1);
  if (ev_x0) (void)hipEventDestroy(ev_x0);
  if (ev_x1) (void)hipEventDestroy(ev_x1);
  if (own_stream) (void)hipStreamDestroy(own_stream);
}

StageTrace::StageTrace(hipStream_t stream) : stream_(stream) {
  enabled_ = Debug().trace != 0;
  if (enabled_) Mark("begin");
}
StageTrace::~StageTrace() {
  for (auto& m : marks_) (void)hipEventDestroy(m.second);
}
void StageTrace::Mark(const char* name) {
  {
    const bool sync_debug = Debug().sync != 0;
    if (sync_debug) {   // debugging aid: localise a faulting kernel
      fprintf(stderr, "[cmx sync] %s ...\n", name);
      (void)hipStreamSynchronize(stream_);
    }
  }
  if (!enabled_) return;
  hipEvent_t ev;
  if (hipEventCreate(&ev) != hipSuccess) return;
  (void)hipEventRecord(ev, stream_);
  marks_.emplace_back(name, ev);
}
void StageTrace::Report() {
  if (!enabled_ || marks_.size() < 2) return;
  float total = 0.f;
  (void)hipEventElapsedTime(&total, marks_.front().second, marks_.back().second);
  fprintf(stderr, "[cmx trace] total %.1f us:", total * 1e3f);
  for (size_t i = 1; i < marks_.size(); ++i) {
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, marks_[i - 1].second, marks_[i].second);
    fprintf(stderr, " %s=%.1f", marks_[i].first.c_str(), ms * 1e3f);
  }
  fprintf(stderr, "\n");
}

bool TimelineEnabled() { return Debug().timeline != 0; }

void ReportTimeline(const char* name, const unsigned long long* device, int blocks,
                    hipStream_t stream) {
  constexpr int K = 16;
  std::vector<unsigned long long> h(static_cast<size_t>(blocks) * K);
  if (hipMemcpyAsync(h.data(), device, h.size() * sizeof(unsigned long long),
                     hipMemcpyDeviceToHost, stream) != hipSuccess ||
      hipStreamSynchronize(stream) != hipSuccess)
    return;
  unsigned long long first = ~0ull, last = 0;
  int used = 0;
  for (int b = 0; b < blocks; ++b) {
    if (!h[static_cast<size_t>(b) * K]) continue;
    ++used;
    first = std::min(first, h[static_cast<size_t>(b) * K]);
    for (int k = 0; k < 14; ++k) last = std::max(last, h[static_cast<size_t>(b) * K + k]);
  }
  if (!used) return;
  // (tools: CMX_TIMELINE_DUMP=<file> appends the raw stamps, a line per block -- slots 14 and 15
  // carry what the kernel put there instead of a time: the work item and the hardware id)
  if (const char* path = getenv("CMX_TIMELINE_DUMP")) {
    if (FILE* f = fopen(path, "a")) {
      fprintf(f, "# %s %d\n", name, blocks);
      for (int b = 0; b < blocks; ++b) {
        if (!h[static_cast<size_t>(b) * K]) continue;
        fprintf(f, "%d", b);
        for (int k = 0; k < K; ++k) fprintf(f, " %llu", h[static_cast<size_t>(b) * K + k] - (k < 14 && h[static_cast<size_t>(b) * K + k] ? first : 0));
        fprintf(f, "\n");
      }
      fclose(f);
    }
  }
  fprintf(stderr, "[cmx timeline] %s: %d blocks, launch span %.2f us; stamp: median / max us after "
                  "the block's start (start: median / max after the first block's):\n", name, used,
          (last - first) * 0.01);
  for (int k = 0; k < 14; ++k) {
    std::vector<double> d;
    for (int b = 0; b < blocks; ++b) {
      const unsigned long long t0 = h[static_cast<size_t>(b) * K], t = h[static_cast<size_t>(b) * K + k];
      if (!t0 || !t) continue;
      d.push_back(k == 0 ? (t0 - first) * 0.01 : (t - t0) * 0.01);
    }
    if (d.empty()) continue;
    std::sort(d.begin(), d.end());
    fprintf(stderr, "   [%2d] %7.2f / %7.2f  (%zu blocks)\n", k, d[d.size() / 2], d.back(), d.size());
  }
}

WorkspaceLease::WorkspaceLease(int device) : ws_(nullptr) {
  UseDevice(device);
  Pool& pool = ThePool();
  {
    std::lock_guard<std::mutex> lock(pool.mu);
    auto& list = pool.free_list[device];
    if (!list.empty()) {
      ws_ = list.back();
      list.pop_back();
    }
  }
  if (!ws_) {
    std::unique_ptr<Workspace> ws(new Workspace);
    ws->device = device;
    CMX_HIP(hipStreamCreateWithFlags(&ws->own_stream, hipStreamNonBlocking));
    CMX_HIP(hipEventCreate(&ws->ev_begin));
    CMX_HIP(hipEventCreate(&ws->ev_end));
    CMX_HIP(hipEventCreate(&ws->ev_k0));
    CMX_HIP(hipEventCreate(&ws->ev_k1));
    CMX_HIP(hipEventCreate(&ws->ev_x0));
    CMX_HIP(hipEventCreate(&ws->ev_x1));
    ws_ = ws.release();
  }
  hipStream_t over = OverrideStream(device);
  ws_->stream = over ? over : ws_->own_stream;
}

StreamSetLease::StreamSetLease(int device) : set_(nullptr) {
  UseDevice(device);
  Pool& pool = ThePool();
  {
    std::lock_guard<std::mutex> lock(pool.mu);
    auto& list = pool.free_stream_sets[device];
    if (!list.empty()) {
      set_ = list.back();
      list.pop_back();
    }
  }
  if (!set_) {
    std::unique_ptr<Set> set(new Set);
    set->device = device;
    for (int k = 0; k < kStreams; ++k) set->s[k] = nullptr;
    // (under the pool's lock: no other stream of this library is created in between)
    std::lock_guard<std::mutex> lock(pool.mu);
    for (int k = 0; k < kStreams; ++k)
      CMX_HIP(hipStreamCreateWithFlags(&set->s[k], hipStreamNonBlocking));
    set_ = set.release();       // (leaked with the pool at process exit, like the workspaces)
  }
}

StreamSetLease::~StreamSetLease() {
  Pool& pool = ThePool();
  std::lock_guard<std::mutex> lock(pool.mu);
  pool.free_stream_sets[set_->device].push_back(set_);
}

WorkspaceLease::~WorkspaceLease() {
  Pool& pool = ThePool();
  std::lock_guard<std::mutex> lock(pool.mu);
  pool.free_list[ws_->device].push_back(ws_);
}

}  // namespace cmx

extern "C" {

const char* cmx_version(void) { return "cartographer_mi355x 0.2 (gfx950)"; }
int32_t cmx_sizeof_match_stats(void) { return static_cast<int32_t>(sizeof(cmx_match_stats)); }

cmx_status cmx_debug_set(const char* name, int32_t value) {
  return cmx::Guard([&] {
    CMX_REQUIRE(cmx::DebugSet(name, value), "unknown debug switch '%s'", name ? name : "(null)");
  });
}
void cmx_debug_reset(void) {
#define CMX_DEBUG_RESET(field) (void)cmx::DebugSet(#field, 0);
  CMX_DEBUG_OPTIONS(CMX_DEBUG_RESET)
#undef CMX_DEBUG_RESET
}

const char* cmx_status_string(cmx_status s) {
  switch (s) {
    case CMX_OK: return "CMX_OK";
    case CMX_INVALID_ARGUMENT: return "CMX_INVALID_ARGUMENT";
    case CMX_DEVICE_ERROR: return "CMX_DEVICE_ERROR";
    case CMX_OUT_OF_MEMORY: return "CMX_OUT_OF_MEMORY";
    case CMX_UNSUPPORTED: return "CMX_UNSUPPORTED";
  }
  return "CMX_?";
}

const char* cmx_last_error(void) { return cmx::LastError(); }

int32_t cmx_device_count(void) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess) return 0;
  return count;
}

cmx_status cmx_set_stream(int32_t device, void* hip_stream) {
  return cmx::Guard([&] {
    cmx::UseDevice(device);
    if (hip_stream) {
      cmx::g_stream_override[device] = static_cast<hipStream_t>(hip_stream);
    } else {
      cmx::g_stream_override.erase(device);
    }
  });
}

}  // extern "C"
