// Host plumbing: error text, device selection, workspace pool, library entry
// points that are not tied to one matcher.
#include "cmx_common.h"

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
                int tail_bytes) {
  for (size_t i = threadIdx.x; i < words16; i += blockDim.x) dst[i] = src[i];
  if (static_cast<int>(threadIdx.x) < tail_bytes) dst_tail[threadIdx.x] = src_tail[threadIdx.x];
}
}  // namespace

void SmallCopyAsync(void* dst, const void* src, size_t bytes, bool to_device, hipStream_t stream) {
  static const bool enabled = [] {
    const char* e = getenv("CMX_COPY_KERNELS");
    return e ? e[0] != '0' : true;
  }();
  const bool aligned = (reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) % 16 == 0;
  if (!enabled || !aligned || bytes == 0 || bytes > kCopyKernelMaxBytes) {
    CMX_HIP(hipMemcpyAsync(dst, src, bytes, to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost,
                           stream));
    return;
  }
  const size_t words16 = bytes / 16;
  const int tail = static_cast<int>(bytes - words16 * 16);
  SmallCopyKernel<<<1, 1024, 0, stream>>>(
      static_cast<uint4*>(dst), static_cast<const uint4*>(src), words16,
      static_cast<unsigned char*>(dst) + words16 * 16,
      static_cast<const unsigned char*>(src) + words16 * 16, tail);
  CMX_HIP(hipGetLastError());
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
  const char* env = getenv("CMX_TRACE");
  enabled_ = env && env[0] == '1';
  if (enabled_) Mark("begin");
}
StageTrace::~StageTrace() {
  for (auto& m : marks_) (void)hipEventDestroy(m.second);
}
void StageTrace::Mark(const char* name) {
  {
    static const bool sync_debug = [] {
      const char* env = getenv("CMX_SYNC");
      return env && env[0] == '1';
    }();
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

bool TimelineEnabled() {
  static const bool enabled = [] {
    const char* env = getenv("CMX_TIMELINE");
    return env && env[0] == '1';
  }();
  return enabled;
}

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
    for (int k = 0; k < K; ++k) last = std::max(last, h[static_cast<size_t>(b) * K + k]);
  }
  if (!used) return;
  fprintf(stderr, "[cmx timeline] %s: %d blocks, launch span %.2f us; stamp: median / max us after "
                  "the block's start (start: median / max after the first block's):\n", name, used,
          (last - first) * 0.01);
  for (int k = 0; k < K; ++k) {
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

WorkspaceLease::~WorkspaceLease() {
  Pool& pool = ThePool();
  std::lock_guard<std::mutex> lock(pool.mu);
  pool.free_list[ws_->device].push_back(ws_);
}

}  // namespace cmx

extern "C" {

const char* cmx_version(void) { return "cartographer_mi355x 0.1 (gfx950)"; }

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
