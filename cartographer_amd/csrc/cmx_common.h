// Shared host-side plumbing of libcartographer_mi355x: error reporting,
// device buffers, per-call workspaces (stream + scratch + pinned staging).
#ifndef CMX_COMMON_H_
#define CMX_COMMON_H_

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/cartographer_mi355x.h"

namespace cmx {

void SetLastError(const char* fmt, ...);
const char* LastError();

struct HipError {
  cmx_status status;
};

// Throws HipError on failure; caught at the C boundary.
#define CMX_HIP(expr)                                                              \
  do {                                                                             \
    hipError_t cmx_err__ = (expr);                                                 \
    if (cmx_err__ != hipSuccess) {                                                 \
      ::cmx::SetLastError("%s failed: %s (%s:%d)", #expr,                          \
                          hipGetErrorString(cmx_err__), __FILE__, __LINE__);       \
      throw ::cmx::HipError{cmx_err__ == hipErrorOutOfMemory ? CMX_OUT_OF_MEMORY   \
                                                             : CMX_DEVICE_ERROR};  \
    }                                                                              \
  } while (0)

#define CMX_REQUIRE(cond, ...)                       \
  do {                                               \
    if (!(cond)) {                                   \
      ::cmx::SetLastError(__VA_ARGS__);              \
      throw ::cmx::HipError{CMX_INVALID_ARGUMENT};   \
    }                                                \
  } while (0)

// Runs `body`, translating exceptions into a cmx_status.
template <typename F>
cmx_status Guard(F&& body) {
  try {
    body();
    return CMX_OK;
  } catch (const HipError& e) {
    return e.status;
  } catch (const std::bad_alloc&) {
    SetLastError("host allocation failed");
    return CMX_OUT_OF_MEMORY;
  } catch (const std::exception& e) {
    SetLastError("exception: %s", e.what());
    return CMX_DEVICE_ERROR;
  }
}

// Validates `device` (fails loudly without a GPU) and makes it current.
void UseDevice(int device);

// Host-side parallel loop for the per-item planning of a batch call (range scans, libm rotation
// tables, staging copies: ~2 us per item, serial host time that used to exceed the device time
// of a 128-match batch several times over).  fn(i) runs for i in [0, n) on the calling thread
// and on a small pool of persistent workers (CMX_HOST_THREADS, default min(16, cores / 2));
// workers spin for 500 us (CMX_HOST_SPIN_US) after a job before they sleep, so back-to-back calls pay no wake-up.
// n below `serial_below` runs inline.  fn must not throw across threads: the first exception
// is captured and rethrown on the caller.  Nested calls run inline.
void ParallelFor(int n, int serial_below, const std::function<void(int)>& fn);

// Grow-only device buffer.
class DeviceBuffer {
 public:
  DeviceBuffer() = default;
  DeviceBuffer(const DeviceBuffer&) = delete;
  DeviceBuffer& operator=(const DeviceBuffer&) = delete;
  ~DeviceBuffer() { Free(); }
  void* Reserve(size_t bytes) {
    if (bytes > capacity_) {
      Free();
      const size_t want = bytes + bytes / 4 + 256;
      CMX_HIP(hipMalloc(&ptr_, want));
      capacity_ = want;
    }
    return ptr_;
  }
  template <typename T>
  T* ReserveAs(size_t count) { return static_cast<T*>(Reserve(count * sizeof(T))); }
  void* get() const { return ptr_; }
  size_t capacity() const { return capacity_; }
  void Free() {
    if (ptr_) (void)hipFree(ptr_);
    ptr_ = nullptr;
    capacity_ = 0;
  }
 private:
  void* ptr_ = nullptr;
  size_t capacity_ = 0;
};

class PinnedBuffer {
 public:
  PinnedBuffer() = default;
  PinnedBuffer(const PinnedBuffer&) = delete;
  PinnedBuffer& operator=(const PinnedBuffer&) = delete;
  ~PinnedBuffer() { if (ptr_) (void)hipHostFree(ptr_); }
  void* Reserve(size_t bytes) {
    if (bytes > capacity_) {
      if (ptr_) (void)hipHostFree(ptr_);
      ptr_ = nullptr;
      const size_t want = bytes + bytes / 4 + 256;
      CMX_HIP(hipHostMalloc(&ptr_, want, hipHostMallocDefault));
      capacity_ = want;
    }
    return ptr_;
  }
  template <typename T>
  T* ReserveAs(size_t count) { return static_cast<T*>(Reserve(count * sizeof(T))); }
 private:
  void* ptr_ = nullptr;
  size_t capacity_ = 0;
};

// Small transfers between pinned host memory (mapped into the device's address space) and
// device memory by a one-workgroup kernel instead of a copy command: a latency-bound call is a
// chain of launches on one stream, and a copy command in that chain costs 10-15 us of engine
// start-up where a launch costs 3-4.  Falls back to hipMemcpyAsync above `kCopyKernelMaxBytes`
// or with CMX_COPY_KERNELS=0.  `pinned` is the host side (source for to_device, else target).
constexpr size_t kCopyKernelMaxBytes = 1024 * 1024;   // one workgroup per 64 KB
void SmallCopyAsync(void* dst, const void* src, size_t bytes, bool to_device, hipStream_t stream);

// Everything one in-flight call needs; handed out by a per-device pool so
// concurrent callers never share scratch.
struct Workspace {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;  // own_stream or the caller's (cmx_set_stream)
  hipEvent_t ev_begin = nullptr, ev_end = nullptr;
  hipEvent_t ev_k0 = nullptr, ev_k1 = nullptr;  // dominant-kernel bracket
  hipEvent_t ev_x0 = nullptr, ev_x1 = nullptr;  // branch-and-bound expansion bracket
  static constexpr int kNumBuffers = 24;
  DeviceBuffer dev[kNumBuffers];
  PinnedBuffer pinned[4];
  ~Workspace();
};

// Optional per-stage timing (environment CMX_TRACE=1): an event after every
// stage, durations printed to stderr when the call has synchronised.
class StageTrace {
 public:
  explicit StageTrace(hipStream_t stream);
  ~StageTrace();
  void Mark(const char* name);
  void Report();   // call after the stream has been synchronised
  bool enabled() const { return enabled_; }
 private:
  bool enabled_;
  hipStream_t stream_;
  std::vector<std::pair<std::string, hipEvent_t>> marks_;
};

class WorkspaceLease {
 public:
  explicit WorkspaceLease(int device);
  ~WorkspaceLease();
  Workspace* operator->() { return ws_; }
  Workspace& operator*() { return *ws_; }
 private:
  Workspace* ws_;
};

// In-kernel timelines (CMX_TIMELINE=1): see cmx_device.h Stamp().  `Enabled` is read once.
bool TimelineEnabled();
// Prints, per stamp k, the median / max over blocks of (t_k - t_0) and the span of the whole
// launch (first t_0 to last stamp) in microseconds; `device` holds blocks x 16 stamps.
void ReportTimeline(const char* name, const unsigned long long* device, int blocks,
                    hipStream_t stream);

// Thread-local stream override (cmx_set_stream).
hipStream_t OverrideStream(int device);

inline int DivUp(int64_t a, int64_t b) { return static_cast<int>((a + b - 1) / b); }

}  // namespace cmx

#endif  // CMX_COMMON_H_
