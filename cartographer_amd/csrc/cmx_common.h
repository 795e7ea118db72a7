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

// Test and tool switches (cmx_debug_set, include/cartographer_mi355x_debug.h): which of two
// equivalent paths runs, verification modes, tuning overrides.  All zero in production; the
// product reads no environment variable on a call path.  `Debug()` is a relaxed snapshot: set
// the switches before the calls they should affect.
#define CMX_DEBUG_OPTIONS(X)                                                                      \
  X(rt2d_legacy)          /* 1: real-time 2D on the one-thread-per-candidate kernels only */       \
  X(rt2d_tile)            /* tile core size override (multiple of 8) */                            \
  X(rt2d_groups)          /* most rotation groups (work items) per tile */                         \
  X(rt2d_target)          /* entries per work item the planner aims at */                          \
  X(rt2d_lds_kb)          /* LDS budget of a tile workgroup in KB */                               \
  X(rt2d_no_image_cache)  /* 1: grid images are always built into scratch of the call */           \
  X(rt2d_parts)           /* parts a large batch is issued in (0: default) */                      \
  X(rt2d_host_par)        /* matches per call from which the per-match host loops go to the host pool (0: 4096) */ \
  X(rt2d_tables_serial)   /* 1: the rotation tables of a part by one thread (no host pool) */ \
  X(rt2d_first_part)      /* matches of a large batch's first part (0: default; -1: none, decreasing parts only) */ \
  X(rt2d_image_kernels)   /* 1: a grid's derived images by the three kernels of rounds 4 - 5 (parity partner of Rt2DImageKernel) */ \
  X(rt2d_no_lane)         /* 1: the parts of a batch are prepared by the calling thread itself (no helper lane) */ \
  X(rt2d_parts_pool)      /* 1: the parts of a batch are planned and enqueued by host pool threads */ \
  X(rt2d_grid_share)      /* a part's tile grid and work items sized for its share of the CUs: 1 always, 2 never (0: by batch size) */ \
  X(rt2d_unfused)         /* 1: one-tile matches through the prep kernel too (parity partner) */   \
  X(rt2d_no_bounds)       /* 1: no block bounds, the tile kernel sums every candidate (parity partner) */ \
  X(rt2d_bounds)          /* 1: block bounds for calls of any size (default: from 192 matches per call on) */ \
  X(rt2d_bounds_fused)    /* 1: the bound kernel finishes its matches itself (no tail kernel: round 5's shape) */ \
  X(rt2d_bounds_level)    /* 2: blocks of 2 x 2 translations as the first level (default: 4 x 4, refined through 2 x 2) */ \
  X(rt2d_bounds_verify)   /* 1: every block is summed and checked against its bound (an error if one is below) */ \
  X(timeline)             /* 1: in-kernel timelines (cmx_device.h Stamp) reported on stderr */     \
  X(trace)                /* 1: an event after every stage of a call, durations on stderr */       \
  X(host_trace)           /* 1: wall clock of the host phases of a call on stderr */               \
  X(launch_serial)        /* 1: callers issue their launches without taking turns (LaunchTurn off) */ \
  X(sync)                 /* 1: synchronise after every stage (localises a faulting kernel) */     \
  X(no_copy_kernels)      /* 1: small transfers by copy commands instead of a copy kernel */       \
  X(timing)               /* 1: HIP-event brackets around the device work (cmx_match_stats *_ms; else 0) */ \
  X(no_direct_results)    /* 1: results through a copy kernel, not stored to pinned memory by the last kernel */ \
  X(frontier_capacity)    /* nodes per frontier / leaf buffer (tests: forces the overflow path) */ \
  X(comm_virtual_ranks)   /* N > 1: a one-device cmx_comm becomes N ranks on it (host-side key reduction) */ \
  X(comm_force_rccl)      /* 1: a one-device cmx_comm is built by RCCL too, its all-reduce is RCCL's */ \
  X(fast2d_unfused)       /* 1: fast 2D front end as separate prep / score launches */             \
  X(fast2d_store_scans)   /* 1: never keep the surviving scans' cells, 2: always */                \
  X(fast2d_fused_threads) /* threads per block of the fused front end */                           \
  X(fast2d_fanout)        /* batches of at least this many problems run as independent single searches over the host pool (0: from 32 on, 1: never) */ \
  X(fast2d_group)         /* group bounds of the fused front end (three rotations, one sum over the dilated level): 1 never, 2 at any depth > 1 (0: from depth 5 on) */ \
  X(fast2d_group_verify)  /* 1: every group bound checked against the exact sums of its rotations on the device; 2: every unit treated as if its premise had failed (outer rotations unbounded); 3: both */ \
  X(fast2d_levels_per_stage) /* levels per depth-first stage */                                   \
  X(fast2d_wave_levels)   /* k > 0: k - 1 levels of wave-per-node expansion */                     \
  X(fast2d_xcd_affinity)  /* 1: nodes of a problem on any XCD, 2: on one */                        \
  X(fast2d_queue)         /* 2: tree search by the level-synchronous launches (no work queue), 1: the queue for batches too */ \
  X(fast2d_queue_blocks)  /* workgroups of the work-queue tree search (0: default) */              \
  X(fast2d_queue_capacity) /* nodes per sub-queue (tests: forces the overflow path) */             \
  X(fast2d_queue_lost)    /* lost races after which a wavefront stops looking for work (0: 3) */   \
  X(filters_generic)      /* 1: voxel filters of small clouds through the multi-launch path too */ \
  X(fast3d_byte_loads)    /* 1: every child cell with its own byte load */                         \
  X(fast3d_affinity)      /* 1: nodes of a problem on any XCD, 2: on one */                        \
  X(fast3d_no_families)   /* 1: one node per block in the 3D expansion */                          \
  X(fast3d_no_oct)        /* 1: no oct words (byte levels only) */                                 \
  X(fast3d_batch)         /* pairs per chain of launches */                                        \
  X(rt3d_legacy)          /* 1: real-time 3D on the exhaustive per-candidate kernel only */        \
  X(rt3d_no_tiles)        /* 1: bulk passes by memory gathers instead of LDS tiles */              \
  X(rt3d_verify)          /* 1: every bound checked on the device against what it bounds */        \
  X(rt3d_no_rotblocks)    /* 1: the dense group pass (no rotation-block level above it) */          \
  X(rt3d_rotblock_permille) /* first group round: pairs of blocks within this of the best (0: 970) */ \
  X(rt3d_crosscheck)      /* 1: tiled passes next to the gather kernels, every sum compared */     \
  X(rt3d_expand_all)      /* 1: every group expanded (with rt3d_verify: every group bound checked) */ \
  X(rt3d_unstaged)        /* 1: second candidate round in one piece */                             \
  X(rt3d_no_boxes)        /* 1: chunk boxes computed inside the tile kernels */                    \
  X(rt3d_report)          /* 1: pass-by-pass report on stderr */                                   \
  X(rt3d_segments)        /* segment schedule override (percent | percent << 8) */                 \
  X(rt3d_group_rotations) /* rotations per group-pass workgroup */                                 \
  X(rt3d_group_tile_kb)   /* LDS tile capacity of the group pass */                                \
  X(rt3d_group_float)     /* 1: group centres in f32 instead of fixed point */                     \
  X(rt3d_cand_threads)    /* threads per candidate-pass workgroup */                               \
  X(rt3d_cand_rotations)  /* rotations per candidate work list */                                  \
  X(rt3d_cand_tile_kb)    /* LDS tile capacity of the candidate pass */
struct DebugOptions {
#define CMX_DEBUG_FIELD(name) int name = 0;
  CMX_DEBUG_OPTIONS(CMX_DEBUG_FIELD)
#undef CMX_DEBUG_FIELD
};
const DebugOptions& Debug();
// name = a field of DebugOptions; false: unknown name.
bool DebugSet(const char* name, int value);

// Validates `device` (fails loudly without a GPU) and makes it current.
void UseDevice(int device);

// Host-side parallel loop for the per-item planning of a batch call (range scans, libm rotation
// tables, staging copies: ~2 us per item, serial host time that used to exceed the device time
// of a 128-match batch several times over).  fn(i) runs for i in [0, n) on the calling thread
// and on a small pool of persistent workers (CMX_HOST_THREADS, default min(16, cores / 2));
// workers spin for 500 us (CMX_HOST_SPIN_US) after a job before they sleep, so back-to-back calls pay no wake-up.
// n below `serial_below` runs inline.  fn must not throw across threads: the first exception
// is captured and rethrown on the caller.  Nested calls run inline.  The workers are detached
// threads of the process that first used the pool: after fork() the child has none (POSIX keeps
// only the forking thread).  Its loops still complete -- the caller draws items like any worker
// and then takes all of them -- unless the fork happened in the middle of a job of another
// thread (the copied `active_` count never drops): fork before the first call, or exec.
void ParallelFor(int n, int serial_below, const std::function<void(int)>& fn);

// ONE helper thread next to a caller (round 6): a call that goes out in parts has the next part's
// host-only preparation (search parameters, plan) run here while the calling thread uploads and
// launches the current one.  A lease is exclusive -- a caller that finds the lane taken runs its
// jobs itself (Run() inline) -- and a job is a few tens of microseconds: the helper spins for a
// while after its last job, then sleeps.
// The runtime's launch path does not scale with the number of calling threads: sixteen threads
// that each push a chain of five EMPTY launches and synchronise complete a chain every 15 us,
// four threads every 5 - 12 (tools/probes/dispatch_rate.hip, launch_rate.hip) -- threads that meet
// inside the runtime put each other to sleep.  Callers of the latency chains (a fast-2D search:
// upload, front end, dive, filter, tree) therefore take TURNS at issuing: a ticket lock of our own,
// spinning (a turn is a few microseconds), so that the runtime sees one caller at a time.
class LaunchTurn {
 public:
  LaunchTurn();
  ~LaunchTurn();
  LaunchTurn(const LaunchTurn&) = delete;
  LaunchTurn& operator=(const LaunchTurn&) = delete;
 private:
  bool held_;
};

class HostLane {
 public:
  HostLane();                        // takes the process-wide lane if it is free
  ~HostLane();                       // (waits for a job in flight)
  HostLane(const HostLane&) = delete;
  HostLane& operator=(const HostLane&) = delete;
  // Starts fn on the helper (or runs it here and now when this lease did not get the lane).
  // One job at a time: Wait() before the next Run().
  void Run(std::function<void()> fn);
  void Wait();
  bool own() const { return own_; }

 private:
  bool own_ = false, pending_ = false;
};

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
// or with the debug switch no_copy_kernels.  `pinned` is the host side (source for to_device, else target).
constexpr size_t kCopyKernelMaxBytes = 1024 * 1024;   // one workgroup per 64 KB
// (`zero`, `zero_bytes`: a device region, 16-byte aligned and sized, the same launch fills with zeros)
void SmallCopyAsync(void* dst, const void* src, size_t bytes, bool to_device, hipStream_t stream,
                    void* zero = nullptr, size_t zero_bytes = 0);

// Opt-in to `bytes` (> 64 KB) of dynamic LDS for kernel `fn` on `device`: HIP keeps function
// attributes per device, so the cache behind this is keyed by (device, kernel), not by the thread
// or the size alone (a host thread that matched on device 0 and then on device 1 would
// otherwise skip the opt-in on the second device).
void OptInLds(const void* fn, int device, size_t bytes);

// Scratch whose words carry the tag ("epoch") of the call that wrote them -- work queues whose
// slots are polled by tag, so that nothing has to be cleared per call: zeroed when it is
// (re)allocated, the epoch counts the calls that have used it (never 0).
struct TaggedBuffer {
  DeviceBuffer buffer;
  unsigned epoch = 0;
  void* Acquire(size_t bytes, hipStream_t stream, unsigned* epoch_out) {
    const bool grew = bytes > buffer.capacity();
    void* p = buffer.Reserve(bytes);
    if (grew || epoch == 0xffffffffu) {
      CMX_HIP(hipMemsetAsync(p, 0, buffer.capacity(), stream));
      epoch = 0;
    }
    *epoch_out = ++epoch;
    return p;
  }
};

// Everything one in-flight call needs; handed out by a per-device pool so
// concurrent callers never share scratch.
struct Workspace {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;  // own_stream or the caller's (cmx_set_stream)
  hipEvent_t ev_begin = nullptr, ev_end = nullptr;
  hipEvent_t ev_k0 = nullptr, ev_k1 = nullptr;  // dominant-kernel bracket
  hipEvent_t ev_x0 = nullptr, ev_x1 = nullptr;  // branch-and-bound expansion bracket
  static constexpr int kNumBuffers = 32;
  DeviceBuffer dev[kNumBuffers];
  TaggedBuffer tagged[2];
  PinnedBuffer pinned[4];
  ~Workspace();
};

// The HIP-event brackets behind cmx_match_stats' *_ms fields: four to six event packets in a
// chain of seven launches cost a single fast-2D search 24 of its 156 us (round 4, same box), so
// they are recorded only under the debug option `timing` (bench legs that report kernel times,
// the profiling tools); without it the *_ms fields are 0.
inline bool TimingEnabled() { return Debug().timing != 0; }
inline void RecordEvent(hipEvent_t ev, hipStream_t stream) {
  if (TimingEnabled()) CMX_HIP(hipEventRecord(ev, stream));
}
inline float ElapsedMs(hipEvent_t from, hipEvent_t to) {
  if (!TimingEnabled()) return 0.f;
  float ms = 0.f;
  // (the switch is process-wide and may be turned on while a call is in flight: that call's
  // events were never recorded -- it reports no timing, it does not fail)
  if (hipEventElapsedTime(&ms, from, to) != hipSuccess) {
    (void)hipGetLastError();
    return 0.f;
  }
  return ms;
}

// Optional per-stage timing (debug switch trace): an event after every
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

// Four streams created one right after the other, for the parts of ONE call that must overlap on
// the device.  The runtime deals its hardware queues (four by default) to streams as they are
// created, least-loaded queue first, and two streams on one queue run one after the other: the
// workspaces' own streams are created whenever a caller first needs one, so the two workspaces a
// batch call happens to get may well share a queue (a 512-match RT-2D call measured 409 us in a
// process that had served eight concurrent fast-2D callers, 256 us in a fresh one).  Streams
// created back to back land on different queues.  Leased like workspaces; a set is never used by
// two calls at once.
class StreamSetLease {
 public:
  static constexpr int kStreams = 4;
  explicit StreamSetLease(int device);
  ~StreamSetLease();
  hipStream_t stream(int k) const { return set_->s[k % kStreams]; }
  struct Set { int device; hipStream_t s[kStreams]; };
 private:
  Set* set_;
};

// In-kernel timelines (debug switch timeline): see cmx_device.h Stamp().
bool TimelineEnabled();
// Prints, per stamp k, the median / max over blocks of (t_k - t_0) and the span of the whole
// launch (first t_0 to last stamp) in microseconds; `device` holds blocks x 16 stamps.
void ReportTimeline(const char* name, const unsigned long long* device, int blocks,
                    hipStream_t stream);

// filters.hip: stable sort of n (32-bit key, index) pairs on ws.stream; scratch from
// ws.dev[temp_slot].
void StableSortPairs32(Workspace& ws, int temp_slot, const unsigned* keys_in, unsigned* keys_out,
                       const int* values_in, int* values_out, int n);

// Thread-local stream override (cmx_set_stream).
hipStream_t OverrideStream(int device);

inline int DivUp(int64_t a, int64_t b) { return static_cast<int>((a + b - 1) / b); }

}  // namespace cmx

#endif  // CMX_COMMON_H_
