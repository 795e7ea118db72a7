// Multi-GPU loop-closure search behind the C ABI: one host process driving the GPUs of a node.
//
// The unit of work is one (scan, submap) search and the units are independent
// (constraints/constraint_builder_2d.cc:97-111, constraint_builder_3d.cc:79-142 schedule them
// as independent thread-pool tasks), so submaps are partitioned over the devices -- every
// matcher (precomputation stack) lives in the HBM of the device it was created on -- and a
// node's scan is searched against every device's submaps concurrently: one host thread per
// device issues that device's batch, results come back in submap order (the reference's
// semantics: one optional constraint per pair).  There is no data-path collective.  The one
// exchange the search has -- the node-wide best match for global localisation -- is an RCCL
// all-reduce(max) of a packed 8-byte key over xGMI (score bits << 32 | ~submap index: equal
// scores resolve to the lowest index on any number of devices), issued on every device's
// stream inside one ncclGroup.  RCCL is bound at cmx_comm_init through dlopen: the library has
// no link-time dependency on it and single-GPU users never load it.
#include <dlfcn.h>

#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <string>
#include <thread>

#include "scan_matching_2d.h"
#include "scan_matching_3d.h"

struct cmx_fast3d;
namespace cmx {
int Fast3DDevice(const cmx_fast3d* matcher);   // fast_3d.hip
namespace {

// The few RCCL entry points used, with the types of rccl.h (ncclComm_t is an opaque pointer,
// ncclInt64 = 4, ncclMax = 2, ncclSuccess = 0).
struct Rccl {
  void* handle = nullptr;
  int (*CommInitAll)(void** comms, int ndev, const int* devlist) = nullptr;
  int (*CommDestroy)(void* comm) = nullptr;
  int (*AllReduce)(const void* send, void* recv, size_t count, int datatype, int op, void* comm,
                   hipStream_t stream) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};

const Rccl& LoadRccl() {
  static Rccl rccl = [] {
    Rccl r;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (r.handle) break;
    }
    if (!r.handle) return r;
    r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(dlsym(r.handle, "ncclCommInitAll"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.handle, "ncclCommDestroy"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(r.handle, "ncclAllReduce"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(r.handle, "ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(r.handle, "ncclGroupEnd"));
    r.GetErrorString =
        reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.handle, "ncclGetErrorString"));
    return r;
  }();
  return rccl;
}

// A worker thread bound to one device; runs one closure at a time.
class DeviceWorker {
 public:
  explicit DeviceWorker(int device) : device_(device), thread_([this] { Loop(); }) {}
  ~DeviceWorker() {
    {
      std::lock_guard<std::mutex> lock(mu_);
      stop_ = true;
    }
    cv_.notify_all();
    thread_.join();
  }
  void Submit(std::function<void()> job) {
    {
      std::lock_guard<std::mutex> lock(mu_);
      job_ = std::move(job);
      done_ = false;
    }
    cv_.notify_all();
  }
  void Wait() {
    std::unique_lock<std::mutex> lock(mu_);
    cv_.wait(lock, [this] { return done_; });
  }
  int device() const { return device_; }

 private:
  void Loop() {
    (void)hipSetDevice(device_);
    std::unique_lock<std::mutex> lock(mu_);
    for (;;) {
      cv_.wait(lock, [this] { return stop_ || job_; });
      if (stop_) return;
      std::function<void()> job = std::move(job_);
      job_ = nullptr;
      lock.unlock();
      job();
      lock.lock();
      done_ = true;
      cv_.notify_all();
    }
  }
  int device_;
  std::mutex mu_;
  std::condition_variable cv_;
  std::function<void()> job_;
  bool done_ = true, stop_ = false;
  std::thread thread_;
};

}  // namespace
}  // namespace cmx

struct cmx_comm {
  std::vector<int> devices;                          // per rank
  bool virtual_ranks = false;                        // several ranks on ONE device (tests): the
                                                     // key is reduced on the host, no RCCL
  std::vector<void*> comms;                          // ncclComm_t per device (empty: no RCCL)
  std::vector<hipStream_t> streams;                  // one per device, for the collectives
  std::vector<long long*> d_keys;                    // 2 x int64 per device (send, recv)
  std::vector<std::unique_ptr<cmx::DeviceWorker>> workers;
  std::mutex mu;                                     // one sharded call at a time per communicator
};

// ---- partition / key helpers (pure host arithmetic: usable and tested without a device) -----
extern "C" {

void cmx_shard_range(int64_t num_items, int32_t rank, int32_t world_size, int64_t* begin,
                     int64_t* end) {
  if (world_size < 1) world_size = 1;
  const int64_t base = num_items / world_size, extra = num_items % world_size;
  const int64_t b = rank * base + (rank < extra ? rank : extra);
  if (begin) *begin = b;
  if (end) *end = b + base + (rank < extra ? 1 : 0);
}

int64_t cmx_pack_best_key(const int32_t* found, const float* scores, int64_t num,
                          int64_t first_global_index) {
  int64_t best = -1;     // CMX_BEST_KEY_NOT_FOUND
  for (int64_t i = 0; i < num; ++i) {
    if (!found[i]) continue;
    uint32_t bits;
    std::memcpy(&bits, &scores[i], sizeof(bits));
    const int64_t key = (static_cast<int64_t>(bits) << 32) |
                        (0xFFFFFFFFll - (first_global_index + i));
    if (key > best) best = key;
  }
  return best;
}

void cmx_unpack_best_key(int64_t key, int32_t* found, float* score, int64_t* global_index) {
  const bool ok = key >= 0;
  if (found) *found = ok ? 1 : 0;
  const uint32_t bits = ok ? static_cast<uint32_t>(key >> 32) : 0u;
  if (score) std::memcpy(score, &bits, sizeof(bits));
  if (global_index) *global_index = ok ? 0xFFFFFFFFll - (key & 0xFFFFFFFFll) : -1;
}

cmx_status cmx_comm_init(const int32_t* devices, int32_t num_devices, cmx_comm** out) {
  return cmx::Guard([&] {
    CMX_REQUIRE(out != nullptr && num_devices >= 1, "bad argument");
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
      cmx::SetLastError("no HIP device available; this library has no CPU fallback");
      throw cmx::HipError{CMX_DEVICE_ERROR};
    }
    int current = -1;
    if (hipGetDevice(&current) != hipSuccess) current = -1;
    // cmx_comm_destroy releases whatever exists so far (comms, streams, keys, workers): a
    // failure half-way through leaks nothing.
    struct Release {
      int device;
      void operator()(cmx_comm* c) const {
        cmx_comm_destroy(c);
        if (device >= 0) (void)hipSetDevice(device);
      }
    };
    std::unique_ptr<cmx_comm, Release> comm(new cmx_comm, Release{current});
    // Debug switch comm_virtual_ranks = N (tests on a one-GPU box): a communicator of ONE device
    // becomes N ranks on it -- N worker threads, N streams, entries dealt to ranks by index range
    // as a caller places its matchers (cmx_comm_device_of), the best key reduced on the host.
    // Everything of a sharded call but the RCCL collective itself runs with world = N.
    const int virtual_ranks = num_devices == 1 ? cmx::Debug().comm_virtual_ranks : 0;
    if (virtual_ranks > 1) {
      const int d = devices ? devices[0] : 0;
      CMX_REQUIRE(d >= 0 && d < count, "device %d out of range [0,%d)", d, count);
      comm->devices.assign(virtual_ranks, d);
      comm->virtual_ranks = true;
      num_devices = virtual_ranks;
    } else {
      for (int i = 0; i < num_devices; ++i) {
        const int d = devices ? devices[i] : i;
        CMX_REQUIRE(d >= 0 && d < count, "device %d out of range [0,%d)", d, count);
        for (int prev : comm->devices) CMX_REQUIRE(prev != d, "device %d listed twice", d);
        comm->devices.push_back(d);
      }
    }
    // (one device needs no collective: a single-GPU deployment works without librccl.  Debug
    // switch comm_force_rccl: a communicator of ONE device is built by RCCL all the same and its
    // all-reduce is RCCL's -- the binding below, its prototypes and enum values, executed on a
    // one-GPU box: tests/test_gpu_r2_paths.py)
    if ((num_devices > 1 || cmx::Debug().comm_force_rccl) && !comm->virtual_ranks) {
      const cmx::Rccl& rccl = cmx::LoadRccl();
      CMX_REQUIRE(rccl.handle && rccl.CommInitAll && rccl.AllReduce && rccl.GroupStart &&
                      rccl.GroupEnd && rccl.CommDestroy,
                  "librccl.so.1 could not be loaded (needed for cmx_comm_init with several devices)");
      comm->comms.assign(num_devices, nullptr);
      const int rc = rccl.CommInitAll(comm->comms.data(), num_devices, comm->devices.data());
      if (rc != 0) {
        cmx::SetLastError("ncclCommInitAll failed: %s",
                          rccl.GetErrorString ? rccl.GetErrorString(rc) : "?");
        throw cmx::HipError{CMX_DEVICE_ERROR};
      }
    }
    comm->streams.assign(num_devices, nullptr);
    comm->d_keys.assign(num_devices, nullptr);
    for (int i = 0; i < num_devices; ++i) {
      CMX_HIP(hipSetDevice(comm->devices[i]));
      CMX_HIP(hipStreamCreateWithFlags(&comm->streams[i], hipStreamNonBlocking));
      CMX_HIP(hipMalloc(reinterpret_cast<void**>(&comm->d_keys[i]), 2 * sizeof(long long)));
      comm->workers.emplace_back(new cmx::DeviceWorker(comm->devices[i]));
    }
    *out = comm.release();
    if (current >= 0) (void)hipSetDevice(current);
  });
}

void cmx_comm_destroy(cmx_comm* comm) {
  if (!comm) return;
  int current = -1;
  if (hipGetDevice(&current) != hipSuccess) current = -1;
  const bool have_comms = !comm->comms.empty();
  const cmx::Rccl rccl_none{};
  const cmx::Rccl& rccl = have_comms ? cmx::LoadRccl() : rccl_none;
  comm->workers.clear();
  for (size_t i = 0; i < comm->devices.size(); ++i) {
    (void)hipSetDevice(comm->devices[i]);
    if (i < comm->d_keys.size() && comm->d_keys[i]) (void)hipFree(comm->d_keys[i]);
    if (i < comm->streams.size() && comm->streams[i]) (void)hipStreamDestroy(comm->streams[i]);
    if (i < comm->comms.size() && comm->comms[i] && rccl.CommDestroy) rccl.CommDestroy(comm->comms[i]);
  }
  if (current >= 0) (void)hipSetDevice(current);
  delete comm;
}

int32_t cmx_comm_uses_rccl(const cmx_comm* comm) {
  return comm && !comm->comms.empty() ? 1 : 0;
}

int32_t cmx_comm_num_devices(const cmx_comm* comm) {
  return comm ? static_cast<int32_t>(comm->devices.size()) : 0;
}

int32_t cmx_comm_device_of(const cmx_comm* comm, int64_t index, int64_t num_items) {
  if (!comm || comm->devices.empty()) return -1;
  const int world = static_cast<int>(comm->devices.size());
  for (int r = 0; r < world; ++r) {
    int64_t b, e;
    cmx_shard_range(num_items, r, world, &b, &e);
    if (index >= b && index < e) return comm->devices[r];
  }
  return -1;
}

}  // extern "C"

namespace cmx {
namespace {

// The sharded calls make every device of the communicator current in turn on the CALLING
// thread; the caller's own current device (torch's, say) is put back on every exit path.
struct RestoreDevice {
  int device = -1;
  RestoreDevice() { if (hipGetDevice(&device) != hipSuccess) device = -1; }
  ~RestoreDevice() { if (device >= 0) (void)hipSetDevice(device); }
};

// Node-wide best match: every device contributes the key of its own block, one
// all-reduce(max) over the communicator; returns the reduced key (read back from rank 0).
int64_t AllReduceBest(cmx_comm* comm, const std::vector<int64_t>& keys) {
  const int world = static_cast<int>(comm->devices.size());
  if (comm->comms.empty()) {       // one device, or virtual ranks on it: max over the ranks' keys
    int64_t best = -1;
    for (int r = 0; r < world; ++r) best = std::max(best, keys[r]);
    return best;
  }
  const RestoreDevice restore;
  const Rccl& rccl = LoadRccl();
  for (int r = 0; r < world; ++r) {
    CMX_HIP(hipSetDevice(comm->devices[r]));
    const long long k = keys[r];
    CMX_HIP(hipMemcpyAsync(comm->d_keys[r], &k, sizeof(k), hipMemcpyHostToDevice, comm->streams[r]));
  }
  int rc = rccl.GroupStart();
  for (int r = 0; r < world && rc == 0; ++r) {
    CMX_HIP(hipSetDevice(comm->devices[r]));
    rc = rccl.AllReduce(comm->d_keys[r], comm->d_keys[r] + 1, 1, /*ncclInt64*/ 4, /*ncclMax*/ 2,
                        comm->comms[r], comm->streams[r]);
  }
  const int rc_end = rccl.GroupEnd();
  if (rc == 0) rc = rc_end;
  if (rc != 0) {
    SetLastError("RCCL all-reduce failed: %s", rccl.GetErrorString ? rccl.GetErrorString(rc) : "?");
    throw HipError{CMX_DEVICE_ERROR};
  }
  long long reduced = -1;
  CMX_HIP(hipSetDevice(comm->devices[0]));
  CMX_HIP(hipMemcpyAsync(&reduced, comm->d_keys[0] + 1, sizeof(reduced), hipMemcpyDeviceToHost,
                         comm->streams[0]));
  for (int r = 0; r < world; ++r) {
    CMX_HIP(hipSetDevice(comm->devices[r]));
    CMX_HIP(hipStreamSynchronize(comm->streams[r]));
  }
  return reduced;
}

// Runs `per_rank(rank, indices of that rank's entries)` on every device's worker thread;
// entries are assigned to the rank whose device owns their matcher.  Collects status + error
// text per rank and rethrows the first failure.
template <typename DeviceOf, typename PerRank>
void FanOut(cmx_comm* comm, int num, DeviceOf device_of, PerRank per_rank) {
  const int world = static_cast<int>(comm->devices.size());
  std::vector<std::vector<int>> mine(world);
  for (int p = 0; p < num; ++p) {
    const int d = device_of(p);
    int rank = -1;
    for (int r = 0; r < world && rank < 0; ++r) {
      if (comm->devices[r] != d) continue;
      if (!comm->virtual_ranks) { rank = r; break; }
      int64_t b, e;                          // (virtual ranks share the device: by index range)
      cmx_shard_range(num, r, world, &b, &e);
      if (p >= b && p < e) rank = r;
    }
    CMX_REQUIRE(rank >= 0, "entry %d lives on device %d, which is not in the communicator", p, d);
    mine[rank].push_back(p);
  }
  std::vector<cmx_status> status(world, CMX_OK);
  std::vector<std::string> errors(world);
  for (int r = 0; r < world; ++r) {
    if (mine[r].empty()) continue;
    comm->workers[r]->Submit([&, r] {
      // The job body allocates (std::vector) before it reaches the guarded ABI call: an
      // exception must not leave the worker thread (std::terminate).
      cmx_status inner = CMX_OK;
      const cmx_status outer = Guard([&] { inner = per_rank(r, mine[r]); });
      status[r] = outer != CMX_OK ? outer : inner;
      if (status[r] != CMX_OK) errors[r] = LastError();
    });
  }
  for (int r = 0; r < world; ++r) comm->workers[r]->Wait();
  for (int r = 0; r < world; ++r) {
    if (status[r] != CMX_OK) {
      SetLastError("device %d: %s", comm->devices[r], errors[r].c_str());
      throw HipError{status[r]};
    }
  }
}

void AddStats(cmx_match_stats* total, const cmx_match_stats& s) {
  total->candidates_scored += s.candidates_scored;
  total->coarse_candidates += s.coarse_candidates;
  total->nodes_expanded += s.nodes_expanded;
  total->num_scans += s.num_scans;
  total->device_ms = std::max(total->device_ms, s.device_ms);        // the devices overlap
  total->dominant_kernel_ms = std::max(total->dominant_kernel_ms, s.dominant_kernel_ms);
  total->expansion_ms = std::max(total->expansion_ms, s.expansion_ms);
  total->expansion_launches += s.expansion_launches;
  total->expansion_nodes += s.expansion_nodes;
  total->expansion_lookups += s.expansion_lookups;
}

}  // namespace
}  // namespace cmx

extern "C" {

cmx_status cmx_fast2d_match_sharded(cmx_comm* comm, const cmx_fast2d* const* matchers,
                                    int32_t num_matchers, const cmx_pose2d* initial_pose_estimates,
                                    const int32_t* match_full_submap, const float* min_scores,
                                    const float* point_cloud_xyz, int32_t num_points,
                                    int32_t* found, float* scores, cmx_pose2d* pose_estimates,
                                    int32_t* best_index, float* best_score,
                                    cmx_match_stats* stats) {
  using namespace cmx;
  return Guard([&] {
    CMX_REQUIRE(comm && matchers && match_full_submap && min_scores && found && scores &&
                    pose_estimates && num_matchers >= 1,
                "null argument");
    for (int p = 0; p < num_matchers; ++p)
      CMX_REQUIRE(matchers[p] && matchers[p]->impl, "null matcher handle");
    std::lock_guard<std::mutex> lock(comm->mu);
    const int world = static_cast<int>(comm->devices.size());
    std::vector<cmx_match_stats> rank_stats(world);
    std::vector<int64_t> keys(world, -1);
    FanOut(comm, num_matchers, [&](int p) { return matchers[p]->impl->device(); },
           [&](int r, const std::vector<int>& idx) -> cmx_status {
             const int m = static_cast<int>(idx.size());
             std::vector<const cmx_fast2d*> h(m);
             std::vector<cmx_pose2d> init(m), poses(m);
             std::vector<int32_t> full(m), f(m);
             std::vector<float> thr(m), sc(m);
             for (int k = 0; k < m; ++k) {
               h[k] = matchers[idx[k]];
               full[k] = match_full_submap[idx[k]];
               thr[k] = min_scores[idx[k]];
               init[k] = initial_pose_estimates ? initial_pose_estimates[idx[k]] : cmx_pose2d{};
             }
             const cmx_status st = cmx_fast2d_match_batch(
                 h.data(), m, init.data(), full.data(), thr.data(), point_cloud_xyz, num_points,
                 f.data(), sc.data(), poses.data(), &rank_stats[r]);
             if (st != CMX_OK) return st;
             int64_t key = -1;
             for (int k = 0; k < m; ++k) {
               found[idx[k]] = f[k];
               scores[idx[k]] = sc[k];
               pose_estimates[idx[k]] = poses[k];
               const int64_t one = cmx_pack_best_key(&f[k], &sc[k], 1, idx[k]);
               if (one > key) key = one;
             }
             keys[r] = key;
             return CMX_OK;
           });
    // (a caller that wants neither the best index nor its score -- the constraint builders take
    // every pair's result -- pays for no collective)
    const int64_t reduced = best_index || best_score ? AllReduceBest(comm, keys) : -1;
    int32_t any;
    float score;
    int64_t index;
    cmx_unpack_best_key(reduced, &any, &score, &index);
    if (best_index) *best_index = any ? static_cast<int32_t>(index) : -1;
    if (best_score) *best_score = any ? score : 0.f;
    if (stats) {
      *stats = cmx_match_stats{};
      for (const cmx_match_stats& s : rank_stats) AddStats(stats, s);
    }
  });
}

cmx_status cmx_fast3d_match_sharded(cmx_comm* comm, const cmx_fast3d* const* matchers,
                                    int32_t num_pairs, const cmx_pose3d* node_poses,
                                    const cmx_pose3d* submap_poses,
                                    const int32_t* match_full_submap, const float* min_scores,
                                    const cmx_node_data3d* data, int32_t* found,
                                    cmx_result3d* results, int32_t* best_index, float* best_score,
                                    cmx_match_stats* stats) {
  using namespace cmx;
  return Guard([&] {
    CMX_REQUIRE(comm && matchers && node_poses && submap_poses && match_full_submap &&
                    min_scores && data && found && results && num_pairs >= 1,
                "null argument");
    for (int p = 0; p < num_pairs; ++p) CMX_REQUIRE(matchers[p] != nullptr, "null matcher handle");
    std::lock_guard<std::mutex> lock(comm->mu);
    const int world = static_cast<int>(comm->devices.size());
    std::vector<cmx_match_stats> rank_stats(world);
    std::vector<int64_t> keys(world, -1);
    FanOut(comm, num_pairs, [&](int p) { return Fast3DDevice(matchers[p]); },
           [&](int r, const std::vector<int>& idx) -> cmx_status {
             const int m = static_cast<int>(idx.size());
             std::vector<const cmx_fast3d*> h(m);
             std::vector<cmx_pose3d> np(m), sp(m);
             std::vector<int32_t> full(m), f(m);
             std::vector<float> thr(m);
             std::vector<cmx_result3d> res(m);
             for (int k = 0; k < m; ++k) {
               h[k] = matchers[idx[k]];
               np[k] = node_poses[idx[k]];
               sp[k] = submap_poses[idx[k]];
               full[k] = match_full_submap[idx[k]];
               thr[k] = min_scores[idx[k]];
             }
             const cmx_status st =
                 cmx_fast3d_match_batch(h.data(), m, np.data(), sp.data(), full.data(), thr.data(),
                                        data, f.data(), res.data(), &rank_stats[r]);
             if (st != CMX_OK) return st;
             int64_t key = -1;
             for (int k = 0; k < m; ++k) {
               found[idx[k]] = f[k];
               results[idx[k]] = res[k];
               const int64_t one = cmx_pack_best_key(&f[k], &res[k].score, 1, idx[k]);
               if (one > key) key = one;
             }
             keys[r] = key;
             return CMX_OK;
           });
    // (a caller that wants neither the best index nor its score -- the constraint builders take
    // every pair's result -- pays for no collective)
    const int64_t reduced = best_index || best_score ? AllReduceBest(comm, keys) : -1;
    int32_t any;
    float score;
    int64_t index;
    cmx_unpack_best_key(reduced, &any, &score, &index);
    if (best_index) *best_index = any ? static_cast<int32_t>(index) : -1;
    if (best_score) *best_score = any ? score : 0.f;
    if (stats) {
      *stats = cmx_match_stats{};
      for (const cmx_match_stats& s : rank_stats) AddStats(stats, s);
    }
  });
}

}  // extern "C"
