// Internal declarations of the 2D matchers (device descriptors + host classes).
#ifndef CMX_SCAN_MATCHING_2D_H_
#define CMX_SCAN_MATCHING_2D_H_

#include <memory>
#include <mutex>
#include <vector>

#include "cmx_common.h"
#include "cmx_device.h"

namespace cmx {

// One level of a PrecomputationGridStack2D in device memory
// (SM2/fast_correlative_scan_matcher_2d.h:49-93): width 2^level, dims
// (nx+w-1) x (ny+w-1), cell (x0,y0) stored at [(x0+w-1) + (y0+w-1)*wx].
//
// `quads` is the layout the tree search reads: one dword per cell position
// (x, y) in [-w, wx) x [-w, wy) packing the four cells the children of a node read for
// one point -- byte 0: (x, y), byte 1: (x, y+w), byte 2: (x+w, y), byte 3: (x+w, y+w),
// 0 where a cell is outside the level.  A wave-wide byte gather with 64 unrelated
// addresses costs ~90 cycles of texture-address time per instruction (profiles/HISTORY.md 5.3);
// fetching the four children at once cuts those instructions by four.
struct LevelDesc {
  const uint8_t* cells;
  int wx, wy;
  // Quads (the four child cells of a node's point in one dword), element (x + w, y + w) of a
  // (wx + w) x (wy + w) array stored in tiles of 8 x 4 elements = one 128-byte line (see
  // QuadOffset); null for the top level.
  const uint32_t* quads;
  int qx, qy;              // wx + w, wy + w
  int qtx;                 // tiles per row of tiles: ceil(qx / 8)
};

// The points of a scan lie along walls: 64 consecutive ones cover ~64 cells of a wall.  In a
// row-major array a wall along y costs a cache line per point (4 useful bytes of 128); in 8 x 4
// tiles the same 64 gathers touch 8 - 16 lines whatever the wall's direction.
__host__ __device__ inline unsigned QuadOffset(unsigned X, unsigned Y, unsigned qtx) {
  return (((Y >> 2) * qtx + (X >> 3)) << 5) | ((Y & 3u) << 3) | (X & 7u);
}

// Device-visible description of one (scan, submap) search.
struct Fast2DProblem {
  LevelDesc level[kMaxDepth];
  int depth;            // branch_and_bound_depth (number of levels)
  int nx, ny;           // CellLimits
  int nl;               // linear window in cells (bounds start at +-nl)
  double res, max_x, max_y;
  double inv_res;       // RN(1 / res), for the division-free cell index (cmx_device.h)
  float tx, ty;         // initial translation narrowed to f32
  float init_qw, init_qz;   // Quaternion(AngleAxisf(f32(theta0), Z))
  int num_scans;
  int coarse_capacity;  // size of coarse_score / coarse_sum (= num_scans * coarse_stride)
  int coarse_stride;    // lowest-resolution candidates of scan s start at s * coarse_stride
                        // (an upper bound of every scan's count, known on the host: the
                        // layout needs no prefix sum over the scans)
  int use_fused;        // 1: PrepScoreFusedKernel (prep + bucketing + plane scoring in one
                        // block per rotation, records never leave LDS)
  int write_all_discrete;   // debug: keep every discretised scan (default: only scans whose
                            // best lowest-resolution score reaches min_score)
  const float2* scan_rot;   // [num_scans] (w, z) of AngleAxisf(f32(delta_theta_s), Z)
  float min_s, score_scale; // ToScore(v) = min_s + v * score_scale
  float min_score;          // caller's acceptance threshold
  // Lowest-resolution level re-laid out as "phase planes" (see fast_2d.hip):
  // plane(py,px)[J][I] = level[depth-1] cell (I*w+px, J*w+py), w = 2^(depth-1).
  const uint8_t* planes;    // [(w*w + 1) planes][plane_stride]; the last plane is all zero
  int plane_i, plane_j;     // cells per plane along x / y
  int plane_stride;         // bytes per plane (multiple of 64)
  int use_planes;           // 0: generic gather scoring of the lowest resolution
  // Group bounds of the fused front end (round 6, fast_2d.hip "group bounds"): the same planes of
  // the lowest-resolution level dilated by two cells either way (and stored two cells up: cell
  // (X + 2, Y + 2) of the dilated image bounds the cells within two of (X, Y)); `group` adjacent
  // rotations share ONE sum over them.  group == 1: every rotation summed on `planes` itself.
  const uint8_t* planes_group;
  int group;                // rotations per workgroup of the fused front end: 1 or 3
  int group_verify;         // debug (fast2d_group_verify): bit 0 every group bound checked against the
                            // exact sums of its rotations on the device (error 3 on a violation),
                            // bit 1 every unit treated as if its premise had failed
  // scratch
  uint32_t* discrete;   // [num_scans][n] packed int16 (x | y << 16)
  int4* bounds;         // [num_scans] (min_x, max_x, min_y, max_y) after ShrinkToFit
  int2* coarse_dims;    // [num_scans] (#x, #y lowest-resolution candidates)
  float* coarse_score;  // [coarse_capacity]
  int* coarse_sum;      // [coarse_capacity]
  uint2* sorted;        // [num_scans][n] points bucketed by lattice block:
                        // (plane byte offset, bx * pitch + by of the plane scorer's accumulators)
  int* sorted_count;    // [num_scans]
  int2* scan_best;      // [num_scans] (best sum, local candidate index) of each scan
  unsigned long long* timeline;   // CMX_TIMELINE=1: 16 stamps per fused-kernel block, else null
  const float* xyz;     // the device point cloud
  int recompute_scans;  // 1 (fused front end): `discrete` is not written; the tree search
                        // re-derives a scan's cells from xyz (60 instructions per point,
                        // bit-identical) instead of 9 MB per match going to HBM and back
  int store_scans;      // with recompute_scans: the coarse filter writes the cells of the scans
                        // that keep a candidate to `discrete`, and the wave-per-node expansion
                        // (several nodes per such scan) reads them instead of recomputing
};

// Branch-and-bound node.
struct Node2D {
  int problem;      // index into the batch | level << 24
  int scan;
  int dx, dy;       // x/y_index_offset of the node's lowest corner
  float score;
  int coarse_index; // generation index of the lowest-resolution ancestor
  unsigned path;    // sibling ranks along the descent, 2 bits per level
  float coarse_score;
};

constexpr int kStatShards = 16;   // work counters are sharded to keep atomics uncontended

struct ProblemState {       // per problem, device
  unsigned best_bits;       // float bits of the best leaf score so far (>= min_score)
  int coarse_total;
  int error;                // 1: cell index outside int16, 2: coarse capacity exceeded
  int done_top;             // units of the fused front end whose group-bound premise failed
  int done_shard[kStatShards];   // (unused since round 5)
  unsigned long long scored_shard[kStatShards];    // candidates scored below the top level
  unsigned long long expanded_shard[kStatShards];  // nodes whose children were scored
};

struct BestLeaf {           // per problem, device → host
  float score;
  int scan, dx, dy;
  int found;                // a leaf with score > min_score exists
  int ties;                 // number of recorded leaves sharing the best score
  int pad0, pad1;
};

class Fast2DMatcher {
 public:
  Fast2DMatcher(const cmx_fast2d_options& options, const cmx_grid2d_limits& limits,
                const uint16_t* cells, int device);
  ~Fast2DMatcher();
  int device() const { return device_; }
  const cmx_fast2d_options& options() const { return options_; }
  const cmx_grid2d_limits& limits() const { return limits_; }
  int depth() const { return options_.branch_and_bound_depth; }
  const LevelDesc& level(int i) const { return levels_[i]; }
  const uint8_t* planes() const { return planes_; }
  const uint8_t* planes_group() const { return planes_group_; }   // (or null: no group bounds)
  // The submap's own correspondence-cost cells (uint16, as uploaded): what the Ceres
  // refinement after a match interpolates (constraint_builder_2d.cc:245-249).
  const uint16_t* grid_cells() const { return grid_cells_; }
  int plane_i() const { return plane_i_; }
  int plane_j() const { return plane_j_; }
  int plane_stride() const { return plane_stride_; }
  size_t level_offset(int i) const { return level_offsets_[i]; }
  float min_s() const { return min_s_; }
  float score_scale() const { return score_scale_; }

 private:
  cmx_fast2d_options options_;
  cmx_grid2d_limits limits_;
  int device_;
  void* stack_mem_ = nullptr;      // all levels, contiguous
  void* quads_mem_ = nullptr;      // quad layouts of levels 0 .. depth-2, contiguous
  uint8_t* planes_ = nullptr;      // phase planes of the lowest-resolution level (or null)
  uint8_t* planes_group_ = nullptr;   // phase planes of its dilation by two cells (or null)
  uint16_t* grid_cells_ = nullptr; // the grid itself (2 B per cell), for the refinement step
  int plane_i_ = 0, plane_j_ = 0, plane_stride_ = 0;
  std::vector<LevelDesc> levels_;
  std::vector<size_t> level_offsets_;
  float min_s_, score_scale_;
};

// Per-rotation (cos, sin) half-angle pairs of a search with `num_angular` perturbations
// of `step` radians (GenerateRotatedScans, SM2/correlative_scan_matcher_2d.cc:99-107),
// evaluated with the host's libm (bit-identical to the reference's AngleAxisf) and kept
// in a bounded process-wide cache of HOST values: every call copies the table it needs
// into its own upload buffer, so no device allocation is shared between calls in flight
// and nothing is ever freed on the hot path.
std::shared_ptr<const std::vector<float2>> HostRotationTable(double step, int num_angular);
// The same values written to out[0 .. 2 num_angular]: no cache, no lock (the real-time matcher:
// every scan has a step of its own -- it depends on the scan's longest range -- so a cache keyed
// by the step only ever misses, and its lock serialised the host threads of concurrent callers).
void FillRotationTable(double step, int num_angular, float2* out);

// The quantised image of a resident grid (rt_2d_tiles.hip, Rt2DQuantKernel): cells as 16-bit
// q = (32767 - value) >> 5 with a zero halo, what the tile workgroups of the real-time matcher
// copy into LDS.  It depends on the grid's cells (`version`, bumped by every insert / grow /
// crop) and on the window (nl); a cmx_grid2d owns one cache.  TWO buffers: the real caller
// inserts a scan after every match, so every match meets a new version -- its image is built
// into the buffer nobody reads while matches on the previous version may still be reading
// theirs.  No lock is held across kernels: a buffer is pinned by a reader count, a buffer being
// built is exclusive, and its metadata is published only after the build's stream has been
// waited for (a failed call leaves it invalid).
struct Rt2DImageCache {
  struct Buffer {
    uint16_t* image = nullptr;   // device
    size_t capacity = 0;
    bool valid = false, building = false;
    int readers = 0;
    unsigned long long version = 0;
    int nl = -1, gpitch = 0, grows = 0;
  };
  std::mutex mutex;              // guards the fields above; never held across device work
  Buffer buffer[2];
  int Acquire(unsigned long long version, int nl, int gpitch, int grows, size_t bytes, bool* build);
  void Publish(int k, unsigned long long version, int nl, int gpitch, int grows);
  void Release(int k, bool was_building);
  ~Rt2DImageCache();
};

// RealTimeCorrelativeScanMatcher2D::Match (rt_2d.hip); see there.
struct Rt2DItem {            // one match of a batch
  const cmx_grid2d_limits* limits;
  const uint16_t* cells;          // host grid (or null with device_cells)
  const uint16_t* weight_cells;   // host TSDF weights (null: probability grid)
  float max_tsd, max_weight;
  const uint16_t* device_cells;   // grid already in HBM
  const cmx_pose2d* initial;
  const float* xyz;
  int n;
  double* score;
  cmx_pose2d* pose;
  const float* device_xyz = nullptr;      // the same cloud already in HBM (cmx_cloud): no upload
  const int* far_points = nullptr;        // cmx_cloud::far_points (null: scan the whole cloud)
  int num_far_points = 0;
  Rt2DImageCache* image_cache = nullptr;  // with device_cells of a cmx_grid2d
  unsigned long long grid_version = 0;
};
// SearchParameters of one match (SM2/correlative_scan_matcher_2d.cc:27-47 on the cloud
// pre-rotated by the initial yaw, real_time_..._2d.cc:123-130): host, libm.
struct Rt2DSearch {
  double step;            // angular_perturbation_step_size
  int na, num_scans, nl;  // num_angular_perturbations, 2 na + 1, linear window in cells
  float q0w, q0z;         // Quaternion(AngleAxisf(f32(theta0), Z))
  float max_range;        // longest xy range of the cloud (f32, as SearchParameters computes it)
};
void Rt2DComputeSearch(const cmx_rt_options* options, const Rt2DItem& item, Rt2DSearch* out);
// Exact weighting (libm) + first-maximum rule over a match's finalists, ascending by candidate
// index (:142-143,170-174): writes the item's score and pose.
void Rt2DFinishOnHost(const cmx_rt_options* options, const Rt2DItem& item, const Rt2DSearch& search,
                      const std::pair<int, float>* finalists, size_t count);
// rt_2d_tiles.hip: the bulk path for probability grids, in three steps so that one host thread
// keeps the parts of a large batch in flight on streams of their own (see there).
class Rt2DTileCall {
 public:
  // `concurrent_calls`: how many calls of this kind share the device at the same time (the parts
  // of a batch): each sizes its tile grid and its work items for 1 / concurrent_calls of the CUs.
  Rt2DTileCall(const cmx_rt_options* options, const Rt2DItem* items, const Rt2DSearch* search,
               int num, int32_t device, int concurrent_calls = 1, int batch_matches = 0);
  // (`batch_matches`: matches of the whole batch this call is a part of -- 0: `num` -- which is
  // what decides between the block bounds and the exhaustive tile kernel)
  ~Rt2DTileCall();
  bool Plan();                          // false: not eligible for this path
  // asynchronous.  `on_stream`: the call's work goes on that stream instead of its workspace's
  // own (the parts of a batch: StreamSetLease).
  void Enqueue(hipStream_t on_stream = nullptr);
  // waits.  Matches the bulk path could not decide go on `redo` (indices into this call's items:
  // the caller repeats them on the per-candidate kernels); without a list: false, nothing written
  bool Collect(cmx_match_stats* stats, std::vector<int>* redo = nullptr);
 private:
  struct Impl;
  std::unique_ptr<Impl> impl_;
};
void Rt2DMatchBatch(const cmx_rt_options* options, const Rt2DItem* items, int num, int32_t device,
                    cmx_match_stats* stats);
void Rt2DMatch(const cmx_rt_options* options, const cmx_grid2d_limits* limits,
               const uint16_t* cells, const uint16_t* weight_cells, float max_tsd,
               float max_weight, const cmx_pose2d* initial_pose_estimate,
               const float* point_cloud_xyz, int32_t num_points, int32_t device, double* score,
               cmx_pose2d* pose_estimate, cmx_match_stats* stats, const uint16_t* device_cells);

}  // namespace cmx

struct cmx_fast2d {
  std::unique_ptr<cmx::Fast2DMatcher> impl;
};

struct cmx_cloud {
  int device = 0;
  int num_points = 0;
  float* xyz = nullptr;          // device
  float max_range_xy = 0.f;      // max ||p.xy|| (f32, as SearchParameters computes it)
  float max_range_xyz = 0.f;
  std::vector<float> host_xyz;
  // Indices of the points whose exact squared xy range is within 1e-4 (relative) of the largest.
  // The real-time matcher needs max ||R p|| over the cloud rotated by the initial yaw, EVALUATED IN
  // f32 (SearchParameters, SM2/correlative_scan_matcher_2d.cc:27-36): that value differs from
  // the exact ||p||^2 by less than 2e-6 relative (ten roundings and |q|^2 = 1 +- 2e-7), so only
  // these points can produce the maximum -- the per-match range scan shrinks from n points
  // to a handful.  Empty: more than 64 candidates (scan everything).
  std::vector<int> far_points;
};

#endif  // CMX_SCAN_MATCHING_2D_H_
