/*
 * cartographer_mi355x.h — C ABI of the MI355X-native correlative scan matchers.
 *
 * Drop-in boundary for cartographer's scan-matching hot path.  Every entry
 * point replaces one method of the four reference classes (paths relative to
 * the reference tree, `SM2` = cartographer/mapping/internal/2d/scan_matching,
 * `SM3` = .../3d/scan_matching):
 *
 *   cmx_rt2d_match                 RealTimeCorrelativeScanMatcher2D::Match
 *                                  SM2/real_time_correlative_scan_matcher_2d.h:66-68, .cc:117-149
 *   cmx_rt2d_match_tsdf            the same method on a TSDF2D grid (.cc:38-59, :159-167)
 *   cmx_fast2d_create / _destroy   FastCorrelativeScanMatcher2D ctor / dtor
 *                                  SM2/fast_correlative_scan_matcher_2d.h:114-118, .cc:188-196
 *   cmx_fast2d_match               FastCorrelativeScanMatcher2D::Match        .h:124-126, .cc:198-208
 *   cmx_fast2d_match_full_submap   FastCorrelativeScanMatcher2D::MatchFullSubmap .h:132-133, .cc:210-225
 *   cmx_fast2d_match_batch, cmx_fast2d_match_full_submap_batch
 *                                  the ConstraintBuilder2D fan-out of independent
 *                                  (node, submap) searches, constraints/constraint_builder_2d.cc:97-137
 *   cmx_ceres2d_match, cmx_ceres2d_match_grid, cmx_fast2d_refine_batch
 *                                  CeresScanMatcher2D::Match, SM2/ceres_scan_matcher_2d.cc:63-107
 *   cmx_ceres3d_match, cmx_fast3d_refine_batch
 *                                  CeresScanMatcher3D::Match, SM3/ceres_scan_matcher_3d.cc:90-156
 *   cmx_rt3d_match                 RealTimeCorrelativeScanMatcher3D::Match
 *                                  SM3/real_time_correlative_scan_matcher_3d.h:47-50, .cc:34-53
 *   cmx_fast3d_*                   FastCorrelativeScanMatcher3D ctor / Match / MatchFullSubmap
 *                                  SM3/fast_correlative_scan_matcher_3d.h:75-101, .cc:112-170
 *
 * Conventions
 *   - Plain C, POD only.  Host pointers are borrowed for the duration of the
 *     call; outputs are written to caller memory.  No exceptions cross the
 *     boundary: every function returns a cmx_status.
 *   - Reference CHECK failures (invalid options, null outputs) map to
 *     CMX_INVALID_ARGUMENT; "no match above min_score" is CMX_OK with
 *     *found == 0, exactly like the reference's `false` / `nullptr`.
 *   - There is NO CPU fallback: without a usable HIP device every compute
 *     entry point returns CMX_DEVICE_ERROR.
 *   - Matcher handles are immutable after creation and may be used from
 *     several host threads at once (the reference calls `Match*` concurrently
 *     from its thread pool, constraints/constraint_builder_2d.cc:97-111).
 */
#ifndef CARTOGRAPHER_MI355X_H_
#define CARTOGRAPHER_MI355X_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum cmx_status {
  CMX_OK = 0,
  CMX_INVALID_ARGUMENT = 1,
  CMX_DEVICE_ERROR = 2,
  CMX_OUT_OF_MEMORY = 3,
  CMX_UNSUPPORTED = 4
} cmx_status;

/* transform::Rigid2d (transform/rigid_transform.h:34-87): translation + yaw. */
typedef struct cmx_pose2d { double x, y, theta; } cmx_pose2d;
/* transform::Rigid3d (rigid_transform.h:117-180): translation + quaternion (w,x,y,z). */
typedef struct cmx_pose3d { double t[3]; double q[4]; } cmx_pose3d;

/* mapping::MapLimits + the Grid2D cost range (mapping/2d/map_limits.h:40-96,
 * grid_2d.h:60-63).  `cells` passed next to it are Grid2D's
 * correspondence_cost_cells(): row-major nx*iy+ix, 0 = unknown. */
typedef struct cmx_grid2d_limits {
  double resolution;
  double max_x, max_y;
  int32_t num_x_cells, num_y_cells;
  float min_correspondence_cost, max_correspondence_cost;
} cmx_grid2d_limits;

/* proto::RealTimeCorrelativeScanMatcherOptions
 * (mapping/proto/scan_matching/real_time_correlative_scan_matcher_options.proto). */
typedef struct cmx_rt_options {
  double linear_search_window;
  double angular_search_window;
  double translation_delta_cost_weight;
  double rotation_delta_cost_weight;
} cmx_rt_options;

/* proto::FastCorrelativeScanMatcherOptions2D. */
typedef struct cmx_fast2d_options {
  double linear_search_window;
  double angular_search_window;
  int32_t branch_and_bound_depth;
} cmx_fast2d_options;

/* proto::FastCorrelativeScanMatcherOptions3D. */
typedef struct cmx_fast3d_options {
  int32_t branch_and_bound_depth;
  int32_t full_resolution_depth;
  double min_rotational_score;
  double min_low_resolution_score;
  double linear_xy_search_window;
  double linear_z_search_window;
  double angular_search_window;
} cmx_fast3d_options;

/* Work counters of one call (or summed over a batch). */
typedef struct cmx_match_stats {
  int64_t candidates_scored;  /* every scored candidate, all depths; real-time matchers: the SEARCH
                                 SPACE the call covered (what the reference scores) -- how much of it
                                 the device summed is coarse_candidates */
  int64_t coarse_candidates;  /* lowest-resolution (or exhaustive) candidates; real-time 2D with block
                                 bounds: the bounds evaluated + the candidates summed behind them.
                                 Fast 2D from branch_and_bound_depth 5 on: every lowest-resolution
                                 candidate is counted (here and in candidates_scored), three
                                 neighbouring rotations share ONE sum that bounds all three */
  int64_t nodes_expanded;     /* branch-and-bound nodes whose children were scored */
  int32_t num_scans;          /* rotated scans */
  int32_t expansion_launches; /* launches inside expansion_ms (0: none timed) */
  /* The three *_ms fields are recorded only after cmx_debug_set("timing", 1) (the event packets
     cost a latency-bound call ~15 % of its wall time); otherwise they are 0.  The switch is
     process-wide: set it while no call is in flight. */
  double device_ms;           /* HIP-event time of the call's device work */
  double dominant_kernel_ms;  /* HIP-event time of the lowest-resolution (or exhaustive) scoring kernel(s) */
  double expansion_ms;        /* HIP-event time of the level-synchronous branch-and-bound expansion
                                 launches (fast 2D: ExpandWaveKernel, fast 3D: Expand3DKernel) */
  int64_t expansion_nodes;    /* nodes those launches took from their frontiers */
  int64_t expansion_lookups;  /* grid lookups they issued (64 per wave-wide gather instruction) */
  int64_t refined_candidates; /* real-time matchers: candidates the integer bounds could not decide,
                                 re-summed with exact integers */
  int64_t finalists;          /* real-time matchers: candidates scored with the reference's own
                                 sequential f32 sum (the only ones whose score is returned) */
} cmx_match_stats;

/* One flattened HybridGrid voxel (mapping/3d/hybrid_grid.h:304-372 Iterator):
 * cell index and raw uint16 probability value (0 never appears). */
typedef struct cmx_voxel { int32_t x, y, z; uint16_t value; uint16_t pad; } cmx_voxel;

typedef struct cmx_fast2d cmx_fast2d;   /* opaque FastCorrelativeScanMatcher2D */
typedef struct cmx_fast3d cmx_fast3d;   /* opaque FastCorrelativeScanMatcher3D */

/* ---- library / device ------------------------------------------------- */
const char* cmx_version(void);
/* sizeof(cmx_match_stats) of the LIBRARY.  The struct has grown between versions (0.1: 88 bytes;
 * since 0.2: 104) and carries no size field: a caller built against another header passes a
 * buffer of the wrong size -- compare once at start-up ("ABI"). */
int32_t cmx_sizeof_match_stats(void);
const char* cmx_status_string(cmx_status s);
/* Last error text of the calling thread ("" if none). */
const char* cmx_last_error(void);
/* Number of HIP devices visible (0 when there is none). */
int32_t cmx_device_count(void);
/* Streams: by default every call runs on a library-owned stream of `device`.
 * A caller that owns a HIP stream (e.g. torch.cuda.current_stream()) can make
 * the calling thread's subsequent calls on `device` use it instead;
 * pass NULL to go back to the library stream. */
cmx_status cmx_set_stream(int32_t device, void* hip_stream);

/* ---- real-time 2D ------------------------------------------------------ */
/* Returns the best score (already weighted by the delta cost) in *score and
 * the pose in *pose_estimate; always succeeds for valid inputs
 * (reference: CHECK_GT(score, 0)). */
cmx_status cmx_rt2d_match(const cmx_rt_options* options, const cmx_grid2d_limits* limits,
                          const uint16_t* cells, const cmx_pose2d* initial_pose_estimate,
                          const float* point_cloud_xyz, int32_t num_points, int32_t device,
                          double* score, cmx_pose2d* pose_estimate, cmx_match_stats* stats);
/* TSDF2D branch of the same method (SM2/real_time_correlative_scan_matcher_2d.cc:38-59,
 * mapping/internal/2d/tsdf_2d.cc:88-98): `tsd_cells` are the grid's
 * correspondence_cost_cells, `weight_cells` its weight plane (both uint16,
 * 0 = unknown, bit 15 = update marker), `truncation_distance` / `max_weight`
 * the TSDValueConverter ranges.  A TSDF may score 0 everywhere; the first
 * candidate then wins, as std::max_element does in the reference. */
cmx_status cmx_rt2d_match_tsdf(const cmx_rt_options* options, const cmx_grid2d_limits* limits,
                               const uint16_t* tsd_cells, const uint16_t* weight_cells,
                               float truncation_distance, float max_weight,
                               const cmx_pose2d* initial_pose_estimate,
                               const float* point_cloud_xyz, int32_t num_points, int32_t device,
                               double* score, cmx_pose2d* pose_estimate, cmx_match_stats* stats);

/* RealTimeCorrelativeScanMatcher2D::ScoreCandidates, the method the reference keeps "visible for
 * testing" (SM2/real_time_correlative_scan_matcher_2d.h:70-76, .cc:147-175), so that its unit
 * tests run against the device: ANY candidate list over caller-made discrete scans.
 * cmx_candidate2d mirrors Candidate2D (SM2/correlative_scan_matcher_2d.h:69-98): the caller fills
 * everything but `score`.  discrete_scans_xy holds the (x, y) cell indices of all scans back to
 * back, scan s being points scan_begin[s] .. scan_begin[s + 1] - 1.  weight_cells == NULL: a
 * ProbabilityGrid; otherwise the TSDF2D planes as in cmx_rt2d_match_tsdf. */
typedef struct cmx_candidate2d {
  int32_t scan_index, x_index_offset, y_index_offset;
  float score;                 /* out: weighted by the delta costs */
  double x, y, orientation;
} cmx_candidate2d;
cmx_status cmx_rt2d_score_candidates(const cmx_rt_options* options,
                                     const cmx_grid2d_limits* limits, const uint16_t* cells,
                                     const uint16_t* weight_cells, float truncation_distance,
                                     float max_weight, const int32_t* discrete_scans_xy,
                                     const int32_t* scan_begin, int32_t num_scans,
                                     cmx_candidate2d* candidates, int32_t num_candidates,
                                     int32_t device);

/* ---- device-resident probability grid ------------------ */
/* The active submap's ProbabilityGrid kept in HBM: LocalTrajectoryBuilder2D's per-scan
 * pair Match() -> InsertRangeData() (mapping/internal/2d/local_trajectory_builder_2d.cc:
 * 78-80, :288-289) then moves only the scan across PCIe.
 *   cmx_grid2d_create     ProbabilityGrid(limits) (all unknown) or a copy of `cells`
 *   cmx_grid2d_insert     ProbabilityGridRangeDataInserter2D::Insert + FinishUpdate
 *                         (mapping/2d/probability_grid_range_data_inserter_2d.cc:33-96,
 *                          mapping/internal/2d/ray_to_pixel_mask.cc:34-156); grows the
 *                         limits like GrowAsNeeded / Grid2D::GrowLimits. Points are in the
 *                         map frame (range data already transformed), xyz triples.
 *   cmx_grid2d_crop       grid = grid->ComputeCroppedGrid(): what Submap2D::Finish does before the
 *                         loop-closure matcher is built (mapping/2d/submap_2d.cc:146-149,
 *                         mapping/2d/probability_grid.cc:90-106, grid_2d.cc:104-114)
 *   cmx_rt2d_match_grid   RealTimeCorrelativeScanMatcher2D::Match on that grid
 *   cmx_fast2d_create_from_grid  FastCorrelativeScanMatcher2D of the (finished) grid */
typedef struct cmx_grid2d cmx_grid2d;
cmx_status cmx_grid2d_create(const cmx_grid2d_limits* limits, const uint16_t* cells_or_null,
                             int32_t device, cmx_grid2d** out);
void cmx_grid2d_destroy(cmx_grid2d* grid);
cmx_status cmx_grid2d_get_limits(const cmx_grid2d* grid, cmx_grid2d_limits* limits);
cmx_status cmx_grid2d_download(const cmx_grid2d* grid, uint16_t* cells);
cmx_status cmx_grid2d_crop(cmx_grid2d* grid);
cmx_status cmx_grid2d_insert(cmx_grid2d* grid, const float* origin_xy, const float* returns_xyz,
                             int32_t num_returns, const float* misses_xyz, int32_t num_misses,
                             float hit_probability, float miss_probability,
                             int32_t insert_free_space);
cmx_status cmx_rt2d_match_grid(const cmx_rt_options* options, const cmx_grid2d* grid,
                               const cmx_pose2d* initial_pose_estimate,
                               const float* point_cloud_xyz, int32_t num_points, double* score,
                               cmx_pose2d* pose_estimate, cmx_match_stats* stats);

/* `num_matches` independent real-time matches (one per trajectory / robot: scan i against
 * grid i around pose i) in one set of kernel launches.  A single match occupies 81 of the
 * chip's 8192 wave slots and is bound by the latency of its sequential f32 sums; batching
 * is what fills the machine. */
cmx_status cmx_rt2d_match_grid_batch(const cmx_rt_options* options,
                                     const cmx_grid2d* const* grids, int32_t num_matches,
                                     const cmx_pose2d* initial_pose_estimates,
                                     const float* const* point_clouds_xyz,
                                     const int32_t* num_points, double* scores,
                                     cmx_pose2d* pose_estimates, cmx_match_stats* stats);

/* The same batch with the scans already in HBM (cmx_cloud_upload, declared below): per call
 * only the initial poses, the per-scan rotation tables (libm, host) and the results cross PCIe.
 * The grid's staged image (quantised cells + halo, what the kernel copies into LDS) is kept
 * with the cmx_grid2d and rebuilt only after the grid changed. */
typedef struct cmx_cloud cmx_cloud;
cmx_status cmx_rt2d_match_grid_batch_resident(const cmx_rt_options* options,
                                              const cmx_grid2d* const* grids,
                                              int32_t num_matches,
                                              const cmx_pose2d* initial_pose_estimates,
                                              const cmx_cloud* const* clouds, double* scores,
                                              cmx_pose2d* pose_estimates, cmx_match_stats* stats);

/* ---- fast 2D (branch and bound) ---------------------------------------- */
/* Uploads the grid and builds the PrecomputationGridStack2D on `device`
 * (SM2/fast_correlative_scan_matcher_2d.cc:171-186). */
cmx_status cmx_fast2d_create(const cmx_fast2d_options* options, const cmx_grid2d_limits* limits,
                             const uint16_t* cells, int32_t device, cmx_fast2d** out);
cmx_status cmx_fast2d_create_from_grid(const cmx_fast2d_options* options, const cmx_grid2d* grid,
                                       cmx_fast2d** out);
void cmx_fast2d_destroy(cmx_fast2d* matcher);

cmx_status cmx_fast2d_match(const cmx_fast2d* matcher, const cmx_pose2d* initial_pose_estimate,
                            const float* point_cloud_xyz, int32_t num_points, float min_score,
                            int32_t* found, float* score, cmx_pose2d* pose_estimate,
                            cmx_match_stats* stats);
cmx_status cmx_fast2d_match_full_submap(const cmx_fast2d* matcher, const float* point_cloud_xyz,
                                        int32_t num_points, float min_score, int32_t* found,
                                        float* score, cmx_pose2d* pose_estimate,
                                        cmx_match_stats* stats);
/* One scan against `num_matchers` submaps (all on the same device), results
 * per submap; `stats` is the sum over the batch. */
cmx_status cmx_fast2d_match_full_submap_batch(const cmx_fast2d* const* matchers,
                                              int32_t num_matchers, const float* point_cloud_xyz,
                                              int32_t num_points, float min_score,
                                              int32_t* found, float* scores,
                                              cmx_pose2d* pose_estimates, cmx_match_stats* stats);

/* The general batch behind a ConstraintBuilder2D front
 * (constraints/constraint_builder_2d.cc:77-137, :194-236): one node's scan against many
 * submaps in ONE device batch, each entry either a windowed Match around its own
 * initial pose (match_full_submap[i] == 0: MaybeAddConstraint) or a MatchFullSubmap
 * (!= 0: MaybeAddGlobalConstraint), each with its own acceptance threshold
 * (min_score / global_localization_min_score). */
cmx_status cmx_fast2d_match_batch(const cmx_fast2d* const* matchers, int32_t num_matchers,
                                  const cmx_pose2d* initial_pose_estimates,
                                  const int32_t* match_full_submap, const float* min_scores,
                                  const float* point_cloud_xyz, int32_t num_points, int32_t* found,
                                  float* scores, cmx_pose2d* pose_estimates,
                                  cmx_match_stats* stats);

/* Device-resident variant for throughput measurement: the point cloud is
 * uploaded once, repeated matches touch no host buffer except the results. */
cmx_status cmx_cloud_upload(const float* point_cloud_xyz, int32_t num_points, int32_t device,
                            cmx_cloud** out);
void cmx_cloud_destroy(cmx_cloud* cloud);
cmx_status cmx_fast2d_match_full_submap_batch_resident(
    const cmx_fast2d* const* matchers, int32_t num_matchers, const cmx_cloud* cloud,
    float min_score, int32_t* found, float* scores, cmx_pose2d* pose_estimates,
    cmx_match_stats* stats);

/* ---- Ceres refinement, 2D -------------------------------- */
/* proto::CeresScanMatcherOptions2D + the three ceres_solver_options cartographer sets
 * (mapping/proto/scan_matching/ceres_scan_matcher_options_2d.proto,
 * common/internal/ceres_solver_options.cc:38-45; num_threads has no meaning here). */
typedef struct cmx_ceres2d_options {
  double occupied_space_weight;
  double translation_weight;
  double rotation_weight;
  int32_t use_nonmonotonic_steps;
  int32_t max_num_iterations;
} cmx_ceres2d_options;
/* The fields of ceres::Solver::Summary the callers and tests read. */
typedef struct cmx_ceres_summary {
  double initial_cost, final_cost;
  int32_t num_successful_steps, num_unsuccessful_steps;
  int32_t termination;   /* 0 CONVERGENCE, 1 NO_CONVERGENCE, 2 FAILURE */
  int32_t reserved;
} cmx_ceres_summary;
/* CeresScanMatcher2D::Match (SM2/ceres_scan_matcher_2d.h:51-56, .cc:63-107) on a probability
 * grid: bicubic occupied-space residuals + translation / rotation delta residuals, Ceres's
 * trust-region Levenberg-Marquardt with its default options.  `cells` as in cmx_rt2d_match. */
cmx_status cmx_ceres2d_match(const cmx_ceres2d_options* options, const cmx_grid2d_limits* limits,
                             const uint16_t* cells, const double* target_translation_xy,
                             const cmx_pose2d* initial_pose_estimate, const float* point_cloud_xyz,
                             int32_t num_points, int32_t device, cmx_pose2d* pose_estimate,
                             cmx_ceres_summary* summary);
/* The same on a grid resident in HBM: LocalTrajectoryBuilder2D::ScanMatch's
 * real-time match -> Ceres match pair (local_trajectory_builder_2d.cc:78-107) without the
 * grid crossing PCIe. */
cmx_status cmx_ceres2d_match_grid(const cmx_ceres2d_options* options, const cmx_grid2d* grid,
                                  const double* target_translation_xy,
                                  const cmx_pose2d* initial_pose_estimate,
                                  const float* point_cloud_xyz, int32_t num_points,
                                  cmx_pose2d* pose_estimate, cmx_ceres_summary* summary);
/* ConstraintBuilder2D::ComputeConstraint's refinement (constraints/constraint_builder_2d.cc:
 * 245-249) for the results of cmx_fast2d_match_batch, one launch for the whole batch, each
 * against the grid its matcher keeps in HBM: entry i refines pose_estimates_in[i] with target
 * translation pose_estimates_in[i].{x,y}; entries with found[i] == 0 (found may be NULL) are
 * passed through. */
cmx_status cmx_fast2d_refine_batch(const cmx_ceres2d_options* options,
                                   const cmx_fast2d* const* matchers, int32_t num_matchers,
                                   const int32_t* found, const cmx_pose2d* pose_estimates_in,
                                   const float* point_cloud_xyz, int32_t num_points,
                                   cmx_pose2d* pose_estimates_out, cmx_ceres_summary* summaries);

/* ---- CeresScanMatcher3D ------------------------------------------- */
/* proto::CeresScanMatcherOptions3D (mapping/proto/scan_matching/ceres_scan_matcher_options_3d.proto)
 * + the ceres_solver_options cartographer sets; the per-pair IntensityCostFunctionOptions ride
 * with the pair they belong to (cmx_ceres3d_pair). */
typedef struct cmx_ceres3d_options {
  double occupied_space_weight[3];   /* one per (point cloud, hybrid grid) pair */
  double translation_weight;
  double rotation_weight;
  int32_t num_pairs;                 /* 1 .. 3: high-resolution, low-resolution, ... */
  int32_t only_optimize_yaw;
  int32_t use_nonmonotonic_steps;
  int32_t max_num_iterations;
} cmx_ceres3d_options;
/* One cell of an IntensityHybridGrid (mapping/3d/hybrid_grid.h:543-571): AverageIntensityData
 * {sum, count}; GetIntensity = sum / count, 0 where count == 0 or the cell is absent. */
typedef struct cmx_intensity_voxel {
  int32_t x, y, z;
  int32_t count;
  float sum;
} cmx_intensity_voxel;
/* CeresScanMatcher3D::PointCloudAndHybridGridsPointers (SM3/ceres_scan_matcher_3d.h:42-46); the
 * grid as the voxel list HybridGrid::Iterator yields. */
typedef struct cmx_ceres3d_pair {
  const float* point_cloud_xyz;
  int32_t num_points;
  float resolution;
  const cmx_voxel* voxels;
  int64_t num_voxels;
  /* IntensityCostFunction3D (SM3/intensity_cost_function_3d.h, ceres_scan_matcher_3d.cc:118-137):
   * `intensities` == NULL: the pair has no intensity_hybrid_grid (what ConstraintBuilder3D
   * passes).  Otherwise num_points intensities (PointCloud::intensities()), the intensity grid
   * (same resolution as the pair's hybrid grid) and its IntensityCostFunctionOptions; the block
   * carries ceres::HuberLoss(intensity_huber_scale). */
  const float* intensities;
  const cmx_intensity_voxel* intensity_voxels;
  int64_t num_intensity_voxels;
  double intensity_weight;
  double intensity_huber_scale;
  float intensity_threshold;
  int32_t reserved;
} cmx_ceres3d_pair;
/* CeresScanMatcher3D::Match (SM3/ceres_scan_matcher_3d.h:55-60, .cc:90-156): occupied-space
 * residuals through InterpolatedGrid (SM3/interpolated_grid.h), translation / rotation delta
 * residuals, quaternion (or yaw-only) local parameterization, Ceres's trust-region
 * Levenberg-Marquardt with its default options.  The rotation target is the initial rotation. */
cmx_status cmx_ceres3d_match(const cmx_ceres3d_options* options,
                             const double* target_translation_xyz,
                             const cmx_pose3d* initial_pose_estimate,
                             const cmx_ceres3d_pair* pairs, int32_t device,
                             cmx_pose3d* pose_estimate, cmx_ceres_summary* summary);

/* Introspection used by the parity tests (not needed by a caller). */
cmx_status cmx_fast2d_level_dims(const cmx_fast2d* matcher, int32_t level, int32_t* wide_x,
                                 int32_t* wide_y);
cmx_status cmx_fast2d_level_cells(const cmx_fast2d* matcher, int32_t level, uint8_t* out);
/* Prepared search of a (full-submap or windowed) match: rotated+discretised
 * scans [num_scans][n][2], shrunk linear bounds [num_scans][4]
 * (min_x,max_x,min_y,max_y) and the integer sums of every lowest-resolution
 * candidate in the reference's generation order (scan, x, y).  Any output may
 * be NULL; capacities are in elements. */
cmx_status cmx_fast2d_debug_prepare(const cmx_fast2d* matcher,
                                    const cmx_pose2d* initial_pose_estimate,
                                    const float* point_cloud_xyz, int32_t num_points,
                                    int32_t full_submap, int32_t* num_scans,
                                    double* angular_step, int32_t* discrete_xy,
                                    int64_t discrete_capacity, int32_t* bounds,
                                    int64_t bounds_capacity, int32_t* coarse_sums,
                                    int64_t sums_capacity, int64_t* num_coarse);

/* ---- real-time 3D ------------------------------------------------------ */
cmx_status cmx_rt3d_match(const cmx_rt_options* options, float grid_resolution,
                          const cmx_voxel* voxels, int64_t num_voxels,
                          const cmx_pose3d* initial_pose_estimate, const float* point_cloud_xyz,
                          int32_t num_points, int32_t device, float* score,
                          cmx_pose3d* pose_estimate, cmx_match_stats* stats);

/* ---- device-resident hybrid grid -------------------- */
/* The active 3D submap's HybridGrid kept in HBM as a dense uint16 brick:
 *   cmx_grid3d_create    HybridGrid(resolution) (mapping/3d/hybrid_grid.h:459-462)
 *   cmx_grid3d_insert    RangeDataInserter3D::Insert without intensities
 *                        (mapping/3d/range_data_inserter_3d.cc:27-52, :93-114): hits, then the last
 *                        `num_free_space_voxels` voxels of every ray as misses, FinishUpdate;
 *                        points in the map frame, xyz triples
 *   cmx_grid3d_info      resolution(), DynamicGrid::grid_size() (hybrid_grid.h:259,381-398) and
 *                        the number of known voxels
 *   cmx_grid3d_download  the voxels HybridGrid::Iterator yields (hybrid_grid.h:304-372), sorted
 *                        (z, y, x): the list cmx_rt3d_match / cmx_fast3d_create take */
typedef struct cmx_grid3d cmx_grid3d;
cmx_status cmx_grid3d_create(float resolution, int32_t device, cmx_grid3d** out);
void cmx_grid3d_destroy(cmx_grid3d* grid);
cmx_status cmx_grid3d_insert(cmx_grid3d* grid, const float* origin_xyz, const float* returns_xyz,
                             int32_t num_returns, float hit_probability, float miss_probability,
                             int32_t num_free_space_voxels);
cmx_status cmx_grid3d_info(const cmx_grid3d* grid, float* resolution, int32_t* grid_size,
                           int64_t* num_voxels);
cmx_status cmx_grid3d_download(const cmx_grid3d* grid, cmx_voxel* voxels, int64_t capacity,
                               int64_t* num_voxels);

/* RealTimeCorrelativeScanMatcher3D::Match on the resident grid: LocalTrajectoryBuilder3D's per-scan
 * pair Match() -> InsertRangeData() (mapping/internal/3d/local_trajectory_builder_3d.cc:96-108,
 * :344-347) then moves only the scan across PCIe.  Same result as cmx_rt3d_match on
 * cmx_grid3d_download's voxel list. */
cmx_status cmx_rt3d_match_grid(const cmx_rt_options* options, const cmx_grid3d* grid,
                               const cmx_pose3d* initial_pose_estimate,
                               const float* point_cloud_xyz, int32_t num_points, float* score,
                               cmx_pose3d* pose_estimate, cmx_match_stats* stats);

/* CeresScanMatcher3D::Match as LocalTrajectoryBuilder3D::ScanMatch calls it
 * (mapping/internal/3d/local_trajectory_builder_3d.cc:96-123) against the ACTIVE submap: pair k is
 * (point_clouds_xyz[k], num_points[k]) with the resident HybridGrid grids[k] (its resolution is
 * the grid's); options->num_pairs pairs, all grids on one device.  Nothing but the clouds is
 * uploaded.  Pairs with an intensity term: cmx_ceres3d_match_grids_intensity below. */
cmx_status cmx_ceres3d_match_grids(const cmx_ceres3d_options* options,
                                   const double* target_translation_xyz,
                                   const cmx_pose3d* initial_pose_estimate,
                                   const cmx_grid3d* const* grids,
                                   const float* const* point_clouds_xyz,
                                   const int32_t* num_points, cmx_pose3d* pose_estimate,
                                   cmx_ceres_summary* summary);

/* ---- IntensityHybridGrid in HBM -------------------------------------------
 * mapping/3d/hybrid_grid.h:543-571: AverageIntensityData {sum, count} per voxel.
 *   cmx_grid3d_insert_with_intensities  RangeDataInserter3D::Insert with an intensity grid
 *                                       (mapping/3d/range_data_inserter_3d.cc:93-114): hits and
 *                                       misses into `grid` exactly as cmx_grid3d_insert, then
 *                                       InsertIntensitiesIntoGrid (:54-70) -- returns whose
 *                                       intensity exceeds `intensity_threshold` are skipped, the
 *                                       others add to their voxel's count and, IN POINT ORDER,
 *                                       to its f32 sum (bit-identical to the reference's loop).
 *                                       `intensities` = PointCloud::intensities() of the returns
 *                                       (num_returns floats); NULL inserts none (:57).
 *   cmx_intensity_grid3d_download       the voxels with count > 0, (z, y, x) order.
 *   cmx_ceres3d_match_grids_intensity   cmx_ceres3d_match_grids with IntensityCostFunction3D
 *                                       terms (SM3/ceres_scan_matcher_3d.cc:118-137) read from
 *                                       resident intensity grids: terms[k].grid == NULL: pair k
 *                                       has none.  Bit-identical to cmx_ceres3d_match on the same
 *                                       voxels. */
typedef struct cmx_intensity_grid3d cmx_intensity_grid3d;
cmx_status cmx_intensity_grid3d_create(float resolution, int32_t device,
                                       cmx_intensity_grid3d** out);
void cmx_intensity_grid3d_destroy(cmx_intensity_grid3d* grid);
cmx_status cmx_grid3d_insert_with_intensities(cmx_grid3d* grid,
                                              cmx_intensity_grid3d* intensity_grid,
                                              const float* origin_xyz, const float* returns_xyz,
                                              const float* intensities, int32_t num_returns,
                                              float hit_probability, float miss_probability,
                                              int32_t num_free_space_voxels,
                                              float intensity_threshold);
cmx_status cmx_intensity_grid3d_download(const cmx_intensity_grid3d* grid,
                                         cmx_intensity_voxel* voxels, int64_t capacity,
                                         int64_t* num_voxels);
typedef struct cmx_ceres3d_intensity_term {
  cmx_intensity_grid3d* grid;   /* NULL: no intensity term for this pair */
  const float* intensities;     /* num_points[k] intensities of the pair's cloud */
  double weight;                /* IntensityCostFunctionOptions::weight */
  double huber_scale;           /* ... ::huber_scale */
  float intensity_threshold;    /* ... ::intensity_threshold */
  int32_t reserved;
} cmx_ceres3d_intensity_term;
cmx_status cmx_ceres3d_match_grids_intensity(const cmx_ceres3d_options* options,
                                             const double* target_translation_xyz,
                                             const cmx_pose3d* initial_pose_estimate,
                                             const cmx_grid3d* const* grids,
                                             const float* const* point_clouds_xyz,
                                             const int32_t* num_points,
                                             const cmx_ceres3d_intensity_term* terms,
                                             cmx_pose3d* pose_estimate,
                                             cmx_ceres_summary* summary);

/* ---- fast 3D ------------------------------------------------------------ */
/* hybrid_grid.h:137 grid_size(): 8*8*2^bits cells per axis of the dynamic
 * grid the voxels came from (needed by MatchFullSubmap's window). */
cmx_status cmx_fast3d_create(const cmx_fast3d_options* options, float resolution,
                             int32_t grid_size, const cmx_voxel* voxels, int64_t num_voxels,
                             float low_resolution, const cmx_voxel* low_resolution_voxels,
                             int64_t num_low_resolution_voxels,
                             const float* rotational_scan_matcher_histogram,
                             int32_t histogram_size, int32_t device, cmx_fast3d** out);
void cmx_fast3d_destroy(cmx_fast3d* matcher);

/* TrajectoryNode::Data (mapping/trajectory_node.h:45-63), the fields the
 * matcher reads. */
typedef struct cmx_node_data3d {
  double gravity_alignment[4];                 /* quaternion w,x,y,z */
  const float* high_resolution_point_cloud;    /* xyz */
  int32_t num_high_resolution_points;
  const float* low_resolution_point_cloud;     /* xyz */
  int32_t num_low_resolution_points;
  const float* rotational_scan_matcher_histogram;
  int32_t histogram_size;
} cmx_node_data3d;

/* FastCorrelativeScanMatcher3D::Result. */
typedef struct cmx_result3d {
  float score;
  cmx_pose3d pose_estimate;
  float rotational_score;
  float low_resolution_score;
} cmx_result3d;

cmx_status cmx_fast3d_match(const cmx_fast3d* matcher, const cmx_pose3d* global_node_pose,
                            const cmx_pose3d* global_submap_pose, const cmx_node_data3d* data,
                            float min_score, int32_t* found, cmx_result3d* result,
                            cmx_match_stats* stats);
cmx_status cmx_fast3d_match_full_submap(const cmx_fast3d* matcher,
                                        const double* global_node_rotation_wxyz,
                                        const double* global_submap_rotation_wxyz,
                                        const cmx_node_data3d* data, float min_score,
                                        int32_t* found, cmx_result3d* result,
                                        cmx_match_stats* stats);

/* ConstraintBuilder3D's fan-out for one node (constraints/constraint_builder_3d.cc:79-147):
 * `data` against num_pairs matchers, pair p a windowed Match around node_poses[p] /
 * submap_poses[p] or, where match_full_submap[p] != 0, a MatchFullSubmap with their rotations,
 * against its own threshold min_scores[p].  All pairs whose matchers live on one device are ONE
 * chain of launches on that device (every kernel takes the array of problems; frontier and
 * leaf lists are shared, bounds and seeds are per problem); found[p] / results[p] as in the
 * single calls; *stats summed. */
cmx_status cmx_fast3d_match_batch(const cmx_fast3d* const* matchers, int32_t num_pairs,
                                  const cmx_pose3d* node_poses, const cmx_pose3d* submap_poses,
                                  const int32_t* match_full_submap, const float* min_scores,
                                  const cmx_node_data3d* data, int32_t* found,
                                  cmx_result3d* results, cmx_match_stats* stats);
/* ConstraintBuilder3D::ComputeConstraint's refinement (constraints/constraint_builder_3d.cc:
 * 263-276) for the results of cmx_fast3d_match_batch: pair i runs CeresScanMatcher3D::Match
 * (target translation = pose_estimates_in[i].t, initial pose = pose_estimates_in[i]) with
 * {data's high-resolution cloud, matcher i's high-resolution grid} and {low-resolution cloud,
 * low-resolution grid}; both grids are the ones the matcher keeps in HBM since
 * cmx_fast3d_create.  One launch per device, one workgroup per pair.  options->num_pairs must
 * be 2.  Entries with found[i] == 0 (found may be NULL) are passed through. */
cmx_status cmx_fast3d_refine_batch(const cmx_ceres3d_options* options,
                                   const cmx_fast3d* const* matchers, int32_t num_pairs,
                                   const int32_t* found, const cmx_pose3d* pose_estimates_in,
                                   const cmx_node_data3d* data, cmx_pose3d* pose_estimates_out,
                                   cmx_ceres_summary* summaries);

/* ---- upstream point preparation ---------------------------------------- */
/* sensor::VoxelFilter(PointCloud, resolution) (sensor/internal/voxel_filter.cc:88-152): one
 * RANDOM point per voxel -- the reference's reservoir sample, reproduced exactly (the draws of
 * its default-seeded std::minstd_rand0 are located in the generator's stream by a prefix sum).
 * `filtered_xyz` needs room for num_points points; kept points stay in input order. */
cmx_status cmx_voxel_filter(const float* point_cloud_xyz, int32_t num_points, float resolution,
                            int32_t device, float* filtered_xyz, int32_t* num_filtered);
/* The same filter, returning WHICH points it kept (ascending indices into the input; room for
 * num_points of them): what the overloads of sensor::VoxelFilter over timed points and range
 * measurements (voxel_filter.cc:154-191) need to carry the points' payload along --
 * LocalTrajectoryBuilder3D filters TimedPointCloudOriginData::RangeMeasurement lists
 * (3d/local_trajectory_builder_3d.cc:158-159). */
cmx_status cmx_voxel_filter_indices(const float* point_cloud_xyz, int32_t num_points,
                                    float resolution, int32_t device, int32_t* kept_indices,
                                    int32_t* num_filtered);
/* sensor::AdaptiveVoxelFilter (voxel_filter.cc:30-75,193-198): range cut, then the search for
 * the voxel length that keeps at least min_num_points (proto::AdaptiveVoxelFilterOptions). */
cmx_status cmx_adaptive_voxel_filter(const float* point_cloud_xyz, int32_t num_points,
                                     float max_length, float min_num_points, float max_range,
                                     int32_t device, float* filtered_xyz, int32_t* num_filtered);
/* The same filter, returning WHICH points it kept (ascending indices into the input; room for
 * num_points of them), so that a caller can carry the points' payload along: sensor::PointCloud
 * keeps the intensities of the kept points (voxel_filter.cc:138-161,193-198), which
 * LocalTrajectoryBuilder3D needs when use_intensities is set
 * (3d/local_trajectory_builder_3d.cc:262,298). */
cmx_status cmx_adaptive_voxel_filter_indices(const float* point_cloud_xyz, int32_t num_points,
                                             float max_length, float min_num_points,
                                             float max_range, int32_t device,
                                             int32_t* kept_indices, int32_t* num_filtered);
/* RotationalScanMatcher::ComputeHistogram (SM3/rotational_scan_matcher.cc:164-177): slices of
 * 0.2 m, points sorted by angle around the slice centroid, one weighted vote per point pair.
 * Same accumulation order as the reference; atan2f is the device's (<= 1 ulp from libm's). */
cmx_status cmx_compute_histogram(const float* point_cloud_xyz, int32_t num_points,
                                 int32_t histogram_size, int32_t device, float* histogram);

/* ---- multi-GPU: one scan against submaps spread over the GPUs of a node ---------------- */
/* The (node, submap) searches of a ConstraintBuilder fan-out are independent
 * (constraints/constraint_builder_2d.cc:97-111, constraint_builder_3d.cc:79-142): submaps are
 * partitioned over the devices -- a matcher lives on the device it was created on
 * (cmx_comm_device_of gives the block partition) -- and every device searches its own block
 * concurrently, one host thread per device.  Results come back per pair, in the caller's order.
 * The node-wide best match (global localisation) is agreed on by ONE RCCL all-reduce(max) of a
 * packed 8-byte key over xGMI (score bits << 32 | 0xFFFFFFFF - index: equal scores resolve to
 * the lowest index whatever the partition).  RCCL is loaded by cmx_comm_init (dlopen). */
typedef struct cmx_comm cmx_comm;
/* devices == NULL: devices 0 .. num_devices-1. */
cmx_status cmx_comm_init(const int32_t* devices, int32_t num_devices, cmx_comm** out);
void cmx_comm_destroy(cmx_comm* comm);
int32_t cmx_comm_num_devices(const cmx_comm* comm);
/* 1 if the communicator's collective is RCCL's (several devices), 0 if it needs none (one
 * device: the key is its own). */
int32_t cmx_comm_uses_rccl(const cmx_comm* comm);
/* Device that owns item `index` of `num_items` under the contiguous block partition. */
int32_t cmx_comm_device_of(const cmx_comm* comm, int64_t index, int64_t num_items);
/* As cmx_fast2d_match_batch, the matchers spread over the communicator's devices.
 * best_index / best_score (may be NULL): the best found pair node-wide, -1 if none. */
cmx_status cmx_fast2d_match_sharded(cmx_comm* comm, const cmx_fast2d* const* matchers,
                                    int32_t num_matchers, const cmx_pose2d* initial_pose_estimates,
                                    const int32_t* match_full_submap, const float* min_scores,
                                    const float* point_cloud_xyz, int32_t num_points,
                                    int32_t* found, float* scores, cmx_pose2d* pose_estimates,
                                    int32_t* best_index, float* best_score,
                                    cmx_match_stats* stats);
/* As cmx_fast3d_match_batch (BASELINE config C5: one node against 256 submaps on 8 GPUs). */
cmx_status cmx_fast3d_match_sharded(cmx_comm* comm, const cmx_fast3d* const* matchers,
                                    int32_t num_pairs, const cmx_pose3d* node_poses,
                                    const cmx_pose3d* submap_poses,
                                    const int32_t* match_full_submap, const float* min_scores,
                                    const cmx_node_data3d* data, int32_t* found,
                                    cmx_result3d* results, int32_t* best_index, float* best_score,
                                    cmx_match_stats* stats);
/* The partition and the key, as plain host arithmetic (no device needed; rank processes that
 * shard with torch.distributed / MPI instead of cmx_comm use the same rules):
 * items [begin, end) of `rank`; key of the best found entry of a block (-1: none found). */
void cmx_shard_range(int64_t num_items, int32_t rank, int32_t world_size, int64_t* begin,
                     int64_t* end);
int64_t cmx_pack_best_key(const int32_t* found, const float* scores, int64_t num,
                          int64_t first_global_index);
void cmx_unpack_best_key(int64_t key, int32_t* found, float* score, int64_t* global_index);

/* Introspection used by the parity tests: one precomputation level as a dense
 * brick (x fastest) with the cell index of its first element. */
cmx_status cmx_fast3d_level_info(const cmx_fast3d* matcher, int32_t depth, int32_t* lo_xyz,
                                 int32_t* dims_xyz);
cmx_status cmx_fast3d_level_cells(const cmx_fast3d* matcher, int32_t depth, uint8_t* out);

#ifdef __cplusplus
}
#endif

#endif /* CARTOGRAPHER_MI355X_H_ */
