// ORACLE — TEST INFRASTRUCTURE ONLY.  The reference's real-time 3D matcher over candidate RANGES.
//
// RealTimeCorrelativeScanMatcher3D::Match (real_time_correlative_scan_matcher_3d.cc:34-53) is one
// sequential loop over GenerateExhaustiveSearchTransforms(): at BASELINE config C4 (65 536 points,
// 1.77 M candidates) that is 1.2e11 transformed points on one core.  This wrapper runs the SAME
// reference functions -- GenerateExhaustiveSearchTransforms, sensor::TransformPointCloud,
// ScoreCandidate, all compiled unmodified from /root/reference -- for ranges of the candidate
// list on several host threads and joins the ranges with the loop's own rule (`score >
// best_score`, candidates in generation order: the FIRST maximum wins).  Only the six lines of the
// loop body are restated here; `ref_rt3d_match` (ref_wrapper.cc) is the unthreaded original and
// tests/golden/make_rt3d_c4_golden.py checks the two against each other on small windows (and,
// with --full-match, on C4 itself).
//
// The two functions are private members: this translation unit reads the class definition with
// `private` spelled `public` (access specifiers do not change layout or code; the reference's own
// .cc is compiled separately, untouched).
#include <algorithm>
#include <cstdint>
#include <memory>
#include <thread>
#include <vector>

#include "cartographer/mapping/3d/hybrid_grid.h"
#include "cartographer/sensor/point_cloud.h"
#include "cartographer/transform/rigid_transform.h"
#define private public
#include "cartographer/mapping/internal/3d/scan_matching/real_time_correlative_scan_matcher_3d.h"
#undef private

namespace {
namespace cm = cartographer::mapping;
namespace sm = cartographer::mapping::scan_matching;
struct RefVoxelMt { int32_t x, y, z; uint16_t value; uint16_t pad; };   // = oracle::Voxel
}  // namespace

extern "C" {

// Returns the best score; pose7 = t xyz, q wxyz of the winning candidate; best_index = its
// position in the reference's generation order; num_candidates = size of that list.
// scores_out (optional, num_candidates floats) receives every candidate's weighted score.
// first_candidate / max_candidates (max_candidates <= 0: all): a contiguous RANGE of the
// generation order -- bench.py's bounded CPU sample of BASELINE config C4 (the whole search
// space is ten minutes on 8 cores); best_index then refers to the best of that range.
float ref_rt3d_match_mt(float resolution, const void* voxels, int64_t n, const double* init7,
                        const float* xyz, int npts, double lin, double ang, double tw, double rw,
                        int num_threads, double* pose7, int64_t* best_index,
                        int64_t* num_candidates, float* scores_out, int64_t first_candidate,
                        int64_t max_candidates) {
  auto grid = std::make_unique<cm::HybridGrid>(resolution);
  const auto* vox = static_cast<const RefVoxelMt*>(voxels);
  for (int64_t i = 0; i != n; ++i)
    *grid->mutable_value(Eigen::Array3i(vox[i].x, vox[i].y, vox[i].z)) = vox[i].value;
  sm::proto::RealTimeCorrelativeScanMatcherOptions options;
  options.set_linear_search_window(lin);
  options.set_angular_search_window(ang);
  options.set_translation_delta_cost_weight(tw);
  options.set_rotation_delta_cost_weight(rw);
  const sm::RealTimeCorrelativeScanMatcher3D matcher(options);
  cartographer::sensor::PointCloud cloud;
  for (int i = 0; i != npts; ++i)
    cloud.push_back({Eigen::Vector3f(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2])});
  const cartographer::transform::Rigid3d initial_pose_estimate(
      Eigen::Vector3d(init7[0], init7[1], init7[2]),
      Eigen::Quaterniond(init7[3], init7[4], init7[5], init7[6]));

  const std::vector<cartographer::transform::Rigid3f> transforms =
      matcher.GenerateExhaustiveSearchTransforms(grid->resolution(), cloud);
  const int64_t all = static_cast<int64_t>(transforms.size());
  if (num_candidates) *num_candidates = all;
  const int64_t range_begin = std::min(std::max<int64_t>(first_candidate, 0), all);
  const int64_t total = max_candidates > 0 ? std::min(max_candidates, all - range_begin)
                                           : all - range_begin;
  num_threads = std::max(1, num_threads);
  struct Best { float score = -1.f; int64_t index = -1; };
  std::vector<Best> best(num_threads);
  std::vector<std::thread> threads;
  for (int t = 0; t != num_threads; ++t) {
    threads.emplace_back([&, t] {
      const int64_t begin = range_begin + total * t / num_threads,
                    end = range_begin + total * (t + 1) / num_threads;
      Best b;
      for (int64_t i = begin; i != end; ++i) {
        // real_time_correlative_scan_matcher_3d.cc:42-51
        const cartographer::transform::Rigid3f candidate =
            initial_pose_estimate.cast<float>() * transforms[i];
        const float score = matcher.ScoreCandidate(
            *grid, cartographer::sensor::TransformPointCloud(cloud, candidate), transforms[i]);
        if (scores_out) scores_out[i] = score;
        if (score > b.score) { b.score = score; b.index = i; }
      }
      best[t] = b;
    });
  }
  for (std::thread& th : threads) th.join();
  Best win;
  for (const Best& b : best)          // ranges in generation order, strict '>': first maximum
    if (b.score > win.score) win = b;
  if (best_index) *best_index = win.index;
  if (win.index >= 0) {
    const cartographer::transform::Rigid3d pose =
        (initial_pose_estimate.cast<float>() * transforms[win.index]).cast<double>();
    pose7[0] = pose.translation().x(); pose7[1] = pose.translation().y();
    pose7[2] = pose.translation().z();
    pose7[3] = pose.rotation().w(); pose7[4] = pose.rotation().x();
    pose7[5] = pose.rotation().y(); pose7[6] = pose.rotation().z();
  }
  return win.score;
}

}  // extern "C"
