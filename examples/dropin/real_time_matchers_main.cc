// The scenarios of the reference's RealTimeCorrelativeScanMatcherTest
// (mapping/internal/2d/scan_matching/real_time_correlative_scan_matcher_2d_test.cc:34-212, all
// four ScoreCandidates cases, ProbabilityGrid and TSDF2D) and RealTimeCorrelativeScanMatcher3DTest
// (mapping/internal/3d/scan_matching/real_time_correlative_scan_matcher_3d_test.cc:34-121, seven
// initial poses), with gtest replaced by plain checks and the Lua dictionaries by the options'
// setters -- plus Match() itself on the 2D fixtures, which the reference's test does not call.
//
// The same file is linked twice (Makefile):
//   _build/real_time_matchers_reference   with the reference's own
//       real_time_correlative_scan_matcher_2d.cc / _3d.cc          (runs anywhere, CPU)
//   _build/real_time_matchers_mi355x      with real_time_matchers_mi355x.cc over the library
// Every number is printed as a hex float: the two outputs must be IDENTICAL, which
// tests/test_dropin.py checks against tests/golden/real_time_matchers_reference.txt.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>

#include "cartographer/mapping/2d/probability_grid.h"
#include "cartographer/mapping/2d/probability_grid_range_data_inserter_2d.h"
#include "cartographer/mapping/3d/hybrid_grid.h"
#include "cartographer/mapping/internal/2d/scan_matching/real_time_correlative_scan_matcher_2d.h"
#include "cartographer/mapping/internal/2d/tsdf_2d.h"
#include "cartographer/mapping/internal/2d/tsdf_range_data_inserter_2d.h"
#include "cartographer/mapping/internal/3d/scan_matching/real_time_correlative_scan_matcher_3d.h"
#include "cartographer/sensor/point_cloud.h"
#include "cartographer/transform/transform.h"

using namespace cartographer;
using namespace cartographer::mapping;
using namespace cartographer::mapping::scan_matching;

#define EXPECT(cond)                                                         \
  do {                                                                       \
    if (!(cond)) {                                                           \
      std::fprintf(stderr, "%s:%d: expectation failed: %s\n", __FILE__, __LINE__, #cond); \
      std::exit(1);                                                          \
    }                                                                        \
  } while (0)

namespace {

scan_matching::proto::RealTimeCorrelativeScanMatcherOptions Options(double linear, double angular,
                                                                    double tw, double rw) {
  scan_matching::proto::RealTimeCorrelativeScanMatcherOptions o;
  o.set_linear_search_window(linear);
  o.set_angular_search_window(angular);
  o.set_translation_delta_cost_weight(tw);
  o.set_rotation_delta_cost_weight(rw);
  return o;
}

struct Fixture2D {
  Fixture2D() : matcher(Options(0.6, 0.16, 0., 0.)) {
    for (const Eigen::Vector3f& p :
         {Eigen::Vector3f{0.025f, 0.175f, 0.f}, Eigen::Vector3f{-0.025f, 0.175f, 0.f},
          Eigen::Vector3f{-0.075f, 0.175f, 0.f}, Eigen::Vector3f{-0.125f, 0.175f, 0.f},
          Eigen::Vector3f{-0.125f, 0.125f, 0.f}, Eigen::Vector3f{-0.125f, 0.075f, 0.f},
          Eigen::Vector3f{-0.125f, 0.025f, 0.f}})
      point_cloud.push_back({p});
  }
  void SetUpTSDF() {
    grid = std::make_unique<TSDF2D>(MapLimits(0.05, Eigen::Vector2d(0.3, 0.5), CellLimits(20, 20)),
                                    0.3, 1.0, &conversion_tables);
    mapping::proto::TSDFRangeDataInserterOptions2D o;
    o.set_truncation_distance(0.3);
    o.set_maximum_weight(10.);
    o.set_update_free_space(false);
    o.mutable_normal_estimation_options()->set_num_normal_samples(4);
    o.mutable_normal_estimation_options()->set_sample_radius(0.5);
    o.set_project_sdf_distance_to_scan_normal(true);
    o.set_update_weight_range_exponent(0);
    o.set_update_weight_angle_scan_normal_to_ray_kernel_bandwidth(0.5);
    o.set_update_weight_distance_cell_to_hit_kernel_bandwidth(0.5);
    inserter = std::make_unique<TSDFRangeDataInserter2D>(o);
    inserter->Insert(sensor::RangeData{Eigen::Vector3f(0.5f, -0.5f, 0.f), point_cloud, {}},
                     grid.get());
    grid->FinishUpdate();
  }
  void SetUpProbabilityGrid() {
    grid = std::make_unique<ProbabilityGrid>(
        MapLimits(0.05, Eigen::Vector2d(0.05, 0.25), CellLimits(6, 6)), &conversion_tables);
    mapping::proto::ProbabilityGridRangeDataInserterOptions2D o;
    o.set_insert_free_space(true);
    o.set_hit_probability(0.7);
    o.set_miss_probability(0.4);
    inserter = std::make_unique<ProbabilityGridRangeDataInserter2D>(o);
    inserter->Insert(sensor::RangeData{Eigen::Vector3f::Zero(), point_cloud, {}}, grid.get());
    grid->FinishUpdate();
  }
  // One candidate scored through ScoreCandidates, as every case of the reference's test does.
  Candidate2D Score(int x_offset, int y_offset) {
    const SearchParameters parameters(0, 0, 0., 0.);
    const std::vector<sensor::PointCloud> scans = GenerateRotatedScans(point_cloud, parameters);
    const std::vector<DiscreteScan2D> discrete_scans =
        DiscretizeScans(grid->limits(), scans, Eigen::Translation2f::Identity());
    std::vector<Candidate2D> candidates;
    candidates.emplace_back(0, x_offset, y_offset, parameters);
    matcher.ScoreCandidates(*grid, discrete_scans, parameters, &candidates);
    EXPECT(candidates[0].scan_index == 0);
    EXPECT(candidates[0].x_index_offset == x_offset);
    EXPECT(candidates[0].y_index_offset == y_offset);
    return candidates[0];
  }
  ValueConversionTables conversion_tables;
  std::unique_ptr<Grid2D> grid;
  std::unique_ptr<RangeDataInserterInterface> inserter;
  sensor::PointCloud point_cloud;
  RealTimeCorrelativeScanMatcher2D matcher;
};

void Match2D(const char* name, Fixture2D* f, double tw, double rw) {
  // Match() on the test's fixture: a window of 3 cells and 0.16 rad around an offset pose.
  const RealTimeCorrelativeScanMatcher2D matcher(Options(0.15, 0.16, tw, rw));
  for (const transform::Rigid2d& initial :
       {transform::Rigid2d({0., 0.}, 0.), transform::Rigid2d({0.06, -0.04}, 0.05),
        transform::Rigid2d({-0.05, 0.1}, -0.1)}) {
    transform::Rigid2d pose;
    const double score = matcher.Match(initial, f->point_cloud, *f->grid, &pose);
    std::printf("2d %s match tw=%g rw=%g from (%g %g %g): score %a pose %a %a %a\n", name, tw, rw,
                initial.translation().x(), initial.translation().y(), initial.rotation().angle(),
                score, pose.translation().x(), pose.translation().y(), pose.rotation().angle());
  }
}

void Scenarios2D() {
  {
    Fixture2D f;                                          // ScorePerfect...ProbabilityGrid
    f.SetUpProbabilityGrid();
    const Candidate2D c = f.Score(0, 0);
    EXPECT(std::abs(0.7 - c.score) < 1e-2);               // every point aligns perfectly
    std::printf("2d probability_grid perfect: score %a\n", c.score);
    const Candidate2D p = f.Score(0, 1);                  // ScorePartiallyCorrect...ProbabilityGrid
    EXPECT(0.7 * 3. / 7. < p.score);
    EXPECT(0.7 > p.score);
    std::printf("2d probability_grid partial: score %a\n", p.score);
    Match2D("probability_grid", &f, 0., 0.);
    Match2D("probability_grid", &f, 0.5, 0.3);
  }
  {
    Fixture2D f;                                          // ScorePerfect...TSDF
    f.SetUpTSDF();
    const Candidate2D c = f.Score(0, 0);
    EXPECT(std::abs(1.0 - c.score) < 1e-1);
    EXPECT(0.95 < c.score);
    std::printf("2d tsdf perfect: score %a\n", c.score);
    const Candidate2D p = f.Score(0, 1);                  // ScorePartiallyCorrect...TSDF
    EXPECT(1.0 - 4. / (7. * 6.) < p.score);
    EXPECT(1.0 > p.score);
    std::printf("2d tsdf partial: score %a\n", p.score);
    Match2D("tsdf", &f, 0., 0.);
    Match2D("tsdf", &f, 0.5, 0.3);
  }
}

// transform::IsNearly (rigid_transform_test_helpers.h:34-43): Eigen isApprox on the homogeneous
// matrices, relative Frobenius norm.
bool IsNearly(const transform::Rigid3d& a, const transform::Rigid3d& b, double eps) {
  double diff = 0., na = 0., nb = 0.;
  const auto matrix = [](const transform::Rigid3d& t, double m[16]) {
    const Eigen::Vector3d columns[3] = {t.rotation() * Eigen::Vector3d(1., 0., 0.),
                                        t.rotation() * Eigen::Vector3d(0., 1., 0.),
                                        t.rotation() * Eigen::Vector3d(0., 0., 1.)};
    for (int i = 0; i != 3; ++i) {
      for (int j = 0; j != 3; ++j) m[4 * i + j] = columns[j][i];
      m[4 * i + 3] = t.translation()[i];
    }
    m[12] = m[13] = m[14] = 0.;
    m[15] = 1.;
  };
  double ma[16], mb[16];
  matrix(a, ma);
  matrix(b, mb);
  for (int i = 0; i != 16; ++i) {
    diff += (ma[i] - mb[i]) * (ma[i] - mb[i]);
    na += ma[i] * ma[i];
    nb += mb[i] * mb[i];
  }
  return diff <= eps * eps * std::min(na, nb);
}

void Scenarios3D() {
  HybridGrid hybrid_grid(0.1f);
  const transform::Rigid3d expected_pose(Eigen::Vector3d(-1., 0., 0.),
                                         Eigen::Quaterniond::Identity());
  sensor::PointCloud point_cloud;
  for (const Eigen::Vector3f& point :
       {Eigen::Vector3f(-3.f, 2.f, 0.f), Eigen::Vector3f(-4.f, 2.f, 0.f),
        Eigen::Vector3f(-5.f, 2.f, 0.f), Eigen::Vector3f(-6.f, 2.f, 0.f),
        Eigen::Vector3f(-6.f, 3.f, 1.f), Eigen::Vector3f(-6.f, 4.f, 2.f),
        Eigen::Vector3f(-7.f, 3.f, 1.f)}) {
    point_cloud.push_back({point});
    hybrid_grid.SetProbability(hybrid_grid.GetCellIndex(expected_pose.cast<float>() * point), 1.);
  }
  const RealTimeCorrelativeScanMatcher3D matcher(Options(0.3, M_PI / 180., 1e-1, 1.));
  const double a = 0.8 / 180. * M_PI;
  struct Case { const char* name; transform::Rigid3d initial; };
  const Case cases[] = {
      {"PerfectEstimate", transform::Rigid3d::Translation(Eigen::Vector3d(-1., 0., 0.))},
      {"AlongX", transform::Rigid3d::Translation(Eigen::Vector3d(-0.8, 0., 0.))},
      {"AlongZ", transform::Rigid3d::Translation(Eigen::Vector3d(-1., 0., -0.2))},
      {"AlongXYZ", transform::Rigid3d::Translation(Eigen::Vector3d(-0.9, -0.2, 0.2))},
      {"RotationAroundX", transform::Rigid3d(Eigen::Vector3d(-1., 0., 0.),
                                             Eigen::AngleAxisd(a, Eigen::Vector3d(1., 0., 0.)))},
      {"RotationAroundY", transform::Rigid3d(Eigen::Vector3d(-1., 0., 0.),
                                             Eigen::AngleAxisd(a, Eigen::Vector3d(0., 1., 0.)))},
      {"RotationAroundYZ", transform::Rigid3d(Eigen::Vector3d(-1., 0., 0.),
                                              Eigen::AngleAxisd(a, Eigen::Vector3d(0., 1., 1.)))},
  };
  for (const Case& c : cases) {
    transform::Rigid3d pose;
    const float score = matcher.Match(c.initial, point_cloud, hybrid_grid, &pose);
    EXPECT(IsNearly(pose, expected_pose, 1e-3));
    std::printf("3d %s: score %a pose %a %a %a %a %a %a %a\n", c.name, score,
                pose.translation().x(), pose.translation().y(), pose.translation().z(),
                pose.rotation().w(), pose.rotation().x(), pose.rotation().y(), pose.rotation().z());
  }
}

}  // namespace

int main() {
  Scenarios2D();
  Scenarios3D();
  std::printf("reference scenarios: RealTimeCorrelativeScanMatcherTest x4 + "
              "RealTimeCorrelativeScanMatcher3DTest x7 OK\n");
  return 0;
}
