// The reference's own ConstraintBuilder2D -- constraint_builder_2d.cc, thread_pool.cc, task.cc,
// fixed_ratio_sampler.cc, the grid classes, compiled UNMODIFIED from the reference tree --
// running its loop-closure searches on the MI355X through the adapter bodies next to this file.
//
//   1. the scenario of ConstraintBuilder2DTest.CallsBack and .FindsConstraints
//      (mapping/internal/constraints/constraint_builder_2d_test.cc:58-112), same calls, same
//      expectations (gtest / gmock replaced by plain checks);
//   2. a realistic node: a 1000-point scan against a few 400 x 400 submaps read from a fixture
//      file (written by tests/test_dropin.py), local + global constraints, the constraint
//      transforms printed for the test to compare with the oracle's.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <memory>
#include <vector>

#include "cartographer/common/internal/testing/thread_pool_for_testing.h"
#include "cartographer/mapping/2d/probability_grid.h"
#include "cartographer/mapping/2d/submap_2d.h"
#include "cartographer/mapping/internal/constraints/constraint_builder_2d.h"

using namespace cartographer;
using mapping::constraints::ConstraintBuilder2D;

#define EXPECT(cond)                                                         \
  do {                                                                       \
    if (!(cond)) {                                                           \
      std::fprintf(stderr, "%s:%d: expectation failed: %s\n", __FILE__, __LINE__, #cond); \
      std::exit(1);                                                          \
    }                                                                        \
  } while (0)

namespace {

// pose_graph.lua constraint_builder defaults (configuration_files/pose_graph.lua:20-39) with
// the test's overrides (sampling_ratio 1, min scores 0).
mapping::constraints::proto::ConstraintBuilderOptions TestOptions(double min_score,
                                                                   double global_min_score) {
  mapping::constraints::proto::ConstraintBuilderOptions o;
  o.sampling_ratio_ = 1.;
  o.max_constraint_distance_ = 15.;
  o.min_score_ = min_score;
  o.global_localization_min_score_ = global_min_score;
  o.loop_closure_translation_weight_ = 1.1e4;
  o.loop_closure_rotation_weight_ = 1e5;
  o.log_matches_ = false;
  o.fast_.set_linear_search_window(7.);
  o.fast_.set_angular_search_window(30. * M_PI / 180.);
  o.fast_.set_branch_and_bound_depth(7);
  o.ceres_.set_occupied_space_weight(20.);
  o.ceres_.set_translation_weight(10.);
  o.ceres_.set_rotation_weight(1.);
  o.ceres_.mutable_ceres_solver_options()->set_use_nonmonotonic_steps(true);
  o.ceres_.mutable_ceres_solver_options()->set_max_num_iterations(10);
  o.ceres_.mutable_ceres_solver_options()->set_num_threads(1);
  return o;
}

void ReferenceTestScenario() {
  common::testing::ThreadPoolForTesting thread_pool;
  auto builder = std::make_unique<ConstraintBuilder2D>(TestOptions(0., 0.), &thread_pool);
  // CallsBack.
  EXPECT(builder->GetNumFinishedNodes() == 0);
  int calls = 0;
  size_t last_size = 99;
  builder->NotifyEndOfNode();
  builder->WhenDone([&](const ConstraintBuilder2D::Result& result) {
    ++calls;
    last_size = result.size();
  });
  thread_pool.WaitUntilIdle();
  EXPECT(calls == 1 && last_size == 0);
  EXPECT(builder->GetNumFinishedNodes() == 1);
  builder.reset(new ConstraintBuilder2D(TestOptions(0., 0.), &thread_pool));

  // FindsConstraints.
  mapping::TrajectoryNode::Data node_data;
  node_data.filtered_gravity_aligned_point_cloud.push_back({Eigen::Vector3f(0.1, 0.2, 0.3)});
  node_data.gravity_alignment = Eigen::Quaterniond::Identity();
  node_data.local_pose = transform::Rigid3d::Identity();
  mapping::SubmapId submap_id{0, 1};
  mapping::MapLimits map_limits(1., Eigen::Vector2d(2., 3.), mapping::CellLimits(100, 110));
  mapping::ValueConversionTables conversion_tables;
  mapping::Submap2D submap(
      Eigen::Vector2f(4.f, 5.f),
      std::make_unique<mapping::ProbabilityGrid>(map_limits, &conversion_tables),
      &conversion_tables);
  int expected_nodes = 0;
  for (int i = 0; i < 2; ++i) {
    EXPECT(builder->GetNumFinishedNodes() == expected_nodes);
    for (int j = 0; j < 2; ++j) {
      builder->MaybeAddConstraint(submap_id, &submap, mapping::NodeId{0, 0}, &node_data,
                                  transform::Rigid2d::Identity());
    }
    builder->MaybeAddGlobalConstraint(submap_id, &submap, mapping::NodeId{0, 0}, &node_data);
    builder->NotifyEndOfNode();
    thread_pool.WaitUntilIdle();
    EXPECT(builder->GetNumFinishedNodes() == ++expected_nodes);
    builder->NotifyEndOfNode();
    thread_pool.WaitUntilIdle();
    EXPECT(builder->GetNumFinishedNodes() == ++expected_nodes);
    size_t size = 0;
    bool all_inter_submap = true;
    builder->WhenDone([&](const ConstraintBuilder2D::Result& result) {
      size = result.size();
      for (const auto& c : result)
        all_inter_submap &= c.tag == mapping::PoseGraphInterface::Constraint::INTER_SUBMAP;
    });
    thread_pool.WaitUntilIdle();
    EXPECT(size == 3 && all_inter_submap);
    builder->DeleteScanMatcher(submap_id);
  }
  std::printf("reference scenario: CallsBack + FindsConstraints OK\n");
}

// Fixture: int32 num_submaps, nx, ny; double res; then per submap: double max_x, max_y,
// origin_x, origin_y, uint16 cells[nx * ny]; then int32 n, float xyz[3 n]; then double
// initial_relative_pose[3] (x, y, theta) for the local constraints.
void FixtureScenario(const char* path) {
  std::ifstream in(path, std::ios::binary);
  EXPECT(in.good());
  int32_t num = 0, nx = 0, ny = 0;
  double res = 0.;
  in.read(reinterpret_cast<char*>(&num), 4);
  in.read(reinterpret_cast<char*>(&nx), 4);
  in.read(reinterpret_cast<char*>(&ny), 4);
  in.read(reinterpret_cast<char*>(&res), 8);
  mapping::ValueConversionTables conversion_tables;
  std::vector<std::unique_ptr<mapping::Submap2D>> submaps;
  for (int k = 0; k < num; ++k) {
    double hdr[4];
    in.read(reinterpret_cast<char*>(hdr), 32);
    std::vector<uint16_t> cells(static_cast<size_t>(nx) * ny);
    in.read(reinterpret_cast<char*>(cells.data()), cells.size() * 2);
    mapping::proto::Grid2D proto;
    proto.mutable_limits()->set_resolution(res);
    proto.mutable_limits()->mutable_max()->set_x(hdr[0]);
    proto.mutable_limits()->mutable_max()->set_y(hdr[1]);
    proto.mutable_limits()->mutable_cell_limits()->set_num_x_cells(nx);
    proto.mutable_limits()->mutable_cell_limits()->set_num_y_cells(ny);
    for (uint16_t c : cells) proto.mutable_cells()->push_back(c);
    proto.set_min_correspondence_cost(mapping::kMinCorrespondenceCost);
    proto.set_max_correspondence_cost(mapping::kMaxCorrespondenceCost);
    proto.mutable_probability_grid_2d();
    submaps.emplace_back(new mapping::Submap2D(
        Eigen::Vector2f(static_cast<float>(hdr[2]), static_cast<float>(hdr[3])),
        std::make_unique<mapping::ProbabilityGrid>(proto, &conversion_tables),
        &conversion_tables));
  }
  int32_t n = 0;
  in.read(reinterpret_cast<char*>(&n), 4);
  std::vector<float> xyz(3 * static_cast<size_t>(n));
  in.read(reinterpret_cast<char*>(xyz.data()), xyz.size() * 4);
  double rel[3];
  in.read(reinterpret_cast<char*>(rel), 24);
  EXPECT(in.good());
  mapping::TrajectoryNode::Data node_data;
  for (int i = 0; i < n; ++i)
    node_data.filtered_gravity_aligned_point_cloud.push_back(
        {Eigen::Vector3f(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2])});
  node_data.gravity_alignment = Eigen::Quaterniond::Identity();
  node_data.local_pose = transform::Rigid3d::Identity();

  common::testing::ThreadPoolForTesting thread_pool;
  ConstraintBuilder2D builder(TestOptions(0.55, 0.6), &thread_pool);
  for (int k = 0; k < num; ++k) {
    builder.MaybeAddConstraint(mapping::SubmapId{0, k}, submaps[k].get(), mapping::NodeId{0, 7},
                               &node_data, transform::Rigid2d({rel[0], rel[1]}, rel[2]));
    builder.MaybeAddGlobalConstraint(mapping::SubmapId{0, k}, submaps[k].get(),
                                     mapping::NodeId{0, 7}, &node_data);
  }
  builder.NotifyEndOfNode();
  builder.WhenDone([&](const ConstraintBuilder2D::Result& result) {
    for (const auto& c : result) {
      const auto& t = c.pose.zbar_ij.translation();
      const auto& q = c.pose.zbar_ij.rotation();
      std::printf("constraint submap %d node %d t %.9f %.9f yaw %.9f tag %d\n",
                  c.submap_id.submap_index, c.node_id.node_index, t.x(), t.y(),
                  2. * std::atan2(q.z(), q.w()), static_cast<int>(c.tag));
    }
    std::printf("constraints %zu\n", result.size());
  });
  thread_pool.WaitUntilIdle();
  EXPECT(builder.GetNumFinishedNodes() == 1);
  for (int k = 0; k < num; ++k) builder.DeleteScanMatcher(mapping::SubmapId{0, k});
}

}  // namespace

int main(int argc, char** argv) {
  ReferenceTestScenario();
  if (argc > 1) FixtureScenario(argv[1]);
  return 0;
}
