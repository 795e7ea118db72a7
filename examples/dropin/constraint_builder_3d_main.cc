// The reference's own ConstraintBuilder3D -- constraint_builder_3d.cc, thread_pool.cc, task.cc,
// fixed_ratio_sampler.cc and the HybridGrid of mapping/3d/hybrid_grid.h, compiled UNMODIFIED from
// the reference tree -- running its loop-closure searches and refinements on the MI355X through
// the adapter bodies next to this file (scan_matchers_3d_mi355x.cc).
//
//   1. the scenario of ConstraintBuilder3DTest.CallsBack and .FindsConstraints
//      (mapping/internal/constraints/constraint_builder_3d_test.cc:61-122), same calls, same
//      expectations (gtest / gmock replaced by plain checks);
//   2. a realistic node against a few submaps read from a fixture file (written by
//      tests/test_dropin.py): local and global constraints, the constraint transforms printed
//      for the test to compare with the oracle's.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <memory>
#include <vector>

#include "cartographer/common/internal/testing/thread_pool_for_testing.h"
#include "cartographer/mapping/3d/submap_3d.h"
#include "cartographer/mapping/internal/constraints/constraint_builder_3d.h"

using namespace cartographer;
using mapping::constraints::ConstraintBuilder3D;

#define EXPECT(cond)                                                         \
  do {                                                                       \
    if (!(cond)) {                                                           \
      std::fprintf(stderr, "%s:%d: expectation failed: %s\n", __FILE__, __LINE__, #cond); \
      std::exit(1);                                                          \
    }                                                                        \
  } while (0)

namespace {

// pose_graph.lua constraint_builder defaults (configuration_files/pose_graph.lua:17-63).
mapping::constraints::proto::ConstraintBuilderOptions Defaults() {
  mapping::constraints::proto::ConstraintBuilderOptions o;
  o.sampling_ratio_ = 0.3;
  o.max_constraint_distance_ = 15.;
  o.min_score_ = 0.55;
  o.global_localization_min_score_ = 0.6;
  o.loop_closure_translation_weight_ = 1.1e4;
  o.loop_closure_rotation_weight_ = 1e5;
  o.log_matches_ = false;
  o.fast_3d_.set_branch_and_bound_depth(8);
  o.fast_3d_.set_full_resolution_depth(3);
  o.fast_3d_.set_min_rotational_score(0.77);
  o.fast_3d_.set_min_low_resolution_score(0.55);
  o.fast_3d_.set_linear_xy_search_window(5.);
  o.fast_3d_.set_linear_z_search_window(1.);
  o.fast_3d_.set_angular_search_window(15. * M_PI / 180.);
  o.ceres_3d_.add_occupied_space_weight(5.);
  o.ceres_3d_.add_occupied_space_weight(30.);
  o.ceres_3d_.set_translation_weight(10.);
  o.ceres_3d_.set_rotation_weight(1.);
  o.ceres_3d_.set_only_optimize_yaw(false);
  o.ceres_3d_.mutable_ceres_solver_options()->set_use_nonmonotonic_steps(false);
  o.ceres_3d_.mutable_ceres_solver_options()->set_max_num_iterations(10);
  o.ceres_3d_.mutable_ceres_solver_options()->set_num_threads(1);
  return o;
}

void ReferenceTestScenario() {
  // The test's overrides of the defaults (constraint_builder_3d_test.cc:43-50).
  auto options = Defaults();
  options.sampling_ratio_ = 1.;
  options.min_score_ = 0.;
  options.global_localization_min_score_ = 0.;
  options.fast_3d_.set_min_low_resolution_score(0.);
  options.fast_3d_.set_min_rotational_score(0.);
  common::testing::ThreadPoolForTesting thread_pool;
  auto builder = std::make_unique<ConstraintBuilder3D>(options, &thread_pool);
  // CallsBack.
  EXPECT(builder->GetNumFinishedNodes() == 0);
  int calls = 0;
  size_t last_size = 99;
  builder->NotifyEndOfNode();
  builder->WhenDone([&](const ConstraintBuilder3D::Result& result) {
    ++calls;
    last_size = result.size();
  });
  thread_pool.WaitUntilIdle();
  EXPECT(calls == 1 && last_size == 0);
  EXPECT(builder->GetNumFinishedNodes() == 1);
  builder.reset(new ConstraintBuilder3D(options, &thread_pool));

  // FindsConstraints: one point against an EMPTY submap; every search "finds" it (min scores 0).
  mapping::TrajectoryNode::Data node_data;
  node_data.gravity_alignment = Eigen::Quaterniond::Identity();
  node_data.high_resolution_point_cloud.push_back({Eigen::Vector3f(0.1f, 0.2f, 0.3f)});
  node_data.low_resolution_point_cloud.push_back({Eigen::Vector3f(0.1f, 0.2f, 0.3f)});
  node_data.rotational_scan_matcher_histogram = Eigen::VectorXf::Zero(3);
  node_data.local_pose = transform::Rigid3d::Identity();
  mapping::SubmapId submap_id{0, 1};
  mapping::Submap3D submap(0.1f, 0.1f, transform::Rigid3d::Identity(), Eigen::VectorXf::Zero(3));
  int expected_nodes = 0;
  for (int i = 0; i < 2; ++i) {
    EXPECT(builder->GetNumFinishedNodes() == expected_nodes);
    for (int j = 0; j < 2; ++j) {
      builder->MaybeAddConstraint(submap_id, &submap, mapping::NodeId{0, 0}, &node_data,
                                  transform::Rigid3d::Identity(), transform::Rigid3d::Identity());
    }
    builder->MaybeAddGlobalConstraint(submap_id, &submap, mapping::NodeId{0, 0}, &node_data,
                                      Eigen::Quaterniond::Identity(),
                                      Eigen::Quaterniond::Identity());
    builder->NotifyEndOfNode();
    thread_pool.WaitUntilIdle();
    EXPECT(builder->GetNumFinishedNodes() == ++expected_nodes);
    builder->NotifyEndOfNode();
    thread_pool.WaitUntilIdle();
    EXPECT(builder->GetNumFinishedNodes() == ++expected_nodes);
    size_t size = 0;
    bool all_inter_submap = true;
    builder->WhenDone([&](const ConstraintBuilder3D::Result& result) {
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

struct FileVoxel { int32_t x, y, z; uint16_t value, pad; };

void ReadGrid(std::ifstream& in, mapping::HybridGrid* grid) {
  int64_t n = 0;
  in.read(reinterpret_cast<char*>(&n), 8);
  std::vector<FileVoxel> voxels(static_cast<size_t>(n));
  in.read(reinterpret_cast<char*>(voxels.data()), voxels.size() * sizeof(FileVoxel));
  for (const FileVoxel& v : voxels)
    *grid->mutable_value(Eigen::Array3i(v.x, v.y, v.z)) = v.value;     // raw uint16, as stored
}

sensor::PointCloud ReadCloud(std::ifstream& in) {
  int32_t n = 0;
  in.read(reinterpret_cast<char*>(&n), 4);
  std::vector<float> xyz(3 * static_cast<size_t>(n));
  in.read(reinterpret_cast<char*>(xyz.data()), xyz.size() * 4);
  sensor::PointCloud cloud;
  for (int i = 0; i < n; ++i)
    cloud.push_back({Eigen::Vector3f(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2])});
  return cloud;
}

Eigen::VectorXf ReadHistogram(std::ifstream& in) {
  int32_t n = 0;
  in.read(reinterpret_cast<char*>(&n), 4);
  std::vector<float> h(static_cast<size_t>(n));
  in.read(reinterpret_cast<char*>(h.data()), h.size() * 4);
  Eigen::VectorXf v = Eigen::VectorXf::Zero(n);
  for (int i = 0; i < n; ++i) v[i] = h[i];
  return v;
}

// Fixture: int32 num_submaps; double options[9] (min_score, global min_score, depth,
// full_resolution_depth, min_rotational_score, min_low_resolution_score, xy window, z window,
// angular window); float high_resolution, low_resolution; per submap: histogram, high voxels,
// low voxels; then the node: high cloud, low cloud, histogram, double global_node_pose[7]
// (t, q wxyz).  Submaps are at the identity pose.
void FixtureScenario(const char* path) {
  std::ifstream in(path, std::ios::binary);
  EXPECT(in.good());
  int32_t num = 0;
  double opt[9];
  float res[2];
  in.read(reinterpret_cast<char*>(&num), 4);
  in.read(reinterpret_cast<char*>(opt), sizeof opt);
  in.read(reinterpret_cast<char*>(res), sizeof res);
  std::vector<std::unique_ptr<mapping::Submap3D>> submaps;
  for (int k = 0; k < num; ++k) {
    const Eigen::VectorXf histogram = ReadHistogram(in);
    submaps.emplace_back(
        new mapping::Submap3D(res[0], res[1], transform::Rigid3d::Identity(), histogram));
    ReadGrid(in, submaps.back()->mutable_high_resolution_hybrid_grid());
    ReadGrid(in, submaps.back()->mutable_low_resolution_hybrid_grid());
  }
  mapping::TrajectoryNode::Data node_data;
  node_data.gravity_alignment = Eigen::Quaterniond::Identity();
  node_data.high_resolution_point_cloud = ReadCloud(in);
  node_data.low_resolution_point_cloud = ReadCloud(in);
  node_data.rotational_scan_matcher_histogram = ReadHistogram(in);
  node_data.local_pose = transform::Rigid3d::Identity();
  double pose[7];
  in.read(reinterpret_cast<char*>(pose), sizeof pose);
  EXPECT(in.good());
  const transform::Rigid3d global_node_pose(
      Eigen::Vector3d(pose[0], pose[1], pose[2]),
      Eigen::Quaterniond(pose[3], pose[4], pose[5], pose[6]));

  auto options = Defaults();
  options.sampling_ratio_ = 1.;
  options.min_score_ = opt[0];
  options.global_localization_min_score_ = opt[1];
  options.fast_3d_.set_branch_and_bound_depth(static_cast<int>(opt[2]));
  options.fast_3d_.set_full_resolution_depth(static_cast<int>(opt[3]));
  options.fast_3d_.set_min_rotational_score(opt[4]);
  options.fast_3d_.set_min_low_resolution_score(opt[5]);
  options.fast_3d_.set_linear_xy_search_window(opt[6]);
  options.fast_3d_.set_linear_z_search_window(opt[7]);
  options.fast_3d_.set_angular_search_window(opt[8]);
  common::testing::ThreadPoolForTesting thread_pool;
  ConstraintBuilder3D builder(options, &thread_pool);
  for (int k = 0; k < num; ++k) {
    builder.MaybeAddConstraint(mapping::SubmapId{0, k}, submaps[k].get(), mapping::NodeId{0, 7},
                               &node_data, global_node_pose, transform::Rigid3d::Identity());
    builder.MaybeAddGlobalConstraint(mapping::SubmapId{0, k}, submaps[k].get(),
                                     mapping::NodeId{0, 7}, &node_data,
                                     global_node_pose.rotation(), Eigen::Quaterniond::Identity());
  }
  builder.NotifyEndOfNode();
  builder.WhenDone([&](const ConstraintBuilder3D::Result& result) {
    for (const auto& c : result) {
      const auto& t = c.pose.zbar_ij.translation();
      const auto& q = c.pose.zbar_ij.rotation();
      std::printf("constraint submap %d node %d t %.9f %.9f %.9f q %.9f %.9f %.9f %.9f tag %d\n",
                  c.submap_id.submap_index, c.node_id.node_index, t.x(), t.y(), t.z(), q.w(),
                  q.x(), q.y(), q.z(), static_cast<int>(c.tag));
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
