// See batched_constraint_builder_3d.h.  Reference behaviour cited as CB3 = mapping/internal/
// constraints/constraint_builder_3d.cc.
#include "batched_constraint_builder_3d.h"

#include <cstdio>
#include <cstdlib>
#include <set>
#include <string>

#include "cartographer/mapping/3d/hybrid_grid.h"

namespace cartographer {
namespace mapping {
namespace constraints {
namespace {

void CheckOk(cmx_status status, const char* what) {
  if (status == CMX_OK) return;
  std::fprintf(stderr, "Check failed: %s: %s (%s)\n", what, cmx_status_string(status),
               cmx_last_error());
  std::abort();
}
void Require(bool condition, const char* what) {
  if (condition) return;
  std::fprintf(stderr, "Check failed: %s\n", what);
  std::abort();
}

std::vector<cmx_voxel> Flatten(const HybridGrid& grid) {     // the HybridGrid::Iterator walk
  std::vector<cmx_voxel> out;
  for (auto it = HybridGrid::Iterator(grid); !it.Done(); it.Next()) {
    const Eigen::Array3i index = it.GetCellIndex();
    out.push_back(cmx_voxel{index.x(), index.y(), index.z(), it.GetValue(), 0});
  }
  return out;
}
std::vector<float> Flatten(const sensor::PointCloud& cloud) {
  std::vector<float> xyz;
  xyz.reserve(3 * cloud.size());
  for (const sensor::RangefinderPoint& p : cloud) {
    xyz.push_back(p.position.x());
    xyz.push_back(p.position.y());
    xyz.push_back(p.position.z());
  }
  return xyz;
}
std::vector<float> Flatten(const Eigen::VectorXf& v) {
  std::vector<float> out;
  for (int i = 0; i != v.size(); ++i) out.push_back(v[i]);
  return out;
}
cmx_pose3d PoseOf(const transform::Rigid3d& t) {
  return cmx_pose3d{{t.translation().x(), t.translation().y(), t.translation().z()},
                    {t.rotation().w(), t.rotation().x(), t.rotation().y(), t.rotation().z()}};
}
transform::Rigid3d PoseFrom(const cmx_pose3d& p) {
  return transform::Rigid3d(Eigen::Vector3d(p.t[0], p.t[1], p.t[2]),
                            Eigen::Quaterniond(p.q[0], p.q[1], p.q[2], p.q[3]));
}
// The GPUs the builder spreads its submaps over: CMX_DEVICES="0,1,2,3" (or "all"), else the one
// device CMX_DEVICE names (default 0).
std::vector<int32_t> Devices() {
  std::vector<int32_t> devices;
  if (const char* list = std::getenv("CMX_DEVICES")) {
    if (std::string(list) == "all") {
      for (int d = 0; d < cmx_device_count(); ++d) devices.push_back(d);
    } else {
      for (const char* at = list; *at;) {
        char* end = nullptr;
        const long d = std::strtol(at, &end, 10);
        if (end == at) break;
        devices.push_back(static_cast<int32_t>(d));
        at = *end == ',' ? end + 1 : end;
      }
    }
  }
  if (devices.empty()) {
    const char* e = std::getenv("CMX_DEVICE");
    devices.push_back(e ? std::atoi(e) : 0);
  }
  return devices;
}

}  // namespace

ConstraintBuilder3D::ConstraintBuilder3D(const proto::ConstraintBuilderOptions& options,
                                         common::ThreadPoolInterface* const thread_pool)
    : options_(options), thread_pool_(thread_pool),
      when_done_task_(std::make_unique<common::Task>()) {
  const std::vector<int32_t> devices = Devices();
  CheckOk(cmx_comm_init(devices.data(), static_cast<int32_t>(devices.size()), &comm_),
          "cmx_comm_init");
}

ConstraintBuilder3D::~ConstraintBuilder3D() {
  absl::MutexLock locker(&mutex_);
  Require(when_done_task_->GetState() == common::Task::NEW, "a WhenDone task is in flight");
  Require(pending_.empty(), "NotifyEndOfNode() was not called for the last node");
  Require(constraints_.empty(), "WhenDone() was not called");
  Require(num_started_nodes_ == num_finished_nodes_, "nodes still being computed");
  Require(when_done_ == nullptr, "WhenDone callback pending");
  matchers_.clear();
  cmx_comm_destroy(comm_);
}

// CB3:79-114: distance between the GLOBAL poses, then the per-submap sampler.
void ConstraintBuilder3D::MaybeAddConstraint(const SubmapId& submap_id, const Submap3D* const submap,
                                             const NodeId& node_id,
                                             const TrajectoryNode::Data* const constant_data,
                                             const transform::Rigid3d& global_node_pose,
                                             const transform::Rigid3d& global_submap_pose) {
  if ((global_node_pose.translation() - global_submap_pose.translation()).norm() >
      options_.max_constraint_distance()) {
    return;
  }
  if (!per_submap_sampler_
           .emplace(std::piecewise_construct, std::forward_as_tuple(submap_id),
                    std::forward_as_tuple(options_.sampling_ratio()))
           .first->second.Pulse()) {
    return;
  }
  Enqueue(submap_id, submap, node_id, constant_data, global_node_pose, global_submap_pose, false);
}

// CB3:116-147.
void ConstraintBuilder3D::MaybeAddGlobalConstraint(const SubmapId& submap_id,
                                                   const Submap3D* const submap,
                                                   const NodeId& node_id,
                                                   const TrajectoryNode::Data* const constant_data,
                                                   const Eigen::Quaterniond& global_node_rotation,
                                                   const Eigen::Quaterniond& global_submap_rotation) {
  Enqueue(submap_id, submap, node_id, constant_data,
          transform::Rigid3d::Rotation(global_node_rotation),
          transform::Rigid3d::Rotation(global_submap_rotation), true);
}

void ConstraintBuilder3D::Enqueue(const SubmapId& submap_id, const Submap3D* const submap,
                                  const NodeId& node_id,
                                  const TrajectoryNode::Data* const constant_data,
                                  const transform::Rigid3d& node_pose,
                                  const transform::Rigid3d& submap_pose,
                                  const bool match_full_submap) {
  absl::MutexLock locker(&mutex_);
  if (when_done_) std::fprintf(stderr, "MaybeAdd*Constraint was called while WhenDone was scheduled.\n");
  constraints_.emplace_back();
  pending_.push_back(Pair{&constraints_.back(), submap_id, node_id, constant_data, node_pose,
                          submap_pose, match_full_submap, MatcherOf(submap_id, submap)});
}

// CB3:172-202: one matcher per submap id, built once, on the thread pool.  The device keeps the
// precomputation stack AND both raw grids (the refinement reads them).
std::shared_ptr<ConstraintBuilder3D::DeviceMatcher> ConstraintBuilder3D::MatcherOf(
    const SubmapId& submap_id, const Submap3D* const submap) {
  auto it = matchers_.find(submap_id);
  if (it != matchers_.end()) return it->second;
  auto matcher = std::make_shared<DeviceMatcher>();
  const auto& o = options_.fast_correlative_scan_matcher_options_3d();
  const cmx_fast3d_options fast{o.branch_and_bound_depth(), o.full_resolution_depth(),
                                o.min_rotational_score(), o.min_low_resolution_score(),
                                o.linear_xy_search_window(), o.linear_z_search_window(),
                                o.angular_search_window()};
  const HybridGrid* const high = &submap->high_resolution_hybrid_grid();
  const HybridGrid* const low = &submap->low_resolution_hybrid_grid();
  const Eigen::VectorXf* const histogram = &submap->rotational_scan_matcher_histogram();
  // Placement: submap k of this builder lives in the HBM of device k mod world (mutex_ held).
  const int world = cmx_comm_num_devices(comm_);
  const int device = cmx_comm_device_of(comm_, num_matchers_created_++ % world, world);
  auto task = std::make_unique<common::Task>();
  task->SetWorkItem([matcher, high, low, histogram, fast, device] {
    const std::vector<cmx_voxel> voxels = Flatten(*high), low_voxels = Flatten(*low);
    const std::vector<float> h = Flatten(*histogram);
    CheckOk(cmx_fast3d_create(&fast, high->resolution(), high->grid_size(), voxels.data(),
                              static_cast<int64_t>(voxels.size()), low->resolution(),
                              low_voxels.data(), static_cast<int64_t>(low_voxels.size()), h.data(),
                              static_cast<int32_t>(h.size()), device, &matcher->handle),
            "cmx_fast3d_create");
  });
  matcher->creation_task = thread_pool_->Schedule(std::move(task));
  matchers_[submap_id] = matcher;
  return matcher;
}

void ConstraintBuilder3D::NotifyEndOfNode() {
  absl::MutexLock locker(&mutex_);
  auto pairs = std::make_shared<std::vector<Pair>>(std::move(pending_));
  pending_.clear();
  auto task = std::make_unique<common::Task>();
  task->SetWorkItem([this, pairs] {
    if (!pairs->empty()) ComputeNode(*pairs);
    absl::MutexLock finished(&mutex_);
    ++num_finished_nodes_;
  });
  std::set<const DeviceMatcher*> seen;              // (a Task accepts a dependency only once)
  for (const Pair& p : *pairs)
    if (seen.insert(p.matcher.get()).second) task->AddDependency(p.matcher->creation_task);
  when_done_task_->AddDependency(thread_pool_->Schedule(std::move(task)));
  ++num_started_nodes_;
}

// ComputeConstraint (CB3:204-281) for all pairs of the node.
void ConstraintBuilder3D::ComputeNode(const std::vector<Pair>& pairs) {
  std::map<const TrajectoryNode::Data*, std::vector<const Pair*>> by_node;
  for (const Pair& p : pairs) by_node[p.constant_data].push_back(&p);
  const auto& co = options_.ceres_scan_matcher_options_3d();
  Require(co.occupied_space_weight_size() == 2, "two (cloud, grid) pairs in the refinement");
  cmx_ceres3d_options ceres{};
  ceres.num_pairs = 2;
  ceres.occupied_space_weight[0] = co.occupied_space_weight(0);
  ceres.occupied_space_weight[1] = co.occupied_space_weight(1);
  ceres.translation_weight = co.translation_weight();
  ceres.rotation_weight = co.rotation_weight();
  ceres.only_optimize_yaw = co.only_optimize_yaw() ? 1 : 0;
  ceres.use_nonmonotonic_steps = co.ceres_solver_options().use_nonmonotonic_steps() ? 1 : 0;
  ceres.max_num_iterations = co.ceres_solver_options().max_num_iterations();
  for (const auto& entry : by_node) {
    const TrajectoryNode::Data& d = *entry.first;
    const std::vector<const Pair*>& group = entry.second;
    const int num = static_cast<int>(group.size());
    const std::vector<float> high = Flatten(d.high_resolution_point_cloud),
                             low = Flatten(d.low_resolution_point_cloud),
                             histogram = Flatten(d.rotational_scan_matcher_histogram);
    cmx_node_data3d data{};
    data.gravity_alignment[0] = d.gravity_alignment.w();
    data.gravity_alignment[1] = d.gravity_alignment.x();
    data.gravity_alignment[2] = d.gravity_alignment.y();
    data.gravity_alignment[3] = d.gravity_alignment.z();
    data.high_resolution_point_cloud = high.data();
    data.num_high_resolution_points = static_cast<int32_t>(high.size() / 3);
    data.low_resolution_point_cloud = low.data();
    data.num_low_resolution_points = static_cast<int32_t>(low.size() / 3);
    data.rotational_scan_matcher_histogram = histogram.data();
    data.histogram_size = static_cast<int32_t>(histogram.size());
    std::vector<const cmx_fast3d*> handles(num);
    std::vector<cmx_pose3d> node_poses(num), submap_poses(num), searched(num), refined(num);
    std::vector<int32_t> full(num), found(num);
    std::vector<float> min_scores(num);
    std::vector<cmx_result3d> results(num);
    for (int i = 0; i < num; ++i) {
      handles[i] = group[i]->matcher->handle;
      node_poses[i] = PoseOf(group[i]->node_pose);
      submap_poses[i] = PoseOf(group[i]->submap_pose);
      full[i] = group[i]->match_full_submap ? 1 : 0;
      min_scores[i] = static_cast<float>(group[i]->match_full_submap
                                             ? options_.global_localization_min_score()
                                             : options_.min_score());
    }
    // 1. + 2.: Match / MatchFullSubmap of every pair, pruned by its threshold (CB3:224-256).
    // (every device of the communicator searches the submaps it holds, concurrently)
    CheckOk(cmx_fast3d_match_sharded(comm_, handles.data(), num, node_poses.data(),
                                     submap_poses.data(), full.data(), min_scores.data(), &data,
                                     found.data(), results.data(), nullptr, nullptr, nullptr),
            "cmx_fast3d_match_sharded");
    for (int i = 0; i < num; ++i) searched[i] = results[i].pose_estimate;
    // 3.: the refinement from the found pose, which is also its target (CB3:263-276).
    CheckOk(cmx_fast3d_refine_batch(&ceres, handles.data(), num, found.data(), searched.data(),
                                    &data, refined.data(), nullptr),
            "cmx_fast3d_refine_batch");
    for (int i = 0; i < num; ++i) {
      if (!found[i]) continue;                      // `return;` at CB3:239 / :255
      const Pair& p = *group[i];
      p.slot->reset(new Constraint{p.submap_id, p.node_id,
                                   {PoseFrom(refined[i]), options_.loop_closure_translation_weight(),
                                    options_.loop_closure_rotation_weight()},
                                   Constraint::INTER_SUBMAP});
      absl::MutexLock locker(&mutex_);
      scores_.push_back(results[i].score);
      rotational_scores_.push_back(results[i].rotational_score);
      low_resolution_scores_.push_back(results[i].low_resolution_score);
    }
  }
}

void ConstraintBuilder3D::WhenDone(const std::function<void(const Result&)>& callback) {
  absl::MutexLock locker(&mutex_);
  Require(when_done_ == nullptr, "WhenDone() called twice");
  when_done_ = std::make_unique<std::function<void(const Result&)>>(callback);
  when_done_task_->SetWorkItem([this] { RunWhenDoneCallback(); });
  thread_pool_->Schedule(std::move(when_done_task_));
  when_done_task_ = std::make_unique<common::Task>();
}

void ConstraintBuilder3D::RunWhenDoneCallback() {
  Result result;
  std::unique_ptr<std::function<void(const Result&)>> callback;
  {
    absl::MutexLock locker(&mutex_);
    Require(when_done_ != nullptr, "no WhenDone callback");
    for (const std::unique_ptr<Constraint>& constraint : constraints_)
      if (constraint != nullptr) result.push_back(*constraint);
    constraints_.clear();
    callback = std::move(when_done_);
    when_done_.reset();
  }
  (*callback)(result);
}

int ConstraintBuilder3D::GetNumFinishedNodes() {
  absl::MutexLock locker(&mutex_);
  return num_finished_nodes_;
}

void ConstraintBuilder3D::DeleteScanMatcher(const SubmapId& submap_id) {
  absl::MutexLock locker(&mutex_);
  if (when_done_) std::fprintf(stderr, "DeleteScanMatcher was called while WhenDone was scheduled.\n");
  matchers_.erase(submap_id);
  per_submap_sampler_.erase(submap_id);
}

}  // namespace constraints
}  // namespace mapping
}  // namespace cartographer
