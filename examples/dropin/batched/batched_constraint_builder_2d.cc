// See batched_constraint_builder_2d.h.  Reference behaviour cited as CB = mapping/internal/
// constraints/constraint_builder_2d.cc.
#include "batched_constraint_builder_2d.h"

#include <cstdio>
#include <cstdlib>
#include <set>
#include <string>

#include "cartographer/mapping/2d/grid_2d.h"
#include "cartographer/transform/transform.h"

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

// Grid2D keeps its raw uint16 cells behind a protected accessor (mapping/2d/grid_2d.h:96).
struct CellAccess : Grid2D {
  using Grid2D::correspondence_cost_cells;
};
const uint16_t* CellsOf(const Grid2D& grid) {
  return (grid.*(&CellAccess::correspondence_cost_cells))().data();
}
cmx_grid2d_limits LimitsOf(const Grid2D& grid) {
  const MapLimits& l = grid.limits();
  return cmx_grid2d_limits{l.resolution(), l.max().x(), l.max().y(),
                           l.cell_limits().num_x_cells, l.cell_limits().num_y_cells,
                           grid.GetMinCorrespondenceCost(), grid.GetMaxCorrespondenceCost()};
}
cmx_pose2d PoseOf(const transform::Rigid2d& t) {
  return cmx_pose2d{t.translation().x(), t.translation().y(), t.rotation().angle()};
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

transform::Rigid2d ComputeSubmapPose(const Submap2D& submap) {
  return transform::Project2D(submap.local_pose());
}

ConstraintBuilder2D::ConstraintBuilder2D(const proto::ConstraintBuilderOptions& options,
                                         common::ThreadPoolInterface* const thread_pool)
    : options_(options), thread_pool_(thread_pool),
      when_done_task_(std::make_unique<common::Task>()) {
  const std::vector<int32_t> devices = Devices();
  CheckOk(cmx_comm_init(devices.data(), static_cast<int32_t>(devices.size()), &comm_),
          "cmx_comm_init");
}

ConstraintBuilder2D::~ConstraintBuilder2D() {
  absl::MutexLock locker(&mutex_);
  Require(when_done_task_->GetState() == common::Task::NEW, "a WhenDone task is in flight");
  Require(pending_.empty(), "NotifyEndOfNode() was not called for the last node");
  Require(constraints_.empty(), "WhenDone() was not called");
  Require(num_started_nodes_ == num_finished_nodes_, "nodes still being computed");
  Require(when_done_ == nullptr, "WhenDone callback pending");
  matchers_.clear();
  cmx_comm_destroy(comm_);
}

// CB:77-111.  The distance filter and the sampler run at the call, the search later.
void ConstraintBuilder2D::MaybeAddConstraint(const SubmapId& submap_id, const Submap2D* const submap,
                                             const NodeId& node_id,
                                             const TrajectoryNode::Data* const constant_data,
                                             const transform::Rigid2d& initial_relative_pose) {
  if (initial_relative_pose.translation().norm() > options_.max_constraint_distance()) return;
  if (!per_submap_sampler_
           .emplace(std::piecewise_construct, std::forward_as_tuple(submap_id),
                    std::forward_as_tuple(options_.sampling_ratio()))
           .first->second.Pulse()) {
    return;
  }
  Enqueue(submap_id, submap, node_id, constant_data,
          ComputeSubmapPose(*submap) * initial_relative_pose, /*match_full_submap=*/false);
}

// CB:113-137.
void ConstraintBuilder2D::MaybeAddGlobalConstraint(const SubmapId& submap_id,
                                                   const Submap2D* const submap,
                                                   const NodeId& node_id,
                                                   const TrajectoryNode::Data* const constant_data) {
  Enqueue(submap_id, submap, node_id, constant_data, transform::Rigid2d::Identity(),
          /*match_full_submap=*/true);
}

void ConstraintBuilder2D::Enqueue(const SubmapId& submap_id, const Submap2D* const submap,
                                  const NodeId& node_id,
                                  const TrajectoryNode::Data* const constant_data,
                                  const transform::Rigid2d& initial_pose,
                                  const bool match_full_submap) {
  absl::MutexLock locker(&mutex_);
  if (when_done_) std::fprintf(stderr, "MaybeAdd*Constraint was called while WhenDone was scheduled.\n");
  constraints_.emplace_back();                      // one slot per pair, filled if found
  Require(submap->grid() != nullptr, "submap without a grid");
  pending_.push_back(Pair{&constraints_.back(), submap_id, node_id, submap, constant_data,
                          initial_pose, match_full_submap, MatcherOf(submap_id, submap->grid())});
}

// CB:165-186: one matcher per submap id, built once, on the thread pool.
std::shared_ptr<ConstraintBuilder2D::DeviceMatcher> ConstraintBuilder2D::MatcherOf(
    const SubmapId& submap_id, const Grid2D* const grid) {
  auto it = matchers_.find(submap_id);
  if (it != matchers_.end()) return it->second;
  auto matcher = std::make_shared<DeviceMatcher>();
  const auto& o = options_.fast_correlative_scan_matcher_options();
  const cmx_fast2d_options fast{o.linear_search_window(), o.angular_search_window(),
                                o.branch_and_bound_depth()};
  // Placement: submap k of this builder lives in the HBM of device k mod world (mutex_ held).
  const int world = cmx_comm_num_devices(comm_);
  const int device = cmx_comm_device_of(comm_, num_matchers_created_++ % world, world);
  auto task = std::make_unique<common::Task>();
  task->SetWorkItem([matcher, grid, fast, device] {   // (the task keeps the matcher alive)
    const cmx_grid2d_limits limits = LimitsOf(*grid);
    CheckOk(cmx_fast2d_create(&fast, &limits, CellsOf(*grid), device, &matcher->handle),
            "cmx_fast2d_create");
  });
  matcher->creation_task = thread_pool_->Schedule(std::move(task));
  matchers_[submap_id] = matcher;
  return matcher;
}

// One task per node: every pair added since the last call, as one device batch.
void ConstraintBuilder2D::NotifyEndOfNode() {
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

// ComputeConstraint (CB:188-262) for all pairs of the node.  Pairs sharing a point cloud (all
// of them, in cartographer: a node is one scan) go to the device together.
void ConstraintBuilder2D::ComputeNode(const std::vector<Pair>& pairs) {
  std::map<const TrajectoryNode::Data*, std::vector<const Pair*>> by_cloud;
  for (const Pair& p : pairs) by_cloud[p.constant_data].push_back(&p);
  const auto& co = options_.ceres_scan_matcher_options();
  const cmx_ceres2d_options ceres{co.occupied_space_weight(), co.translation_weight(),
                                  co.rotation_weight(),
                                  co.ceres_solver_options().use_nonmonotonic_steps() ? 1 : 0,
                                  co.ceres_solver_options().max_num_iterations()};
  for (const auto& entry : by_cloud) {
    const std::vector<const Pair*>& group = entry.second;
    const int num = static_cast<int>(group.size());
    std::vector<float> xyz;
    for (const sensor::RangefinderPoint& point : entry.first->filtered_gravity_aligned_point_cloud) {
      xyz.push_back(point.position.x());
      xyz.push_back(point.position.y());
      xyz.push_back(point.position.z());
    }
    const int num_points = static_cast<int>(xyz.size() / 3);
    std::vector<const cmx_fast2d*> handles(num);
    std::vector<cmx_pose2d> initial(num), searched(num), refined(num);
    std::vector<int32_t> full(num), found(num);
    std::vector<float> min_scores(num), scores(num);
    for (int i = 0; i < num; ++i) {
      handles[i] = group[i]->matcher->handle;
      initial[i] = PoseOf(group[i]->initial_pose);
      full[i] = group[i]->match_full_submap ? 1 : 0;
      min_scores[i] = static_cast<float>(group[i]->match_full_submap
                                             ? options_.global_localization_min_score()
                                             : options_.min_score());
    }
    // 1. + 2.: the correlative searches, pruned by their thresholds (CB:211-236): every device
    // of the communicator searches the submaps it holds, concurrently.
    CheckOk(cmx_fast2d_match_sharded(comm_, handles.data(), num, initial.data(), full.data(),
                                     min_scores.data(), xyz.data(), num_points, found.data(),
                                     scores.data(), searched.data(), nullptr, nullptr, nullptr),
            "cmx_fast2d_match_sharded");
    // 3.: refinement from the found pose, which is also its target (CB:242-249).
    CheckOk(cmx_fast2d_refine_batch(&ceres, handles.data(), num, found.data(), searched.data(),
                                    xyz.data(), num_points, refined.data(), nullptr),
            "cmx_fast2d_refine_batch");
    for (int i = 0; i < num; ++i) {
      if (!found[i]) continue;                      // `return;` at CB:219 / :232
      const Pair& p = *group[i];
      const transform::Rigid2d pose_estimate({refined[i].x, refined[i].y}, refined[i].theta);
      const transform::Rigid2d constraint_transform =
          ComputeSubmapPose(*p.submap).inverse() * pose_estimate;      // CB:251-252
      p.slot->reset(new Constraint{p.submap_id, p.node_id,
                                   {transform::Embed3D(constraint_transform),
                                    options_.loop_closure_translation_weight(),
                                    options_.loop_closure_rotation_weight()},
                                   Constraint::INTER_SUBMAP});
      absl::MutexLock locker(&mutex_);
      scores_.push_back(scores[i]);
    }
  }
}

// CB:152-163.
void ConstraintBuilder2D::WhenDone(const std::function<void(const Result&)>& callback) {
  absl::MutexLock locker(&mutex_);
  Require(when_done_ == nullptr, "WhenDone() called twice");
  when_done_ = std::make_unique<std::function<void(const Result&)>>(callback);
  when_done_task_->SetWorkItem([this] { RunWhenDoneCallback(); });
  thread_pool_->Schedule(std::move(when_done_task_));
  when_done_task_ = std::make_unique<common::Task>();
}

// CB:278-299: the found constraints in the order their pairs were added.
void ConstraintBuilder2D::RunWhenDoneCallback() {
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

int ConstraintBuilder2D::GetNumFinishedNodes() {
  absl::MutexLock locker(&mutex_);
  return num_finished_nodes_;
}

// CB:306-316.
void ConstraintBuilder2D::DeleteScanMatcher(const SubmapId& submap_id) {
  absl::MutexLock locker(&mutex_);
  if (when_done_) std::fprintf(stderr, "DeleteScanMatcher was called while WhenDone was scheduled.\n");
  matchers_.erase(submap_id);
  per_submap_sampler_.erase(submap_id);
}

}  // namespace constraints
}  // namespace mapping
}  // namespace cartographer
