// ConstraintBuilder3D for the MI355X: the public interface of the reference's class
// (mapping/internal/constraints/constraint_builder_3d.h:50-106), a node's pairs as ONE
// cmx_fast3d_match_sharded (every search in one chain of launches per GPU of the builder's cmx_comm:
// CMX_DEVICES lists them, submap k lives on device k mod world) followed by ONE
// cmx_fast3d_refine_batch (CeresScanMatcher3D::Match for every found pair against the grids
// the matcher keeps in HBM).  See batched_constraint_builder_2d.h for the structure.
#ifndef DROPIN_BATCHED_CONSTRAINT_BUILDER_3D_H_
#define DROPIN_BATCHED_CONSTRAINT_BUILDER_3D_H_

#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <vector>

#include "Eigen/Core"
#include "Eigen/Geometry"
#include "absl/synchronization/mutex.h"
#include "cartographer/common/fixed_ratio_sampler.h"
#include "cartographer/common/task.h"
#include "cartographer/common/thread_pool.h"
#include "cartographer/mapping/3d/submap_3d.h"
#include "cartographer/mapping/id.h"
#include "cartographer/mapping/pose_graph_interface.h"
#include "cartographer/mapping/proto/pose_graph/constraint_builder_options.pb.h"
#include "cartographer/mapping/trajectory_node.h"
#include "cartographer/metrics/family_factory.h"
#include "cartographer/transform/rigid_transform.h"
#include "cartographer_mi355x.h"

namespace cartographer {
namespace mapping {
namespace constraints {

class ConstraintBuilder3D {
 public:
  using Constraint = PoseGraphInterface::Constraint;
  using Result = std::vector<Constraint>;

  ConstraintBuilder3D(const proto::ConstraintBuilderOptions& options,
                      common::ThreadPoolInterface* thread_pool);
  ~ConstraintBuilder3D();
  ConstraintBuilder3D(const ConstraintBuilder3D&) = delete;
  ConstraintBuilder3D& operator=(const ConstraintBuilder3D&) = delete;

  void MaybeAddConstraint(const SubmapId& submap_id, const Submap3D* submap, const NodeId& node_id,
                          const TrajectoryNode::Data* constant_data,
                          const transform::Rigid3d& global_node_pose,
                          const transform::Rigid3d& global_submap_pose);
  void MaybeAddGlobalConstraint(const SubmapId& submap_id, const Submap3D* submap,
                                const NodeId& node_id, const TrajectoryNode::Data* constant_data,
                                const Eigen::Quaterniond& global_node_rotation,
                                const Eigen::Quaterniond& global_submap_rotation);
  void NotifyEndOfNode();
  void WhenDone(const std::function<void(const Result&)>& callback);
  int GetNumFinishedNodes();
  void DeleteScanMatcher(const SubmapId& submap_id);
  static void RegisterMetrics(metrics::FamilyFactory*) {}

 private:
  struct DeviceMatcher {                    // stack + both raw grids of a submap in HBM
    cmx_fast3d* handle = nullptr;
    std::weak_ptr<common::Task> creation_task;
    ~DeviceMatcher() { cmx_fast3d_destroy(handle); }
  };
  struct Pair {
    std::unique_ptr<Constraint>* slot;
    SubmapId submap_id;
    NodeId node_id;
    const TrajectoryNode::Data* constant_data;
    transform::Rigid3d node_pose, submap_pose;     // rotations only for full-submap pairs
    bool match_full_submap;
    std::shared_ptr<DeviceMatcher> matcher;
  };

  std::shared_ptr<DeviceMatcher> MatcherOf(const SubmapId& submap_id, const Submap3D* submap);
  void Enqueue(const SubmapId& submap_id, const Submap3D* submap, const NodeId& node_id,
               const TrajectoryNode::Data* constant_data, const transform::Rigid3d& node_pose,
               const transform::Rigid3d& submap_pose, bool match_full_submap);
  void ComputeNode(const std::vector<Pair>& pairs);
  void RunWhenDoneCallback();

  const proto::ConstraintBuilderOptions options_;
  // The node's GPUs (constraint_builder_3d.cc schedules a node's pairs as independent thread-pool
  // tasks; here they are one sharded device call over this communicator): submap k's matcher is
  // created on device k mod world, NotifyEndOfNode's task issues ONE cmx_fast3d_match_sharded.
  cmx_comm* comm_ = nullptr;
  int num_matchers_created_ = 0;
  common::ThreadPoolInterface* const thread_pool_;
  absl::Mutex mutex_;
  std::unique_ptr<std::function<void(const Result&)>> when_done_;
  int num_started_nodes_ = 0;
  int num_finished_nodes_ = 0;
  std::unique_ptr<common::Task> when_done_task_;
  std::deque<std::unique_ptr<Constraint>> constraints_;
  std::vector<Pair> pending_;
  std::map<SubmapId, std::shared_ptr<DeviceMatcher>> matchers_;
  std::map<SubmapId, common::FixedRatioSampler> per_submap_sampler_;
  std::vector<float> scores_, rotational_scores_, low_resolution_scores_;   // histogram inputs
};

}  // namespace constraints
}  // namespace mapping
}  // namespace cartographer

#endif  // DROPIN_BATCHED_CONSTRAINT_BUILDER_3D_H_
