// ConstraintBuilder2D for the MI355X: the public interface of the reference's class
// (mapping/internal/constraints/constraint_builder_2d.h:60-109 -- PoseGraph2D compiles against
// either), a different inside.  The reference schedules one thread-pool task per (node, submap)
// pair, each running a search and a refinement on a host core.  Here a node's pairs are
// queued and NotifyEndOfNode schedules ONE task that hands all of them to the device:
// cmx_fast2d_match_sharded (every search of the node in one chain of launches per GPU of the
// builder's cmx_comm: CMX_DEVICES lists them, submap k lives on device k mod world), then
// cmx_fast2d_refine_batch (every found pair's Ceres refinement in one launch), against
// precomputation stacks and grids that stay in HBM from the first use of a submap until
// DeleteScanMatcher.  Results, their order, the sampler, the distance filter, the WhenDone /
// GetNumFinishedNodes contract are the reference's.
#ifndef DROPIN_BATCHED_CONSTRAINT_BUILDER_2D_H_
#define DROPIN_BATCHED_CONSTRAINT_BUILDER_2D_H_

#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <vector>

#include "absl/synchronization/mutex.h"
#include "cartographer/common/fixed_ratio_sampler.h"
#include "cartographer/common/task.h"
#include "cartographer/common/thread_pool.h"
#include "cartographer/mapping/2d/submap_2d.h"
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

// (map <- submap), the submap's origin as a 2D pose (reference: constraint_builder_2d.h:47-49).
transform::Rigid2d ComputeSubmapPose(const Submap2D& submap);

class ConstraintBuilder2D {
 public:
  using Constraint = PoseGraphInterface::Constraint;
  using Result = std::vector<Constraint>;

  ConstraintBuilder2D(const proto::ConstraintBuilderOptions& options,
                      common::ThreadPoolInterface* thread_pool);
  ~ConstraintBuilder2D();
  ConstraintBuilder2D(const ConstraintBuilder2D&) = delete;
  ConstraintBuilder2D& operator=(const ConstraintBuilder2D&) = delete;

  void MaybeAddConstraint(const SubmapId& submap_id, const Submap2D* submap, const NodeId& node_id,
                          const TrajectoryNode::Data* constant_data,
                          const transform::Rigid2d& initial_relative_pose);
  void MaybeAddGlobalConstraint(const SubmapId& submap_id, const Submap2D* submap,
                                const NodeId& node_id, const TrajectoryNode::Data* constant_data);
  void NotifyEndOfNode();
  void WhenDone(const std::function<void(const Result&)>& callback);
  int GetNumFinishedNodes();
  void DeleteScanMatcher(const SubmapId& submap_id);
  static void RegisterMetrics(metrics::FamilyFactory*) {}     // no metric families kept here

 private:
  // A submap's stack in HBM.  Shared with the queued pairs: DeleteScanMatcher only drops the
  // builder's reference, a node already queued keeps its matchers alive.
  struct DeviceMatcher {
    cmx_fast2d* handle = nullptr;
    std::weak_ptr<common::Task> creation_task;
    ~DeviceMatcher() { cmx_fast2d_destroy(handle); }
  };
  struct Pair {
    std::unique_ptr<Constraint>* slot;      // into constraints_ (a deque: stable addresses)
    SubmapId submap_id;
    NodeId node_id;
    const Submap2D* submap;
    const TrajectoryNode::Data* constant_data;
    transform::Rigid2d initial_pose;        // ComputeSubmapPose(submap) * initial_relative_pose
    bool match_full_submap;
    std::shared_ptr<DeviceMatcher> matcher;
  };

  std::shared_ptr<DeviceMatcher> MatcherOf(const SubmapId& submap_id, const Grid2D* grid);
  void Enqueue(const SubmapId& submap_id, const Submap2D* submap, const NodeId& node_id,
               const TrajectoryNode::Data* constant_data, const transform::Rigid2d& initial_pose,
               bool match_full_submap);
  void ComputeNode(const std::vector<Pair>& pairs);
  void RunWhenDoneCallback();

  const proto::ConstraintBuilderOptions options_;
  // The node's GPUs (constraint_builder_2d.cc schedules a node's pairs as independent thread-pool
  // tasks; here they are one sharded device call over this communicator): submap k's matcher is
  // created on device k mod world, NotifyEndOfNode's task issues ONE cmx_fast2d_match_sharded.
  cmx_comm* comm_ = nullptr;
  int num_matchers_created_ = 0;
  common::ThreadPoolInterface* const thread_pool_;
  absl::Mutex mutex_;
  std::unique_ptr<std::function<void(const Result&)>> when_done_;
  int num_started_nodes_ = 0;
  int num_finished_nodes_ = 0;
  std::unique_ptr<common::Task> when_done_task_;
  std::deque<std::unique_ptr<Constraint>> constraints_;
  std::vector<Pair> pending_;               // the pairs added since the last NotifyEndOfNode
  std::map<SubmapId, std::shared_ptr<DeviceMatcher>> matchers_;
  std::map<SubmapId, common::FixedRatioSampler> per_submap_sampler_;
  std::vector<float> scores_;               // of every constraint found (the score histogram's input)
};

}  // namespace constraints
}  // namespace mapping
}  // namespace cartographer

#endif  // DROPIN_BATCHED_CONSTRAINT_BUILDER_2D_H_
