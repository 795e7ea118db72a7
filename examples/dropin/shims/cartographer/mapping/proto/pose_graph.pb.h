// Stand-in for the generated messages mapping/id.h converts to (never called here).
#ifndef DROPIN_SHIMS_POSE_GRAPH_PB_H_
#define DROPIN_SHIMS_POSE_GRAPH_PB_H_
namespace cartographer { namespace mapping { namespace proto {
struct NodeId {
  int trajectory_id_ = 0, node_index_ = 0;
  void set_trajectory_id(int v) { trajectory_id_ = v; }
  void set_node_index(int v) { node_index_ = v; }
};
struct SubmapId {
  int trajectory_id_ = 0, submap_index_ = 0;
  void set_trajectory_id(int v) { trajectory_id_ = v; }
  void set_submap_index(int v) { submap_index_ = v; }
};
} } }
#endif  // DROPIN_SHIMS_POSE_GRAPH_PB_H_
