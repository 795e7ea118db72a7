// Stand-in for mapping/pose_graph_interface.h: the Constraint struct (pose_graph_interface.h:
// 36-53) ConstraintBuilder2D produces.
#ifndef DROPIN_SHIMS_POSE_GRAPH_INTERFACE_H_
#define DROPIN_SHIMS_POSE_GRAPH_INTERFACE_H_
#include "cartographer/mapping/id.h"
#include "cartographer/transform/rigid_transform.h"
namespace cartographer { namespace mapping {
class PoseGraphInterface {
 public:
  struct Constraint {
    struct Pose {
      transform::Rigid3d zbar_ij;
      double translation_weight;
      double rotation_weight;
    };
    SubmapId submap_id;
    NodeId node_id;
    Pose pose;
    enum Tag { INTRA_SUBMAP, INTER_SUBMAP } tag;
  };
};
} }
#endif  // DROPIN_SHIMS_POSE_GRAPH_INTERFACE_H_
