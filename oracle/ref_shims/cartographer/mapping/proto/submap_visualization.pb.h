// Stand-in for mapping::proto::SubmapQuery::Response::SubmapTexture, which the grid classes'
// DrawToSubmapTexture overrides fill (compiled, never called here).
#ifndef ORACLE_REF_SHIMS_SUBMAP_VISUALIZATION_PB_H_
#define ORACLE_REF_SHIMS_SUBMAP_VISUALIZATION_PB_H_
#include <string>
#include "cartographer/transform/proto/transform.pb.h"
namespace cartographer {
namespace mapping {
namespace proto {
class SubmapQuery {
 public:
  class Response {
   public:
    class SubmapTexture {
     public:
      std::string* mutable_cells() { return &cells_; }
      void set_width(int v) { width_ = v; }
      void set_height(int v) { height_ = v; }
      void set_resolution(double v) { resolution_ = v; }
      transform::proto::Rigid3d* mutable_slice_pose() { return &slice_pose_; }
     private:
      std::string cells_;
      int width_ = 0, height_ = 0;
      double resolution_ = 0.;
      transform::proto::Rigid3d slice_pose_;
    };
  };
};
}  // namespace proto
}  // namespace mapping
}  // namespace cartographer
#endif  // ORACLE_REF_SHIMS_SUBMAP_VISUALIZATION_PB_H_
