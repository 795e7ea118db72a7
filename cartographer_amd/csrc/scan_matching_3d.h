// Internal declarations shared by the 3D matchers: dense device bricks that
// stand in for HybridGrid / PrecomputationGrid3D on the read side, and the
// small amount of Eigen-ordered f32 geometry the host needs.
//
// Data layout.  The reference's HybridGrid is a pointer tree (DynamicGrid ->
// NestedGrid -> FlatGrid, mapping/3d/hybrid_grid.h:143-407) that costs three
// dependent loads per lookup.  A caller flattens it once through its Iterator
// (hybrid_grid.h:304-372) into a voxel list; on the device it becomes a dense
// brick over the bounding box of the non-zero voxels, x fastest.  Reads outside
// the brick return 0 — exactly what the tree returns for cells never written
// (hybrid_grid.h:263-279).
#ifndef CMX_SCAN_MATCHING_3D_H_
#define CMX_SCAN_MATCHING_3D_H_

#include <cmath>
#include <memory>
#include <vector>

#include "cmx_common.h"
#include "cmx_device.h"

namespace cmx {

struct Brick {             // device-visible
  const void* cells;       // uint16_t (grid) or uint8_t (precomputation level)
  int lo_x, lo_y, lo_z;    // cell index of cells[0]
  int nx, ny, nz;
};

__device__ __forceinline__ unsigned BrickValueU8(const Brick& b, int x, int y, int z) {
  const int ix = x - b.lo_x, iy = y - b.lo_y, iz = z - b.lo_z;
  const bool inside = static_cast<unsigned>(ix) < static_cast<unsigned>(b.nx) &&
                      static_cast<unsigned>(iy) < static_cast<unsigned>(b.ny) &&
                      static_cast<unsigned>(iz) < static_cast<unsigned>(b.nz);
  // Unconditional load from a clamped offset, masked afterwards (loads inside
  // an `if` cannot overlap).
  const size_t off = inside ? (static_cast<size_t>(iz) * b.ny + iy) * b.nx + ix : 0;
  const unsigned v = static_cast<const uint8_t*>(b.cells)[off];
  return inside ? v : 0u;
}
__device__ __forceinline__ unsigned BrickValueU16(const Brick& b, int x, int y, int z) {
  const int ix = x - b.lo_x, iy = y - b.lo_y, iz = z - b.lo_z;
  const bool inside = static_cast<unsigned>(ix) < static_cast<unsigned>(b.nx) &&
                      static_cast<unsigned>(iy) < static_cast<unsigned>(b.ny) &&
                      static_cast<unsigned>(iz) < static_cast<unsigned>(b.nz);
  const size_t off = inside ? (static_cast<size_t>(iz) * b.ny + iy) * b.nx + ix : 0;
  const unsigned v = static_cast<const uint16_t*>(b.cells)[off];
  return inside ? v : 0u;
}

// kValueToProbability (mapping/probability_values.cc:33-41,59-63) evaluated
// arithmetically: 0 -> kMinProbability, else v*scale + (lo - scale).
__device__ __forceinline__ float ValueToProbabilityDev(unsigned raw) {
  const float kMinP = 0.1f;
  const float kMaxP = 1.f - kMinP;
  const unsigned v = raw & 32767u;
  if (v == 0) return kMinP;
  const float scale = (kMaxP - kMinP) / (32768 - 2.f);
  return static_cast<float>(v) * scale + (kMinP - scale);
}

// HybridGrid::GetCellIndex (hybrid_grid.h:428-433): lround(p / resolution), f32.
__device__ __forceinline__ int3 CellIndex3(const F3& p, float resolution) {
  return make_int3(LRoundF32(p.x / resolution), LRoundF32(p.y / resolution),
                   LRoundF32(p.z / resolution));
}

// ---- host f32 geometry in Eigen 3.3's operation order -----------------------
// (float quaternion products follow the SSE kernel an x86-64 build of the
// reference uses; see DESIGN.md "Eigen parity").
namespace h3 {
struct V3 { float x, y, z; };
struct Q { float w, x, y, z; };
struct Rigid { V3 t{0, 0, 0}; Q q{1, 0, 0, 0}; };

inline V3 Cross(const V3& a, const V3& b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline V3 Rotate(const Q& q, const V3& v) {
  const V3 qv{q.x, q.y, q.z};
  V3 uv = Cross(qv, v);
  uv.x += uv.x; uv.y += uv.y; uv.z += uv.z;
  const V3 c = Cross(qv, uv);
  return {(v.x + q.w * uv.x) + c.x, (v.y + q.w * uv.y) + c.y, (v.z + q.w * uv.z) + c.z};
}
inline Q Mul(const Q& a, const Q& b) {
  Q r;
  r.x = (a.x * b.w - a.z * b.y) + (a.y * b.z + a.w * b.x);
  r.y = (a.y * b.w - a.x * b.z) + (a.z * b.x + a.w * b.y);
  r.z = (a.z * b.w - a.y * b.x) + (a.x * b.y + a.w * b.z);
  r.w = (a.w * b.w - a.x * b.x) + -(a.z * b.z + a.y * b.y);
  return r;
}
inline float SquaredNorm(const Q& q) { return (q.x * q.x + q.z * q.z) + (q.y * q.y + q.w * q.w); }
inline Q Normalized(const Q& q) {
  const float z = SquaredNorm(q);
  if (z > 0.f) {
    const float n = std::sqrt(z);
    return {q.w / n, q.x / n, q.y / n, q.z / n};
  }
  return q;
}
inline Q Inverse(const Q& q) {
  const float n2 = SquaredNorm(q);
  return {q.w / n2, -q.x / n2, -q.y / n2, -q.z / n2};
}
inline Q Conj(const Q& q) { return {q.w, -q.x, -q.y, -q.z}; }
inline Rigid Mul(const Rigid& a, const Rigid& b) {   // transform/rigid_transform.h:183-189
  const V3 r = Rotate(a.q, b.t);
  return {{r.x + a.t.x, r.y + a.t.y, r.z + a.t.z}, Normalized(Mul(a.q, b.q))};
}
inline Rigid InverseRigid(const Rigid& a) {          // rigid_transform.h:151-155
  const Q c = Conj(a.q);
  const V3 r = Rotate(c, a.t);
  return {{-r.x, -r.y, -r.z}, c};
}
inline float Norm(const V3& v) { return std::sqrt((v.x * v.x + v.y * v.y) + v.z * v.z); }
inline Q FromAngleAxisVector(const V3& aa) {         // transform/transform.h:85-99
  float scale = 0.5f, w = 1.f;
  const float squared_norm = (aa.x * aa.x + aa.y * aa.y) + aa.z * aa.z;
  if (squared_norm > 1e-8) {
    const float norm = std::sqrt(squared_norm);
    scale = static_cast<float>(std::sin(norm / 2.) / norm);
    w = static_cast<float>(std::cos(norm / 2.));
  }
  return {w, scale * aa.x, scale * aa.y, scale * aa.z};
}
inline float GetAngle(const Rigid& t) {              // transform/transform.h:33-37
  const float n = std::sqrt((t.q.x * t.q.x + t.q.y * t.q.y) + t.q.z * t.q.z);
  return 2.f * std::atan2(n, std::abs(t.q.w));
}
inline float GetYaw(const Q& q) {                    // transform/transform.h:42-47
  const V3 d = Rotate(q, V3{1.f, 0.f, 0.f});
  return std::atan2(d.y, d.x);
}
inline Rigid FromPose(const cmx_pose3d& p) {         // Rigid3d::cast<float>()
  Rigid r;
  r.t = {static_cast<float>(p.t[0]), static_cast<float>(p.t[1]), static_cast<float>(p.t[2])};
  r.q = {static_cast<float>(p.q[0]), static_cast<float>(p.q[1]), static_cast<float>(p.q[2]),
         static_cast<float>(p.q[3])};
  return r;
}
inline cmx_pose3d ToPose(const Rigid& r) {           // Rigid3f::cast<double>()
  cmx_pose3d p;
  p.t[0] = r.t.x; p.t[1] = r.t.y; p.t[2] = r.t.z;
  p.q[0] = r.q.w; p.q[1] = r.q.x; p.q[2] = r.q.y; p.q[3] = r.q.z;
  return p;
}
}  // namespace h3

// Host copy of a dense brick's geometry + device storage.
struct DeviceBrick {
  Brick desc{};
  void* mem = nullptr;
  size_t bytes = 0;
  ~DeviceBrick() { if (mem) (void)hipFree(mem); }
  DeviceBrick() = default;
  DeviceBrick(const DeviceBrick&) = delete;
  DeviceBrick& operator=(const DeviceBrick&) = delete;
};

// Bounding box of a voxel list; false when the list is empty.
bool VoxelBounds(const cmx_voxel* voxels, int64_t n, int lo[3], int hi[3]);
// Uploads `voxels` and scatters them into a zeroed dense brick of `bytes_per_cell`
// (2: raw uint16 values; 1: ConvertToPrecomputationGrid's uint8 values).
void BuildBrickFromVoxels(Workspace& ws, const cmx_voxel* voxels, int64_t n, int bytes_per_cell,
                          DeviceBrick* out);
// DynamicGrid growth rule (hybrid_grid.h:259,381-398).
int GridSizeOf(const cmx_voxel* voxels, int64_t n);
// fast_3d.hip: device and raw (uint16) grids of a 3D matcher.
int Fast3DDevice(const cmx_fast3d* matcher);
void Fast3DGrids(const cmx_fast3d* matcher, Brick* high, float* resolution, Brick* low,
                 float* low_resolution);

// grid_3d.hip: the resident HybridGrid's brick (false: still empty).
bool Grid3DBrick(const cmx_grid3d* grid, Brick* brick, float* resolution, int* device);
// grid_3d.hip: the f32 brick of a resident IntensityHybridGrid's averages (built on `stream` when
// the grid has changed); false while the grid is empty.
bool IntensityGrid3DBrick(cmx_intensity_grid3d* grid, hipStream_t stream, Brick* brick,
                          float* resolution, int* device);

}  // namespace cmx

#endif  // CMX_SCAN_MATCHING_3D_H_
