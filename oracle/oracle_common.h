// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// CPU restatement of the arithmetic of cartographer's correlative scan
// matching hot path.  Nothing under oracle/ is part of the product: only
// tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may load
// it, and only as the checker / reported baseline.
//
// Parity status: the real reference cannot be compiled as a whole in this image
// (Eigen, glog, Abseil, protobuf, Ceres are absent), so the restatement is pinned
// against every known-answer test the reference holds for the path
// (tests/test_oracle_reference_pins*.py list them with file:line), and -- for the
// value / odds / conversion tables below, the inserter's ray mask, the whole
// 2D matcher algorithm (oracle_2d.cc) and the whole 3D matcher algorithm
// (oracle_3d.cc, over the reference's real hybrid_grid.h) -- against the
// reference's own translation units compiled in place (oracle/_ref,
// tests/test_reference_ref.py and tests/test_reference_ref_3d.py:
// bit-identical).  Bit-level Eigen parity (the quaternion / affine kernels, which
// that build stands in) is UNPINNED; see DESIGN.md.
//
// This header: constants, lookup tables, rounding and the small subset of
// Eigen geometry the path uses, restated with Eigen 3.3's operation order.
#ifndef ORACLE_COMMON_H_
#define ORACLE_COMMON_H_

#include <cmath>
#include <cstdint>
#include <vector>

namespace oracle {

// cartographer/common/port.h:40-42  (RoundToInt = std::lround).
inline int RoundToInt(const float x) { return static_cast<int>(std::lround(x)); }
inline int RoundToInt(const double x) { return static_cast<int>(std::lround(x)); }

// cartographer/mapping/probability_values.h:64-67, all evaluated in f32.
constexpr float kMinProbability = 0.1f;
constexpr float kMaxProbability = 1.f - kMinProbability;
constexpr float kMinCorrespondenceCost = 1.f - kMaxProbability;
constexpr float kMaxCorrespondenceCost = 1.f - kMinProbability;
constexpr uint16_t kUnknownValue = 0;
constexpr uint16_t kUpdateMarker = 1u << 15;

inline float ClampF(float v, float lo, float hi) {  // common/math.h:31-40
  if (v > hi) return hi;
  if (v < lo) return lo;
  return v;
}

// probability_values.h:32-44.
inline uint16_t BoundedFloatToValue(float v, float lo, float hi) {
  return static_cast<uint16_t>(
      RoundToInt((ClampF(v, lo, hi) - lo) * (32766.f / (hi - lo))) + 1);
}
inline uint16_t CorrespondenceCostToValue(float c) {
  return BoundedFloatToValue(c, kMinCorrespondenceCost, kMaxCorrespondenceCost);
}
inline uint16_t ProbabilityToValue(float p) {
  return BoundedFloatToValue(p, kMinProbability, kMaxProbability);
}

// probability_values.cc:33-41 — the global 2x32768 tables.
inline float SlowValueToBoundedFloat32768(uint16_t value, float unknown_result,
                                          float lo, float hi) {
  if (value == 0) return unknown_result;
  const float scale = (hi - lo) / (32768 - 2.f);
  return value * scale + (lo - scale);
}
const std::vector<float>& ValueToProbabilityTable();          // probability_values.cc:59-63
const std::vector<float>& ValueToCorrespondenceCostTable();   // probability_values.cc:65-69
// value_conversion_tables.cc:29-51 — the per-grid 65536-entry table a
// ProbabilityGrid asks for with (unknown=kMaxCC, lo=kMinCC, hi=kMaxCC)
// (grid_2d.cc:69-71).
const std::vector<float>& GridCorrespondenceCostTable();

inline float ValueToProbability(uint16_t v) { return ValueToProbabilityTable()[v]; }

// ---- minimal Eigen 3.3 geometry (float), in Eigen's operation order ----
struct V3f { float x, y, z; };
struct Qf { float w, x, y, z; };

inline V3f Cross(const V3f& a, const V3f& b) {  // Eigen OrthoMethods.h cross()
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
// Quaternion::_transformVector:  uv = q.vec x v; uv += uv; v + w*uv + q.vec x uv.
inline V3f Rotate(const Qf& q, const V3f& v) {
  const V3f qv{q.x, q.y, q.z};
  V3f uv = Cross(qv, v);
  uv.x += uv.x; uv.y += uv.y; uv.z += uv.z;
  const V3f c = Cross(qv, uv);
  return {(v.x + q.w * uv.x) + c.x, (v.y + q.w * uv.y) + c.y, (v.z + q.w * uv.z) + c.z};
}
// Quaternion(AngleAxisf(angle, UnitZ)):  w = cos(a/2), vec = sin(a/2) * axis.
inline Qf QuatFromYaw(float angle) {
  const float ha = 0.5f * angle;
  const float s = std::sin(ha);
  return {std::cos(ha), s * 0.f, s * 0.f, s * 1.f};
}
// Generic (non-SSE) quaternion product of Eigen 3.3 (Quaternion.h quat_product).
inline Qf QuatMul(const Qf& a, const Qf& b) {
  return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z,
          a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
          a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
          a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x};
}
inline Qf QuatNormalized(const Qf& q) {  // coeffs / norm()
  const float n = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  return {q.w / n, q.x / n, q.y / n, q.z / n};
}
inline Qf QuatConj(const Qf& q) { return {q.w, -q.x, -q.y, -q.z}; }

struct Rigid3f {
  V3f t{0, 0, 0};
  Qf q{1, 0, 0, 0};
};
// transform/rigid_transform.h:183-189 (product renormalises the rotation).
inline Rigid3f Mul(const Rigid3f& a, const Rigid3f& b) {
  const V3f r = Rotate(a.q, b.t);
  return {{r.x + a.t.x, r.y + a.t.y, r.z + a.t.z}, QuatNormalized(QuatMul(a.q, b.q))};
}
inline V3f Apply(const Rigid3f& a, const V3f& p) {  // rigid_transform.h:191-196
  const V3f r = Rotate(a.q, p);
  return {r.x + a.t.x, r.y + a.t.y, r.z + a.t.z};
}
inline Rigid3f Inverse(const Rigid3f& a) {  // rigid_transform.h:151-155
  const Qf c = QuatConj(a.q);
  const V3f r = Rotate(c, a.t);
  return {{-r.x, -r.y, -r.z}, c};
}

}  // namespace oracle

#endif  // ORACLE_COMMON_H_
