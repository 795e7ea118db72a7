// Synthetic 2D worlds, range scans and probability grids for the benchmark and
// the parity tests (SURVEY.md §8d "Synthetic inputs"): a rectangular room with
// random box obstacles, ray-cast lidar scans with range noise, and grids
// rendered through the range-data inserter restatement
// (probability_grid_builder.h) with the hit 0.7 / miss 0.4 odds every reference
// fixture uses (e.g. fast_correlative_scan_matcher_2d_test.cc:134-139).
//
// Host-only C entry points (libcmx_synth.so); not on the device hot path.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

#include "probability_grid_builder.h"

using cartographer_amd::host::CellIndex;
using cartographer_amd::host::ProbabilityGridBuilder;

namespace {

// splitmix64: tiny, portable, good enough for fixtures.
struct Rng {
  uint64_t s;
  explicit Rng(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 0x1234567ull) {}
  uint64_t next() {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  double uniform() { return (next() >> 11) * (1.0 / 9007199254740992.0); }
  double uniform(double lo, double hi) { return lo + (hi - lo) * uniform(); }
  double normal() {  // Box–Muller
    const double u1 = std::max(uniform(), 1e-300), u2 = uniform();
    return std::sqrt(-2.0 * std::log(u1)) * std::cos(2.0 * M_PI * u2);
  }
};

struct Segment { double x0, y0, x1, y1; };
struct Box { double cx, cy, hx, hy, c, s; };

struct World {
  double min_x, max_x, min_y, max_y;  // room interior
  std::vector<Segment> segments;
  std::vector<Box> boxes;

  bool InsideBox(double x, double y, double margin) const {
    for (const Box& b : boxes) {
      const double dx = x - b.cx, dy = y - b.cy;
      const double lx = b.c * dx + b.s * dy, ly = -b.s * dx + b.c * dy;
      if (std::abs(lx) < b.hx + margin && std::abs(ly) < b.hy + margin) return true;
    }
    return false;
  }
  bool Free(double x, double y, double margin) const {
    return x > min_x + margin && x < max_x - margin && y > min_y + margin &&
           y < max_y - margin && !InsideBox(x, y, margin);
  }
  double Raycast(double ox, double oy, double angle) const {
    const double dx = std::cos(angle), dy = std::sin(angle);
    double best = std::numeric_limits<double>::infinity();
    for (const Segment& g : segments) {
      const double ex = g.x1 - g.x0, ey = g.y1 - g.y0;
      const double det = ex * dy - ey * dx;
      if (std::abs(det) < 1e-12) continue;
      const double wx = g.x0 - ox, wy = g.y0 - oy;
      const double t = (ex * wy - ey * wx) / det;   // along the ray
      const double u = (dx * wy - dy * wx) / det;   // along the segment
      if (t > 1e-9 && u >= 0. && u <= 1. && t < best) best = t;
    }
    return best;
  }
};

void AddRect(World* w, double cx, double cy, double hx, double hy, double angle) {
  const double c = std::cos(angle), s = std::sin(angle);
  const double px[4] = {-hx, hx, hx, -hx}, py[4] = {-hy, -hy, hy, hy};
  double X[4], Y[4];
  for (int i = 0; i != 4; ++i) {
    X[i] = cx + c * px[i] - s * py[i];
    Y[i] = cy + s * px[i] + c * py[i];
  }
  for (int i = 0; i != 4; ++i) w->segments.push_back({X[i], Y[i], X[(i + 1) % 4], Y[(i + 1) % 4]});
}

}  // namespace

extern "C" {

// ----------------------------------------------------------------- world ---
void* cmx_synth_world_create(uint64_t seed, double min_x, double max_x, double min_y,
                             double max_y) {
  auto* w = new World{min_x, max_x, min_y, max_y, {}, {}};
  AddRect(w, 0.5 * (min_x + max_x), 0.5 * (min_y + max_y), 0.5 * (max_x - min_x),
          0.5 * (max_y - min_y), 0.);
  Rng rng(seed);
  const int num_boxes = 6 + static_cast<int>(rng.next() % 5);
  const double span = std::min(max_x - min_x, max_y - min_y);
  for (int i = 0; i != num_boxes; ++i) {
    const double hx = rng.uniform(0.03, 0.10) * span, hy = rng.uniform(0.03, 0.10) * span;
    const double r = std::hypot(hx, hy);
    const double cx = rng.uniform(min_x + r, max_x - r), cy = rng.uniform(min_y + r, max_y - r);
    const double a = rng.uniform(0., M_PI);
    w->boxes.push_back({cx, cy, hx, hy, std::cos(a), std::sin(a)});
    AddRect(w, cx, cy, hx, hy, a);
  }
  return w;
}
void cmx_synth_world_destroy(void* world) { delete static_cast<World*>(world); }

// A pose at least `clearance` metres from every wall / box.
void cmx_synth_world_free_pose(void* world, uint64_t seed, double clearance, double* pose_xyt) {
  const World& w = *static_cast<World*>(world);
  Rng rng(seed ^ 0xABCDEF0123ull);
  for (int tries = 0; tries != 100000; ++tries) {
    const double x = rng.uniform(w.min_x, w.max_x), y = rng.uniform(w.min_y, w.max_y);
    if (w.Free(x, y, clearance)) {
      pose_xyt[0] = x; pose_xyt[1] = y; pose_xyt[2] = rng.uniform(-M_PI, M_PI);
      return;
    }
  }
  pose_xyt[0] = 0.5 * (w.min_x + w.max_x); pose_xyt[1] = 0.5 * (w.min_y + w.max_y);
  pose_xyt[2] = 0.;
}

// `beams` bearings uniform over 2π in the sensor frame; returns the number of
// hits within `max_range`, written as float xyz (z = 0) in the SENSOR frame.
int cmx_synth_scan(void* world, const double* pose_xyt, int beams, double max_range,
                   double sigma, uint64_t seed, float* xyz_out) {
  const World& w = *static_cast<World*>(world);
  Rng rng(seed ^ 0x5CA9ull);
  int n = 0;
  for (int i = 0; i != beams; ++i) {
    const double bearing = 2.0 * M_PI * i / beams;
    double r = w.Raycast(pose_xyt[0], pose_xyt[1], pose_xyt[2] + bearing);
    const double noise = sigma * rng.normal();
    if (!(r < max_range)) continue;
    r += noise;
    xyz_out[3 * n + 0] = static_cast<float>(r * std::cos(bearing));
    xyz_out[3 * n + 1] = static_cast<float>(r * std::sin(bearing));
    xyz_out[3 * n + 2] = 0.f;
    ++n;
  }
  return n;
}

// ------------------------------------------------------- probability grid ---
void* cmx_pgrid_create(double resolution, double max_x, double max_y, int nx, int ny) {
  try {
    return new ProbabilityGridBuilder(resolution, max_x, max_y, nx, ny);
  } catch (...) {
    return nullptr;
  }
}
void cmx_pgrid_destroy(void* g) { delete static_cast<ProbabilityGridBuilder*>(g); }
void cmx_pgrid_limits(void* g, double* res, double* max_x, double* max_y, int* nx, int* ny) {
  const auto& b = *static_cast<ProbabilityGridBuilder*>(g);
  *res = b.resolution(); *max_x = b.max_x(); *max_y = b.max_y();
  *nx = b.num_x_cells(); *ny = b.num_y_cells();
}
void cmx_pgrid_cells(void* g, uint16_t* out) {
  const auto& b = *static_cast<ProbabilityGridBuilder*>(g);
  std::memcpy(out, b.cells().data(), b.cells().size() * sizeof(uint16_t));
}
int cmx_pgrid_set_probability(void* g, int ix, int iy, float probability) {
  try {
    static_cast<ProbabilityGridBuilder*>(g)->SetProbability({ix, iy}, probability);
    return 0;
  } catch (...) {
    return 1;
  }
}
float cmx_pgrid_get_probability(void* g, int ix, int iy) {
  return static_cast<ProbabilityGridBuilder*>(g)->GetProbability({ix, iy});
}
// Points are in the map frame.
int cmx_pgrid_insert(void* g, const float* origin_xy, const float* returns_xyz, int num_returns,
                     const float* misses_xyz, int num_misses, float hit_probability,
                     float miss_probability, int insert_free_space) {
  using namespace cartographer_amd::host;
  try {
    const auto hit = ComputeLookupTableToApplyCorrespondenceCostOdds(Odds(hit_probability));
    const auto miss = ComputeLookupTableToApplyCorrespondenceCostOdds(Odds(miss_probability));
    static_cast<ProbabilityGridBuilder*>(g)->Insert(origin_xy, returns_xyz, num_returns,
                                                    misses_xyz, num_misses, hit, miss,
                                                    insert_free_space != 0);
    return 0;
  } catch (...) {
    return 1;
  }
}
void* cmx_pgrid_cropped(void* g) {
  return new ProbabilityGridBuilder(static_cast<ProbabilityGridBuilder*>(g)->Cropped());
}
void cmx_pgrid_odds_table(float probability, uint16_t* out32768) {
  using namespace cartographer_amd::host;
  const auto t = ComputeLookupTableToApplyCorrespondenceCostOdds(Odds(probability));
  std::memcpy(out32768, t.data(), 32768 * sizeof(uint16_t));
}
void cmx_cells_on_ray(int bx, int by, int ex, int ey, int scale, int* out_xy, int capacity,
                      int* count) {
  std::vector<CellIndex> cells;
  cartographer_amd::host::CellsOnRay({bx, by}, {ex, ey}, scale, &cells);
  *count = static_cast<int>(cells.size());
  for (int i = 0; i < *count && i < capacity; ++i) {
    out_xy[2 * i] = cells[i].x; out_xy[2 * i + 1] = cells[i].y;
  }
}

// One synthetic submap: a room world spanning the grid (minus a margin),
// rendered from `num_poses` scans of `beams` beams.  The grid is NOT allowed
// to grow (hits are clipped to the room), so cells_out is nx*ny.
// Returns the world handle (caller destroys) or null.
void* cmx_synth_submap(uint64_t seed, int nx, int ny, double resolution, int num_poses,
                       int beams, double max_range, double sigma, uint16_t* cells_out,
                       double* max_xy_out) {
  using namespace cartographer_amd::host;
  // MapLimits: x extent = ny cells, y extent = nx cells (map_limits.h:69-76).
  const double ext_x = ny * resolution, ext_y = nx * resolution;
  const double max_x = 0.5 * ext_x, max_y = 0.5 * ext_y;
  const double margin = std::max(0.75, 12 * resolution);
  auto* w = static_cast<World*>(cmx_synth_world_create(seed, max_x - ext_x + margin,
                                                       max_x - margin, max_y - ext_y + margin,
                                                       max_y - margin));
  ProbabilityGridBuilder grid(resolution, max_x, max_y, nx, ny);
  const auto hit = ComputeLookupTableToApplyCorrespondenceCostOdds(Odds(0.7f));
  const auto miss = ComputeLookupTableToApplyCorrespondenceCostOdds(Odds(0.4f));
  std::vector<float> sensor(3 * static_cast<size_t>(beams)), mapf(3 * static_cast<size_t>(beams));
  for (int p = 0; p != num_poses; ++p) {
    double pose[3];
    cmx_synth_world_free_pose(w, seed * 1000003ull + p, 0.3, pose);
    const int n = cmx_synth_scan(w, pose, beams, max_range, sigma, seed * 7919ull + p,
                                 sensor.data());
    const double c = std::cos(pose[2]), s = std::sin(pose[2]);
    int m = 0;
    for (int i = 0; i != n; ++i) {
      const double x = pose[0] + c * sensor[3 * i] - s * sensor[3 * i + 1];
      const double y = pose[1] + s * sensor[3 * i] + c * sensor[3 * i + 1];
      // keep strictly inside the grid so it never grows
      if (x <= max_x - ext_x + 2 * resolution || x >= max_x - 2 * resolution ||
          y <= max_y - ext_y + 2 * resolution || y >= max_y - 2 * resolution)
        continue;
      mapf[3 * m] = static_cast<float>(x); mapf[3 * m + 1] = static_cast<float>(y);
      mapf[3 * m + 2] = 0.f;
      ++m;
    }
    const float origin[2] = {static_cast<float>(pose[0]), static_cast<float>(pose[1])};
    grid.Insert(origin, mapf.data(), m, nullptr, 0, hit, miss, true);
  }
  if (grid.num_x_cells() != nx || grid.num_y_cells() != ny) {
    delete w;
    return nullptr;
  }
  std::memcpy(cells_out, grid.cells().data(), grid.cells().size() * sizeof(uint16_t));
  max_xy_out[0] = grid.max_x();
  max_xy_out[1] = grid.max_y();
  return w;
}

}  // extern "C"

// ============================================================== 3D ==========
#include "hybrid_grid_builder.h"

using cartographer_amd::host::HybridGridBuilder;
using cartographer_amd::host::VoxelRecord;

namespace {
struct Box3 { double lo[3], hi[3]; };
struct World3 {
  Box3 room;                 // rays start inside and hit its walls from the inside
  std::vector<Box3> boxes;   // solid obstacles
  double Raycast(const double o[3], const double d[3]) const {
    double best = std::numeric_limits<double>::infinity();
    // room: exit distance
    double t_exit = std::numeric_limits<double>::infinity();
    for (int k = 0; k != 3; ++k) {
      if (d[k] > 1e-12) t_exit = std::min(t_exit, (room.hi[k] - o[k]) / d[k]);
      else if (d[k] < -1e-12) t_exit = std::min(t_exit, (room.lo[k] - o[k]) / d[k]);
    }
    best = t_exit;
    for (const Box3& b : boxes) {
      double t0 = 0., t1 = std::numeric_limits<double>::infinity();
      bool miss = false;
      for (int k = 0; k != 3 && !miss; ++k) {
        if (std::abs(d[k]) < 1e-12) {
          if (o[k] < b.lo[k] || o[k] > b.hi[k]) miss = true;
        } else {
          double a = (b.lo[k] - o[k]) / d[k], c = (b.hi[k] - o[k]) / d[k];
          if (a > c) std::swap(a, c);
          t0 = std::max(t0, a);
          t1 = std::min(t1, c);
          if (t0 > t1) miss = true;
        }
      }
      if (!miss && t0 > 1e-9) best = std::min(best, t0);
    }
    return best;
  }
  bool Free(const double p[3], double margin) const {
    for (int k = 0; k != 3; ++k)
      if (p[k] < room.lo[k] + margin || p[k] > room.hi[k] - margin) return false;
    for (const Box3& b : boxes) {
      bool inside = true;
      for (int k = 0; k != 3; ++k)
        if (p[k] < b.lo[k] - margin || p[k] > b.hi[k] + margin) inside = false;
      if (inside) return false;
    }
    return true;
  }
};
}  // namespace

extern "C" {

void* cmx_synth3d_world_create(uint64_t seed, double size_x, double size_y, double size_z) {
  auto* w = new World3;
  w->room = {{-0.5 * size_x, -0.5 * size_y, 0.}, {0.5 * size_x, 0.5 * size_y, size_z}};
  Rng rng(seed + 977);
  const int n = 5 + static_cast<int>(rng.next() % 4);
  for (int i = 0; i != n; ++i) {
    Box3 b;
    const double sx = rng.uniform(0.05, 0.15) * size_x, sy = rng.uniform(0.05, 0.15) * size_y;
    const double sz = rng.uniform(0.2, 0.8) * size_z;
    const double cx = rng.uniform(-0.5 * size_x + sx, 0.5 * size_x - sx);
    const double cy = rng.uniform(-0.5 * size_y + sy, 0.5 * size_y - sy);
    b.lo[0] = cx - 0.5 * sx; b.hi[0] = cx + 0.5 * sx;
    b.lo[1] = cy - 0.5 * sy; b.hi[1] = cy + 0.5 * sy;
    b.lo[2] = 0.; b.hi[2] = sz;
    w->boxes.push_back(b);
  }
  return w;
}
void cmx_synth3d_world_destroy(void* world) { delete static_cast<World3*>(world); }

void cmx_synth3d_free_position(void* world, uint64_t seed, double clearance, double* xyz) {
  const World3& w = *static_cast<World3*>(world);
  Rng rng(seed ^ 0x3D3D3Dull);
  for (int tries = 0; tries != 100000; ++tries) {
    double p[3];
    for (int k = 0; k != 2; ++k) p[k] = rng.uniform(w.room.lo[k], w.room.hi[k]);
    p[2] = rng.uniform(0.8, 1.8);
    if (w.Free(p, clearance)) { xyz[0] = p[0]; xyz[1] = p[1]; xyz[2] = p[2]; return; }
  }
  xyz[0] = xyz[1] = 0.; xyz[2] = 1.;
}

// `rings` elevation rings in [-elev, +elev] x `azimuths` bearings; sensor pose
// = translation + yaw.  Hits within max_range, as float xyz in the SENSOR frame.
int cmx_synth3d_scan(void* world, const double* position_xyz, double yaw, int rings,
                     int azimuths, double elev, double max_range, double sigma, uint64_t seed,
                     float* xyz_out) {
  const World3& w = *static_cast<World3*>(world);
  Rng rng(seed ^ 0x77AAull);
  int n = 0;
  for (int r = 0; r != rings; ++r) {
    const double el = rings > 1 ? -elev + 2. * elev * r / (rings - 1) : 0.;
    for (int a = 0; a != azimuths; ++a) {
      const double az = 2. * M_PI * a / azimuths;
      const double ds[3] = {std::cos(el) * std::cos(az), std::cos(el) * std::sin(az), std::sin(el)};
      const double dw[3] = {std::cos(yaw) * ds[0] - std::sin(yaw) * ds[1],
                            std::sin(yaw) * ds[0] + std::cos(yaw) * ds[1], ds[2]};
      double t = w.Raycast(position_xyz, dw);
      const double noise = sigma * rng.normal();
      if (!(t < max_range)) continue;
      t += noise;
      xyz_out[3 * n] = static_cast<float>(t * ds[0]);
      xyz_out[3 * n + 1] = static_cast<float>(t * ds[1]);
      xyz_out[3 * n + 2] = static_cast<float>(t * ds[2]);
      ++n;
    }
  }
  return n;
}

void* cmx_hgrid_create(float resolution) { return new HybridGridBuilder(resolution); }
void cmx_hgrid_destroy(void* g) { delete static_cast<HybridGridBuilder*>(g); }
int cmx_hgrid_size(void* g) { return static_cast<HybridGridBuilder*>(g)->grid_size(); }
void cmx_hgrid_set_probability(void* g, int x, int y, int z, float p) {
  static_cast<HybridGridBuilder*>(g)->SetProbability(x, y, z, p);
}
float cmx_hgrid_get_probability(void* g, int x, int y, int z) {
  return static_cast<HybridGridBuilder*>(g)->GetProbability(x, y, z);
}
void cmx_hgrid_cell_index(void* g, const float* p, int* out) {
  static_cast<HybridGridBuilder*>(g)->GetCellIndex(p, out);
}
void cmx_hgrid_insert(void* g, const float* origin, const float* returns_xyz, int n,
                      float hit_probability, float miss_probability, int num_free_space_voxels) {
  using namespace cartographer_amd::host;
  auto odds = [](float p) { return p / (1.f - p); };
  static_cast<HybridGridBuilder*>(g)->Insert(
      origin, returns_xyz, n, ComputeLookupTableToApplyOdds(odds(hit_probability)),
      ComputeLookupTableToApplyOdds(odds(miss_probability)), num_free_space_voxels);
}
int64_t cmx_hgrid_num_voxels(void* g) {
  return static_cast<int64_t>(static_cast<HybridGridBuilder*>(g)->Voxels().size());
}
void cmx_hgrid_voxels(void* g, void* out /* VoxelRecord[] */) {
  const auto v = static_cast<HybridGridBuilder*>(g)->Voxels();
  std::memcpy(out, v.data(), v.size() * sizeof(VoxelRecord));
}

}  // extern "C"
