// RealTimeCorrelativeScanMatcher3D::Match on gfx950, plus the dense-brick
// helpers shared with the fast 3D matcher.
//
// Reference: SM3/real_time_correlative_scan_matcher_3d.cc:34-53 (Match),
// :55-95 (GenerateExhaustiveSearchTransforms), :97-114 (ScoreCandidate).
//
// Parity notes
//   * Candidate = init.cast<float>() * Rigid3f(t, q) (renormalised product).
//     The (2A+1)^3 rotations and (2L+1)^3 translations are composed on the
//     host with the reference's f32/f64 operation order (sin/cos/atan2 from
//     libm); the device only performs IEEE +,-,*,/ on them.
//   * The reference transforms every point by every candidate and sums the N
//     probabilities sequentially in f32.  One thread per candidate does exactly
//     that, so the unweighted score is bit-identical; a wave holds 64
//     translations of one rotation, so the point load and the rotation are
//     wave-uniform and neighbouring lanes read neighbouring voxels.
//   * exp() weighting and the strict-'>' first-maximum rule are finished on the
//     host for the candidates within 1e-5 (relative) of the device maximum.
#include <algorithm>

#include "scan_matching_3d.h"

namespace cmx {
namespace {

__global__ void ScatterVoxelsKernel(const cmx_voxel* __restrict__ voxels, long long n, Brick b,
                                    int bytes_per_cell) {
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const cmx_voxel v = voxels[i];
  const size_t off = (static_cast<size_t>(v.z - b.lo_z) * b.ny + (v.y - b.lo_y)) * b.nx +
                     (v.x - b.lo_x);
  if (bytes_per_cell == 2) {
    static_cast<uint16_t*>(const_cast<void*>(b.cells))[off] = v.value;
  } else {
    // ConvertToPrecomputationGrid (SM3/precomputation_grid_3d.cc:49-62).
    const float kMinP = 0.1f;
    const float kMaxP = 1.f - kMinP;
    int value = LRoundF32((ValueToProbabilityDev(v.value) - kMinP) * (255.f / (kMaxP - kMinP)));
    value = min(max(value, 0), 255);
    static_cast<uint8_t*>(const_cast<void*>(b.cells))[off] = static_cast<uint8_t>(value);
  }
}

// Dense f32 probability brick with a one-cell border of kMinProbability: a voxel
// index clamped to the border reads what HybridGrid::GetProbability returns for any
// cell outside the stored box (value 0 -> kMinProbability, hybrid_grid.h:521-523).
struct PaddedBrick {
  const float* cells;   // [(nz+2)][(ny+2)][(nx+2)]
  int lo_x, lo_y, lo_z; // index of padded cell (1,1,1)
  int nx, ny, nz;
};

__global__ void FillFloatKernel(float* __restrict__ out, size_t n, float value) {
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) out[i] = value;
}

__global__ void ScatterProbabilitiesKernel(const cmx_voxel* __restrict__ voxels, long long n,
                                           PaddedBrick b, float* __restrict__ out) {
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const cmx_voxel v = voxels[i];
  const size_t off = (static_cast<size_t>(v.z - b.lo_z + 1) * (b.ny + 2) + (v.y - b.lo_y + 1)) *
                         (b.nx + 2) + (v.x - b.lo_x + 1);
  out[off] = ValueToProbabilityDev(v.value);
}

struct Rt3DParams {
  PaddedBrick grid;
  float resolution, inv_resolution;
  int num_translations, num_rotations, side_t, side_r;
  const float4* rotation;     // [R] candidate rotation (x,y,z,w) = normalized(init.q * q_r)
  const float4* translation;  // [T] candidate translation (xyz) = init.q * t_c + init.t; w = |t_c|
  const float* rotation_angle;  // [R] GetAngle(transform)
  double wt, wr;
};

// lround(c / resolution) (HybridGrid::GetCellIndex, hybrid_grid.h:428-433) without the
// IEEE division in the common case.  q0 = c * RN(1/res) differs from the correctly
// rounded quotient qe by less than |q0| * 2^-21; when q0 is further than |q0| * 2^-20
// from every half-integer, qe lies strictly inside (n - 0.5, n + 0.5) with n = rint(q0),
// so lround(qe) == n.  Otherwise (about |q0| * 2^-19 of the inputs, NaN/inf included)
// `ok` is cleared and the caller recomputes with the reference's exact expression.
__device__ __forceinline__ int FastCellIndex(float c, float inv_resolution, bool* ok) {
  const float q0 = c * inv_resolution;
  const float n = rintf(q0);
  const float margin = 0.5f - fabsf(q0 - n);          // exact
  *ok = *ok && (margin > fabsf(q0) * 0x1p-20f);
  return static_cast<int>(n);
}

// One wavefront = 64 translations of one rotation.  Each lane rotates one point of the
// next 64-point chunk into LDS (the rotation is wave-uniform, so this removes the 64x
// redundant rotation), then all lanes walk the chunk in point order: broadcast LDS read,
// add the lane's translation, cell index, one gather from the padded brick, and the
// reference's sequential f32 accumulation.
__global__ void __launch_bounds__(64)
Rt3DScoreKernel(Rt3DParams P, const float* __restrict__ xyz, int n,
                float* __restrict__ unweighted, float* __restrict__ weighted,
                unsigned* __restrict__ max_bits) {
  __shared__ float4 rotated[64];
  const int lane = threadIdx.x;
  const int t = blockIdx.x * 64 + lane;
  const int r = blockIdx.y;
  const bool valid = t < P.num_translations;
  const float4 q4 = P.rotation[r];
  const Quat q{q4.w, q4.x, q4.y, q4.z};
  const float4 tr = P.translation[valid ? t : 0];
  const float res = P.resolution, inv = P.inv_resolution;
  const int sx = P.grid.nx + 2, sy = P.grid.ny + 2;
  const int ox = 1 - P.grid.lo_x, oy = 1 - P.grid.lo_y, oz = 1 - P.grid.lo_z;
  const int mx = P.grid.nx + 1, my = P.grid.ny + 1, mz = P.grid.nz + 1;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(P.grid.cells), 0, sx * sy * (P.grid.nz + 2) * 4, 0x00020000);

  const auto cell_offset = [&](const float4& rp) -> int {
    const float cx = rp.x + tr.x, cy = rp.y + tr.y, cz = rp.z + tr.z;   // rigid * point
    bool ok = true;
    int ix = FastCellIndex(cx, inv, &ok);
    int iy = FastCellIndex(cy, inv, &ok);
    int iz = FastCellIndex(cz, inv, &ok);
    if (!ok) {
      const int3 idx = CellIndex3(F3{cx, cy, cz}, res);
      ix = idx.x; iy = idx.y; iz = idx.z;
    }
    ix = min(max(ix + ox, 0), mx);
    iy = min(max(iy + oy, 0), my);
    iz = min(max(iz + oz, 0), mz);
    return ((iz * sy + iy) * sx + ix) * 4;
  };

  float acc = 0.f;
  constexpr int kBatch = 8;
  for (int base = 0; base < n; base += 64) {
    const int cnt = min(64, n - base);
    __syncthreads();                       // previous chunk fully consumed
    if (lane < cnt) {
      const int i = base + lane;
      const F3 rp = Rotate(q, F3{xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]});
      rotated[lane] = make_float4(rp.x, rp.y, rp.z, 0.f);
    }
    __syncthreads();
    int j = 0;
    for (; j + kBatch <= cnt; j += kBatch) {
      float v[kBatch];
#pragma unroll
      for (int k = 0; k < kBatch; ++k)
        v[k] = __uint_as_float(
            __builtin_amdgcn_raw_buffer_load_b32(rsrc, cell_offset(rotated[j + k]), 0, 0));
#pragma unroll
      for (int k = 0; k < kBatch; ++k) acc += v[k];
    }
    for (; j < cnt; ++j)
      acc += __uint_as_float(
          __builtin_amdgcn_raw_buffer_load_b32(rsrc, cell_offset(rotated[j]), 0, 0));
  }

  float w = 0.f;
  if (valid) {
    acc /= static_cast<float>(n);
    // Candidate order of the reference: z, y, x, rz, ry, rx nesting.
    const size_t c = static_cast<size_t>(t) * P.num_rotations + r;
    unweighted[c] = acc;
    const double penalty = static_cast<double>(tr.w) * P.wt +
                           static_cast<double>(P.rotation_angle[r]) * P.wr;
    w = static_cast<float>(static_cast<double>(acc) * exp(-(penalty * penalty)));
    weighted[c] = w;
  }
  unsigned bits = __float_as_uint(w);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) bits = max(bits, __shfl_xor(bits, off, 64));
  if (threadIdx.x == 0) atomicMax(max_bits, bits);
}

__global__ void Rt3DCollectKernel(const float* __restrict__ weighted, long long num_candidates,
                                  const unsigned* __restrict__ max_bits, int* __restrict__ count,
                                  long long* __restrict__ finalists, int capacity) {
  const long long c = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (c >= num_candidates) return;
  const float threshold = __uint_as_float(*max_bits) * (1.f - 1e-5f);
  if (weighted[c] >= threshold) {
    const int slot = atomicAdd(count, 1);
    if (slot < capacity) finalists[slot] = c;
  }
}

}  // namespace

bool VoxelBounds(const cmx_voxel* voxels, int64_t n, int lo[3], int hi[3]) {
  if (n <= 0) return false;
  lo[0] = hi[0] = voxels[0].x; lo[1] = hi[1] = voxels[0].y; lo[2] = hi[2] = voxels[0].z;
  for (int64_t i = 1; i < n; ++i) {
    lo[0] = std::min(lo[0], voxels[i].x); hi[0] = std::max(hi[0], voxels[i].x);
    lo[1] = std::min(lo[1], voxels[i].y); hi[1] = std::max(hi[1], voxels[i].y);
    lo[2] = std::min(lo[2], voxels[i].z); hi[2] = std::max(hi[2], voxels[i].z);
  }
  return true;
}

int GridSizeOf(const cmx_voxel* voxels, int64_t n) {
  int gs = 128;
  int lo[3], hi[3];
  if (!VoxelBounds(voxels, n, lo, hi)) return gs;
  auto fits = [&](int g) {
    const int h = g / 2;
    for (int k = 0; k < 3; ++k)
      if (lo[k] < -h || hi[k] >= h) return false;
    return true;
  };
  while (!fits(gs)) gs *= 2;
  return gs;
}

void BuildBrickFromVoxels(Workspace& ws, const cmx_voxel* voxels, int64_t n, int bytes_per_cell,
                          DeviceBrick* out) {
  int lo[3], hi[3];
  if (!VoxelBounds(voxels, n, lo, hi)) {
    lo[0] = lo[1] = lo[2] = 0;
    hi[0] = hi[1] = hi[2] = 0;
  }
  for (int k = 0; k < 3; ++k) {
    CMX_REQUIRE(lo[k] > -(1 << 20) && hi[k] < (1 << 20), "voxel index out of range");
  }
  Brick b{};
  b.lo_x = lo[0]; b.lo_y = lo[1]; b.lo_z = lo[2];
  b.nx = hi[0] - lo[0] + 1; b.ny = hi[1] - lo[1] + 1; b.nz = hi[2] - lo[2] + 1;
  const size_t cells = static_cast<size_t>(b.nx) * b.ny * b.nz;
  CMX_REQUIRE(cells * bytes_per_cell < (size_t(8) << 30),
              "dense grid of %d x %d x %d cells is too large", b.nx, b.ny, b.nz);
  out->bytes = cells * bytes_per_cell;
  CMX_HIP(hipMalloc(&out->mem, out->bytes));
  b.cells = out->mem;
  out->desc = b;
  CMX_HIP(hipMemsetAsync(out->mem, 0, out->bytes, ws.stream));
  if (n > 0) {
    cmx_voxel* d_vox = ws.dev[15].ReserveAs<cmx_voxel>(n);
    CMX_HIP(hipMemcpyAsync(d_vox, voxels, n * sizeof(cmx_voxel), hipMemcpyHostToDevice,
                           ws.stream));
    ScatterVoxelsKernel<<<DivUp(n, 256), 256, 0, ws.stream>>>(d_vox, n, b, bytes_per_cell);
    CMX_HIP(hipGetLastError());
  }
  CMX_HIP(hipStreamSynchronize(ws.stream));   // `voxels` is borrowed host memory
}

}  // namespace cmx

extern "C" cmx_status cmx_rt3d_match(const cmx_rt_options* options, float grid_resolution,
                                     const cmx_voxel* voxels, int64_t num_voxels,
                                     const cmx_pose3d* initial_pose_estimate,
                                     const float* point_cloud_xyz, int32_t num_points,
                                     int32_t device, float* score, cmx_pose3d* pose_estimate,
                                     cmx_match_stats* stats) {
  using namespace cmx;
  return Guard([&] {
    CMX_REQUIRE(options && initial_pose_estimate && point_cloud_xyz, "null argument");
    CMX_REQUIRE(pose_estimate != nullptr && score != nullptr,
                "pose_estimate must not be null");                       // CHECK at :39
    CMX_REQUIRE(num_voxels == 0 || voxels != nullptr, "voxels is null");
    CMX_REQUIRE(num_points >= 1 && num_points <= (1 << 24), "bad point count");
    CMX_REQUIRE(grid_resolution > 0.f, "resolution must be > 0");
    const int n = num_points;
    const float resolution = grid_resolution;

    // GenerateExhaustiveSearchTransforms (:55-95), host side.
    const int L = static_cast<int>(std::lround(options->linear_search_window / resolution));
    float max_scan_range = 3.f * resolution;
    for (int i = 0; i < n; ++i) {
      const h3::V3 p{point_cloud_xyz[3 * i], point_cloud_xyz[3 * i + 1],
                     point_cloud_xyz[3 * i + 2]};
      max_scan_range = std::max(h3::Norm(p), max_scan_range);
    }
    const float kSafetyMargin = 1.f - 1e-3f;
    const float step =
        kSafetyMargin * std::acos(1.f - (resolution * (resolution * 1.f)) /
                                            (2.f * (max_scan_range * (max_scan_range * 1.f))));
    const int A = static_cast<int>(std::lround(options->angular_search_window / step));
    CMX_REQUIRE(L >= 0 && L < 512 && A >= 0 && A < 64, "unsupported search window");
    const int side_t = 2 * L + 1, side_r = 2 * A + 1;
    const long long T = 1ll * side_t * side_t * side_t, R = 1ll * side_r * side_r * side_r;
    const long long num_candidates = T * R;
    CMX_REQUIRE(num_candidates < (1ll << 31), "search window too large");

    const h3::Rigid init = h3::FromPose(*initial_pose_estimate);
    std::vector<float4> rot(R), trans(T);
    std::vector<float> angle(R);
    std::vector<h3::Q> rot_q(R);
    {
      long long k = 0;
      for (int rz = -A; rz <= A; ++rz)
        for (int ry = -A; ry <= A; ++ry)
          for (int rx = -A; rx <= A; ++rx, ++k) {
            h3::Rigid tf;
            tf.q = h3::FromAngleAxisVector({rx * step, ry * step, rz * step});
            rot_q[k] = tf.q;
            angle[k] = h3::GetAngle(tf);
            const h3::Q q = h3::Normalized(h3::Mul(init.q, tf.q));
            rot[k] = make_float4(q.x, q.y, q.z, q.w);
          }
      k = 0;
      for (int z = -L; z <= L; ++z)
        for (int y = -L; y <= L; ++y)
          for (int x = -L; x <= L; ++x, ++k) {
            const h3::V3 tc{x * resolution, y * resolution, z * resolution};
            const h3::V3 r = h3::Rotate(init.q, tc);
            trans[k] = make_float4(r.x + init.t.x, r.y + init.t.y, r.z + init.t.z, h3::Norm(tc));
          }
    }

    WorkspaceLease ws(device);
    // Padded f32 probability brick over the voxels' bounding box.
    PaddedBrick brick{};
    {
      int lo[3], hi[3];
      if (!VoxelBounds(voxels, num_voxels, lo, hi)) {
        lo[0] = lo[1] = lo[2] = 0;
        hi[0] = hi[1] = hi[2] = 0;
      }
      for (int k = 0; k < 3; ++k)
        CMX_REQUIRE(lo[k] > -(1 << 20) && hi[k] < (1 << 20), "voxel index out of range");
      brick.lo_x = lo[0]; brick.lo_y = lo[1]; brick.lo_z = lo[2];
      brick.nx = hi[0] - lo[0] + 1; brick.ny = hi[1] - lo[1] + 1; brick.nz = hi[2] - lo[2] + 1;
      const size_t cells = static_cast<size_t>(brick.nx + 2) * (brick.ny + 2) * (brick.nz + 2);
      CMX_REQUIRE(cells < (size_t(1) << 29), "dense grid of %d x %d x %d cells is too large",
                  brick.nx, brick.ny, brick.nz);
      float* d_cells = ws->dev[7].ReserveAs<float>(cells);
      brick.cells = d_cells;
      FillFloatKernel<<<DivUp(cells, 256), 256, 0, ws->stream>>>(d_cells, cells, 0.1f);
      if (num_voxels > 0) {
        cmx_voxel* d_vox = ws->dev[15].ReserveAs<cmx_voxel>(num_voxels);
        CMX_HIP(hipMemcpyAsync(d_vox, voxels, num_voxels * sizeof(cmx_voxel),
                               hipMemcpyHostToDevice, ws->stream));
        ScatterProbabilitiesKernel<<<DivUp(num_voxels, 256), 256, 0, ws->stream>>>(
            d_vox, num_voxels, brick, d_cells);
      }
      CMX_HIP(hipGetLastError());
    }

    float* d_xyz = ws->dev[0].ReserveAs<float>(3 * static_cast<size_t>(n));
    float4* d_rot = ws->dev[1].ReserveAs<float4>(R);
    float4* d_trans = ws->dev[2].ReserveAs<float4>(T);
    float* d_angle = ws->dev[3].ReserveAs<float>(R);
    float* d_unweighted = ws->dev[4].ReserveAs<float>(num_candidates);
    float* d_weighted = ws->dev[5].ReserveAs<float>(num_candidates);
    const int kFinalistCap = 4096;
    char* d_misc = static_cast<char*>(ws->dev[6].Reserve(16 + sizeof(long long) * kFinalistCap));
    unsigned* d_max = reinterpret_cast<unsigned*>(d_misc);
    int* d_count = reinterpret_cast<int*>(d_misc + 4);
    long long* d_finalists = reinterpret_cast<long long*>(d_misc + 16);
    char* h_misc = static_cast<char*>(ws->pinned[1].Reserve(16 + sizeof(long long) * kFinalistCap));

    CMX_HIP(hipMemcpyAsync(d_xyz, point_cloud_xyz, 3 * sizeof(float) * n, hipMemcpyHostToDevice,
                           ws->stream));
    CMX_HIP(hipMemcpyAsync(d_rot, rot.data(), R * sizeof(float4), hipMemcpyHostToDevice,
                           ws->stream));
    CMX_HIP(hipMemcpyAsync(d_trans, trans.data(), T * sizeof(float4), hipMemcpyHostToDevice,
                           ws->stream));
    CMX_HIP(hipMemcpyAsync(d_angle, angle.data(), R * sizeof(float), hipMemcpyHostToDevice,
                           ws->stream));
    CMX_HIP(hipMemsetAsync(d_misc, 0, 16, ws->stream));
    // pageable sources above: make sure the copies are done before they go away
    CMX_HIP(hipStreamSynchronize(ws->stream));

    Rt3DParams P;
    P.grid = brick;
    P.resolution = resolution;
    P.inv_resolution = 1.f / resolution;
    P.num_translations = static_cast<int>(T);
    P.num_rotations = static_cast<int>(R);
    P.side_t = side_t; P.side_r = side_r;
    P.rotation = d_rot; P.translation = d_trans; P.rotation_angle = d_angle;
    P.wt = options->translation_delta_cost_weight;
    P.wr = options->rotation_delta_cost_weight;

    CMX_HIP(hipEventRecord(ws->ev_begin, ws->stream));
    CMX_HIP(hipEventRecord(ws->ev_k0, ws->stream));
    Rt3DScoreKernel<<<dim3(DivUp(T, 64), static_cast<unsigned>(R)), 64, 0, ws->stream>>>(
        P, d_xyz, n, d_unweighted, d_weighted, d_max);
    CMX_HIP(hipEventRecord(ws->ev_k1, ws->stream));
    Rt3DCollectKernel<<<DivUp(num_candidates, 256), 256, 0, ws->stream>>>(
        d_weighted, num_candidates, d_max, d_count, d_finalists, kFinalistCap);
    CMX_HIP(hipGetLastError());
    CMX_HIP(hipEventRecord(ws->ev_end, ws->stream));
    CMX_HIP(hipMemcpyAsync(h_misc, d_misc, 16 + sizeof(long long) * kFinalistCap,
                           hipMemcpyDeviceToHost, ws->stream));
    CMX_HIP(hipStreamSynchronize(ws->stream));

    const int count = *reinterpret_cast<int*>(h_misc + 4);
    const long long* h_finalists = reinterpret_cast<long long*>(h_misc + 16);
    std::vector<long long> finalists;
    std::vector<float> acc;
    if (count <= kFinalistCap) {
      finalists.assign(h_finalists, h_finalists + count);
      std::sort(finalists.begin(), finalists.end());
      acc.resize(count);
      for (int i = 0; i < count; ++i)
        CMX_HIP(hipMemcpy(&acc[i], d_unweighted + finalists[i], sizeof(float),
                          hipMemcpyDeviceToHost));
    } else {
      finalists.resize(num_candidates);
      for (long long c = 0; c < num_candidates; ++c) finalists[c] = c;
      acc.resize(num_candidates);
      CMX_HIP(hipMemcpy(acc.data(), d_unweighted, sizeof(float) * num_candidates,
                        hipMemcpyDeviceToHost));
    }
    CMX_REQUIRE(!finalists.empty(), "internal error: no candidate collected");
    // Exact weighting and the strict '>' running maximum of :44-50.
    float best_score = -1.f;
    long long best = -1;
    for (size_t i = 0; i < finalists.size(); ++i) {
      const long long c = finalists[i];
      const long long t = c / R, r = c % R;
      float sc = acc[i];
      const double penalty = static_cast<double>(trans[t].w) * P.wt +
                             static_cast<double>(angle[r]) * P.wr;
      sc *= std::exp(-(penalty * (penalty * 1.)));
      if (sc > best_score) { best_score = sc; best = c; }
    }
    {
      const long long t = best / R, r = best % R;
      h3::Rigid candidate;
      candidate.t = {trans[t].x, trans[t].y, trans[t].z};
      candidate.q = {rot[r].w, rot[r].x, rot[r].y, rot[r].z};
      *pose_estimate = h3::ToPose(candidate);
      *score = best_score;
    }
    if (stats) {
      cmx_match_stats st{};
      st.candidates_scored = num_candidates;
      st.coarse_candidates = num_candidates;
      st.num_scans = static_cast<int>(R);
      float ms = 0.f;
      CMX_HIP(hipEventElapsedTime(&ms, ws->ev_begin, ws->ev_end));
      st.device_ms = ms;
      CMX_HIP(hipEventElapsedTime(&ms, ws->ev_k0, ws->ev_k1));
      st.dominant_kernel_ms = ms;
      *stats = st;
    }
  });
}
