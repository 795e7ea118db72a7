// RealTimeCorrelativeScanMatcher2D::Match, the block-bound level of the bulk pass (round 5).
// Included by rt_2d_tiles.hip inside its namespace, behind Rt2DTileParams and its helpers.
//
// Reference: SM2/real_time_correlative_scan_matcher_2d.cc:117-176 scores EVERY candidate of the
// (2 nl + 1)^2 x num_scans search space.  The tile kernel (Rt2DTileKernel) does the same with
// integer sums out of an LDS image: (2 nl + 1)^2 window cells per (point, rotation), and its
// counters say the vector ALU is what it is bound by.  Here the search space is pruned first, and
// the result stays the reference's bit for bit:
//
//   * per grid VERSION, next to the quantised image: the image max-pooled over 2 x 2 cells,
//     m2(X, Y) = ceil(max u(X .. X + 1, Y .. Y + 1) / kBoundUnit) <= kBoundMax = 31 in one byte
//     (u = 32767 - value <= 32766 <= 31 x 1057, so kBoundUnit m2 >= u for the four cells; five
//     bits: eight of them add up inside a byte), stored as FOUR PARITY PLANES
//     plane(Y & 1, X & 1)[Y >> 1][X >> 1] (Rt2DPoolKernel);
//   * a BLOCK = 2 x 2 translations of one rotation.  For a point with window start (Xs, Ys) the
//     block (j, k) covers the cells (Xs + 2 k .. + 1, Ys + 2 j .. + 1): one byte m2(Xs + 2 k,
//     Ys + 2 j) bounds the point's contribution to all four candidates from above.  The NB block
//     columns of a window row are NB CONSECUTIVE bytes of the plane the window start's parity
//     selects: three aligned dwords per block row, shifted into place, NB rows per (point,
//     rotation) instead of (2 nl + 1)^2 window cells -- C1: 7 rows, their 14 dwords added as they
//     are (bytes sum without carries), instead of 169 cells;
//   * phase A: a wavefront takes a rotation, a lane a point; per-lane byte sums in registers, one
//     wave reduction per rotation: the upper bounds of all num_scans x NB^2 blocks in LDS;
//   * phase B: weighted upper bounds (the finish kernel's own formula: slack of the f32 chain,
//     1e-5 relative); the block with the best one has its four candidates summed exactly
//     (quantised cells out of the HBM image, as the tile kernel sums them): the best weighted
//     LOWER bound among them;
//   * phase C: every other block whose weighted upper bound reaches that lower bound is summed
//     the same way (C1: five to six blocks of the 1323, one per rotation around the best).
//
// The sums of the summed candidates go where the tile kernel puts its sums (qsum, quantised
// units), every other candidate keeps 0: Rt2DFinishMatch runs unchanged.  Why the result cannot
// change: a candidate is dropped only if ub(block) < lb strictly, with lb a lower bound (the
// finish kernel's formula) of the weighted f32-chain score of a candidate that IS summed and ub an
// upper bound of the same kind of every member of the block -- so the reference's maximum and
// everything that ties with it are summed, the finish kernel's best lower bound is taken among the
// summed candidates (the candidate that attains it cannot have been dropped: lb_f(c) <= score(c)
// <= ub(block(c))), and its finalists are a subset of the exhaustive run's that contains the
// winner and its ties.  The exhaustive tile kernel stays behind cmx_debug_set("rt2d_no_bounds", 1)
// as the parity partner (tests/test_gpu_r2_paths.py).
#ifndef CMX_RT_2D_BOUNDS_H_
#define CMX_RT_2D_BOUNDS_H_

constexpr int kBoundThreads = 512;          // eight wavefronts; two workgroups per CU
constexpr int kBoundMinMatches = 96;        // matches per call from which the bound kernel is the default
constexpr int kBoundMaxBlocks = 8;          // block columns in one ds_read_b64: side <= 16
#ifndef CMX_RT2D_BOUND_BITS
#define CMX_RT2D_BOUND_BITS 5
#endif
constexpr int kBoundMax = (1 << CMX_RT2D_BOUND_BITS) - 1;          // largest pooled value
constexpr int kBoundUnit = (32766 + kBoundMax - 1) / kBoundMax;   // u units per pooled unit: kBoundMax units >= 32766
constexpr int kBoundFlush = 255 / kBoundMax;                      // pooled values a byte sums without a carry
// (a wavefront's block sums are reduced as 16-bit fields: 64 lanes x 32 chunks x kBoundMax)
static_assert(64 * 32 * kBoundMax < 65536, "block sums of a rotation must fit 16 bits");
// (... and 32 chunks are what the fused path admits: kFusedMaxPoints, rt_2d_tiles.hip -- raising it
// must not overflow the fields unnoticed)
static_assert(kFusedMaxPoints <= 64 * 32, "the bound kernels' 16-bit lane fields hold 32 chunks of points");

// ---------------------------------------------------------------------------------------------
// grid (ceil(m2_rows * m2_pitch / 256), items): four bytes of each parity plane per thread.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
Rt2DPoolKernel(const Rt2DTileParams* __restrict__ params) {
  const Rt2DTileParams& P = params[blockIdx.y];
  if (!P.image_build || P.m2 == nullptr) return;
  const int wpr = P.m2_pitch >> 2;                       // dwords per plane row
  const int words = P.m2_rows * wpr;                     // per plane
  const int v = blockIdx.x * 256 + threadIdx.x;
  if (v >= 4 * words) return;
  const int plane = v / words, w = v - plane * words;
  const int row = w / wpr, col4 = (w - row * wpr) << 2;
  const int py = plane >> 1, px = plane & 1;
  const auto* cells = AsGlobal(P.cells);
  const auto u_of = [&](int X, int Y) -> unsigned {      // image (X, Y) = grid (X - hl, Y - ht)
    const int gx = X - P.hl, gy = Y - P.ht;
    if (static_cast<unsigned>(gx) >= static_cast<unsigned>(P.nx) ||
        static_cast<unsigned>(gy) >= static_cast<unsigned>(P.ny))
      return 0u;
    const unsigned raw = cells[gy * P.nx + gx] & 32767u;
    return raw ? 32767u - raw : 0u;
  };
  const int Y = 2 * row + py;
  unsigned out = 0;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int X = 2 * (col4 + c) + px;
    const unsigned m = max(max(u_of(X, Y), u_of(X + 1, Y)), max(u_of(X, Y + 1), u_of(X + 1, Y + 1)));
    out |= ((m + kBoundUnit - 1) / kBoundUnit) << (8 * c);
  }
  reinterpret_cast<uint32_t*>(const_cast<uint8_t*>(P.m2))[v] = out;
}


// ---------------------------------------------------------------------------------------------
// grid (ceil(m4_rows * m4_pitch / 256), items): the image max-pooled over 4 x 4 cells, from the
// 2 x 2 one: m4(X, Y) = max of m2 at (X, Y), (X + 2, Y), (X, Y + 2), (X + 2, Y + 2) -- ceil is
// monotone, so the maximum of the ceilings is the ceiling of the maximum -- as SIXTEEN phase
// planes plane(Y & 3, X & 3)[Y >> 2][X >> 2]: the NB block columns of a window row are NB
// consecutive bytes of the plane the window start's phase selects.  Four bytes per thread.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
Rt2DPool4Kernel(const Rt2DTileParams* __restrict__ params) {
  const Rt2DTileParams& P = params[blockIdx.y];
  if (!P.image_build || P.m4 == nullptr) return;
  const int wpr = P.m4_pitch >> 2;                       // dwords per plane row
  const int words = P.m4_rows * wpr;                     // per plane
  const int v = blockIdx.x * 256 + threadIdx.x;
  if (v >= 16 * words) return;
  const int plane = v / words, w = v - plane * words;
  const int row = w / wpr, col4 = (w - row * wpr) << 2;
  const int py = plane >> 2, px = plane & 3;
  const auto* m2 = AsGlobal(P.m2);
  const int plane2 = P.m2_rows * P.m2_pitch;
  const auto m2_of = [&](int X, int Y) -> unsigned {     // (beyond the stored planes: beyond the grid, 0)
    const int r = Y >> 1, c = X >> 1;
    if (r >= P.m2_rows || c >= P.m2_pitch) return 0u;
    return m2[(((Y & 1) << 1) | (X & 1)) * plane2 + r * P.m2_pitch + c];
  };
  const int Y = 4 * row + py;
  unsigned out = 0;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int X = 4 * (col4 + c) + px;
    const unsigned m = max(max(m2_of(X, Y), m2_of(X + 2, Y)), max(m2_of(X, Y + 2), m2_of(X + 2, Y + 2)));
    out |= m << (8 * c);
  }
  reinterpret_cast<uint32_t*>(const_cast<uint8_t*>(P.m4))[v] = out;
}


// ---------------------------------------------------------------------------------------------
// grid (tiles_x * tiles_y, items), 256 threads: ALL derived images of a grid in one launch (round
// 6; until then Rt2DQuantKernel, Rt2DPoolKernel and Rt2DPool4Kernel one after the other, every
// cell read four to sixteen times with its bounds checks: 130 us per 256 grids, which a caller
// that inserts a scan after every match pays on every call).  A workgroup takes a tile of
// 64 x 16 cells of the image: `u` of the tile and three cells of halo into LDS (each cell of
// the grid read once), the 2 x 2 maxima of 66 x 18 cells into LDS, then one DWORD per thread and
// image: the 10-bit cells (two), the bytes, the four parity planes, the sixteen phase planes.
// The kernels it replaces stay the statement of what every image holds (rt2d_image_kernels = 1
// launches them instead: the parity partner).
// ---------------------------------------------------------------------------------------------
constexpr int kImageTileX = 64, kImageTileY = 16;
__global__ void __launch_bounds__(256)
Rt2DImageKernel(const Rt2DTileParams* __restrict__ params, int tiles_x) {
  const Rt2DTileParams& P = params[blockIdx.y];
  if (!P.image_build) return;
  __shared__ uint16_t u[kImageTileY + 3][kImageTileX + 4];
  __shared__ uint8_t m2s[kImageTileY + 2][kImageTileX + 4];
  const int tid = threadIdx.x;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int X0 = tx * kImageTileX, Y0 = ty * kImageTileY;
  const int gw = P.gpitch >> 1;
  const bool planes = P.m2 != nullptr;
  const int xe = planes ? max(gw, max(2 * P.m2_pitch, 4 * P.m4_pitch)) : gw;
  const int ye = planes ? max(P.grows, max(2 * P.m2_rows, 4 * P.m4_rows)) : P.grows;
  if (X0 >= xe || Y0 >= ye) return;              // (uniform)
  const auto* cells = AsGlobal(P.cells);
  for (int i = tid; i < (kImageTileY + 3) * (kImageTileX + 3); i += 256) {
    const int r = i / (kImageTileX + 3), c = i - r * (kImageTileX + 3);
    const int gx = X0 + c - P.hl, gy = Y0 + r - P.ht;   // image (X, Y) = grid (X - hl, Y - ht)
    unsigned val = 0;
    if (static_cast<unsigned>(gx) < static_cast<unsigned>(P.nx) &&
        static_cast<unsigned>(gy) < static_cast<unsigned>(P.ny)) {
      const unsigned raw = cells[gy * P.nx + gx] & 32767u;
      val = raw ? 32767u - raw : 0u;
    }
    u[r][c] = static_cast<uint16_t>(val);
  }
  __syncthreads();
  {
    // the 10-bit cells (two dwords per thread) and the bytes (one)
    const int r = tid >> 4, c4 = (tid & 15) << 2;
    const int X = X0 + c4, Y = Y0 + r;
    if (Y < P.grows && X < gw) {                 // (gw is a multiple of 8: four cells in or out)
      const unsigned a = u[r][c4], b = u[r][c4 + 1], c = u[r][c4 + 2], d = u[r][c4 + 3];
      uint2v q;
      q.x = (a >> kQShift) | ((b >> kQShift) << 16);
      q.y = (c >> kQShift) | ((d >> kQShift) << 16);
      *reinterpret_cast<uint2v*>(P.qimage + static_cast<size_t>(Y) * gw + X) = q;
      if (P.q8)
        *reinterpret_cast<uint32_t*>(P.q8 + static_cast<size_t>(Y) * gw + X) =
            (a >> kQ8Shift) | ((b >> kQ8Shift) << 8) | ((c >> kQ8Shift) << 16) | ((d >> kQ8Shift) << 24);
    }
  }
  if (!planes) return;                           // (uniform)
  for (int i = tid; i < (kImageTileY + 2) * (kImageTileX + 2); i += 256) {
    const int r = i / (kImageTileX + 2), c = i - r * (kImageTileX + 2);
    const unsigned m = max(max(static_cast<unsigned>(u[r][c]), static_cast<unsigned>(u[r][c + 1])),
                           max(static_cast<unsigned>(u[r + 1][c]), static_cast<unsigned>(u[r + 1][c + 1])));
    m2s[r][c] = static_cast<uint8_t>((m + kBoundUnit - 1) / kBoundUnit);
  }
  __syncthreads();
  {
    // the parity planes: plane(Y & 1, X & 1)[Y >> 1][X >> 1] -- a thread's dword is four cells
    // two apart of one plane row
    const int plane = tid >> 6, rr = (tid >> 3) & 7, dw = tid & 7;
    const int py = plane >> 1, px = plane & 1;
    const int row = (Y0 >> 1) + rr, col = (X0 >> 1) + 4 * dw;
    if (row < P.m2_rows && col < P.m2_pitch) {   // (the pitch is a multiple of 4)
      const int r = 2 * rr + py, c = 8 * dw + px;
      const unsigned out = m2s[r][c] | (m2s[r][c + 2] << 8) | (m2s[r][c + 4] << 16) | (m2s[r][c + 6] << 24);
      *reinterpret_cast<uint32_t*>(const_cast<uint8_t*>(P.m2) + static_cast<size_t>(plane) * P.m2_rows * P.m2_pitch +
                                   static_cast<size_t>(row) * P.m2_pitch + col) = out;
    }
  }
  {
    // the phase planes of the 4 x 4 pooling: m4 = the maximum of four m2 two cells apart
    const int plane = tid >> 4, rr = (tid >> 2) & 3, dw = tid & 3;
    const int py = plane >> 2, px = plane & 3;
    const int row = (Y0 >> 2) + rr, col = (X0 >> 2) + 4 * dw;
    if (row < P.m4_rows && col < P.m4_pitch) {
      const int r = 4 * rr + py;
      unsigned out = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = 4 * (4 * dw + i) + px;
        const unsigned m = max(max(static_cast<unsigned>(m2s[r][c]), static_cast<unsigned>(m2s[r][c + 2])),
                               max(static_cast<unsigned>(m2s[r + 2][c]), static_cast<unsigned>(m2s[r + 2][c + 2])));
        out |= m << (8 * i);
      }
      *reinterpret_cast<uint32_t*>(const_cast<uint8_t*>(P.m4) + static_cast<size_t>(plane) * P.m4_rows * P.m4_pitch +
                                   static_cast<size_t>(row) * P.m4_pitch + col) = out;
    }
  }
}

// The discretisation of the tile kernel's fused path (see the long comment there): cells from a
// two-FMA f32 estimate where it provably equals the reference's rounding, the exact expressions
// for a whole chunk otherwise.  One set of constants per (match, rotation).
struct BoundDisc {
  float Ci, Si, Kx, Ky, bound_per_m, bound_fixed;
  float2 rot;
  int lo, ix_hi, iy_hi;
};
__device__ __forceinline__ BoundDisc MakeBoundDisc(const Rt2DTileParams& P, float2 rot) {
  BoundDisc D;
  const double inv_res = P.inv_res;
  const double Kyd = (P.max_y - static_cast<double>(P.ty)) * inv_res - 0.5;
  const double Kxd = (P.max_x - static_cast<double>(P.tx)) * inv_res - 0.5;
  const double wd = rot.x, zd = rot.y;
  D.Ky = static_cast<float>(Kyd);
  D.Kx = static_cast<float>(Kxd);
  D.Ci = static_cast<float>((1.0 - 2.0 * zd * zd) * inv_res);
  D.Si = static_cast<float>(2.0 * wd * zd * inv_res);
  const double bound_unit = 1.25 * 0x1p-24 * inv_res;
  D.bound_per_m = static_cast<float>(bound_unit * (4.0 + fmax(2.0 + 4.0 * zd * zd, 1.0 + 6.0 * fabs(zd))));
  D.bound_fixed = static_cast<float>(
      1.25 * 0x1p-24 * (inv_res * fmax(fabs(static_cast<double>(P.tx)), fabs(static_cast<double>(P.ty))) +
                        3.0 * fmax(fabs(Kxd), fabs(Kyd)) + 1.0));
  D.rot = rot;
  D.lo = -(P.nl + 1);
  D.ix_hi = P.nx + P.nl;
  D.iy_hi = P.ny + P.nl;
  return D;
}
// (wave-uniform control flow: every lane of the wavefront calls it for the same chunk)
__device__ __forceinline__ void BoundCellOf(const BoundDisc& D, const Rt2DFrame& F, float x, float y,
                                            bool valid, int* ix, int* iy) {
  const float tY = fmaf(-D.Ci, y, fmaf(-D.Si, x, D.Ky));   // cell x index from the map's y
  const float tX = fmaf(-D.Ci, x, fmaf(D.Si, y, D.Kx));
  const float nY = rintf(tY), nX = rintf(tX);
  const float margin = fminf(0.5f - fabsf(tY - nY), 0.5f - fabsf(tX - nX));
  const float bound = fmaf(fabsf(x) + fabsf(y), D.bound_per_m, D.bound_fixed);
  if (__ballot(valid && !(margin > bound))) {              // (NaN: not greater)
    Rt2DCellOfPrerotated(F, D.rot.x, D.rot.y, x, y, ix, iy);
  } else {
    *ix = min(max(static_cast<int>(nY), D.lo), D.ix_hi);
    *iy = min(max(static_cast<int>(nX), D.lo), D.iy_hi);
  }
}

constexpr int kBoundListCap = 128;           // blocks phase C sums per match; more: the per-candidate kernels
// words of the block-sum region: the sums of the match's blocks, later the summed candidates
// (index, quantised sum) the fused finish looks at
__host__ __device__ constexpr size_t BoundSumWords(int blocks) {
  return (static_cast<size_t>(blocks > 8 * (kBoundListCap + 1) ? blocks : 8 * (kBoundListCap + 1)) + 3) & ~size_t{3};
}

// ---------------------------------------------------------------------------------------------
// Phases B and C and the finish of ONE match by a whole workgroup of kThreads threads, from the
// byte sums of all its blocks in LDS (`ub`): the tail of the fused kernel's item, or the tail
// kernel's whole job (round 6).  `ax`, `ay`: the cloud rotated by the initial yaw; `rots`: the
// rotation table; `list`, `sums`: kBoundListCap / 4 kBoundListCap words; `ctl`, `red`,
// `best_sum`: the caller's static words (ctl[1] = ctl[3] = 0, best_sum = 0); `fin_smem`: where
// Rt2DFinishMatch may lay its region out (nothing of the above inside it, the cloud may be).
// ---------------------------------------------------------------------------------------------
template <int NB, int kThreads>
__device__ __forceinline__ void Rt2DBoundTail(Rt2DTileParams& P, unsigned char* fin_smem, const float* ax,
                                              const float* ay, const float2* rots, int* ub, int* list,
                                              int* sums, int* ctl, unsigned long long* red, int* best_sum,
                                              bool outside, int group, unsigned* __restrict__ host_out,
                                              int match, unsigned long long* tl, int tl_block) {
  constexpr int kWaves = kThreads / 64;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = P.n, n_pad = P.n_pad, S = P.num_scans, nl = P.nl;
  const int side = 2 * nl + 1, cands = side * side;
  const int nblk = S * NB * NB;
  float* ubw = reinterpret_cast<float*>(ub);            // (phase B rewrites the sums as bounds)
  const Rt2DFrame F = FrameOf(P);
  const int off_x = P.hl - nl, off_y = P.ht - nl;       // window start in image coordinates
  const int box_x0 = P.box_x0, box_y0 = P.box_y0, T = P.T;
  const int pchunks = n_pad >> 6;
  // window start of this lane's point of chunk c under D (image coordinates); false: no point
  // here, or one outside the predicted box (flagged: the host repeats the match elsewhere)
  const auto window_start = [&](const BoundDisc& D, int c, int* Xs, int* Ys) {
    const int i = (c << 6) + lane;
    const bool valid = i < n;
    int ix, iy;
    BoundCellOf(D, F, ax[i], ay[i], valid, &ix, &iy);
    *Xs = ix + off_x;
    *Ys = iy + off_y;
    const bool inside = static_cast<unsigned>(*Xs - box_x0) < static_cast<unsigned>(T) &&
                        static_cast<unsigned>(*Ys - box_y0) < static_cast<unsigned>(T);
    if (valid && !inside) outside = true;
    return valid && inside;
  };
  const bool verify = P.b_verify != 0;
  if (verify) {
    // (verify mode leaves every candidate's sum for the finish KERNEL, as the tile kernel does;
    // only this item writes them: zeros first, the blocks' sums behind a barrier)
    auto* qsum = AsGlobal(P.qsum);
    for (int e = tid; e < S * cands; e += kThreads) qsum[e] = 0;
  }

  // ---- phase B: weighted upper bounds, the best block ---------------------------------------
  const float kScale = ((1.f - 0.1f) - (1.f - (1.f - 0.1f))) / 32766.f;   // (kMaxCC - kMinCC) / 32766
  const float slack = Rt2DBoundSlack(n);
  const float per_m = kScale * static_cast<float>(kBoundUnit) / static_cast<float>(n);
  const float per_q = kScale * static_cast<float>(1 << kQShift) / static_cast<float>(n);
  {
    unsigned long long key = 0;
    for (int e = tid; e < nblk; e += kThreads) {
      const int s = e / (NB * NB), b = e - s * (NB * NB);
      const int j = b / NB, k = b - j * NB;
      float wmax = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int dxi = 2 * k + (q & 1), dyi = 2 * j + (q >> 1);
        if (dxi < side && dyi < side) wmax = fmaxf(wmax, TileWeight(P, s, dxi - nl, dyi - nl));
      }
      const float bound = (0.1f + per_m * static_cast<float>(ub[e]) + slack) * wmax * (1.f + 1e-5f);
      ubw[e] = bound;
      const unsigned long long mine =
          (static_cast<unsigned long long>(__float_as_uint(fmaxf(bound, 0.f))) << 32) |
          static_cast<unsigned>(0x7fffffff - e);
      key = mine > key ? mine : key;
    }
    key = WaveMaxU64(key);
    if (lane == 0) red[wave] = key;
  }
  __syncthreads();
  unsigned long long best_key = 0;
#pragma unroll
  for (int w = 0; w < kWaves; ++w) best_key = red[w] > best_key ? red[w] : best_key;
  const int best_e = 0x7fffffff - static_cast<int>(static_cast<unsigned>(best_key));
  Stamp(tl, tl_block, 3);                    // weighted bounds, the best block known

  // The quantised sums of a block's four candidates over `count` chunks from c_first on, c_step
  // apart: what the tile kernel sums for them (cells of the HBM image; outside it: 0).  Four
  // chunks at a time: their sixteen gathers leave together.
  const auto* qimage = AsGlobal(P.qimage);
  const int gw = P.gpitch >> 1, grows = P.grows;
  const auto block_sums = [&](int e, int c_first, int c_step, int count, int (&sum)[4]) {
    const int s = e / (NB * NB), b = e - s * (NB * NB);
    const int j = b / NB, k = b - j * NB;
    const BoundDisc D = MakeBoundDisc(P, rots[s]);
    sum[0] = sum[1] = sum[2] = sum[3] = 0;
#pragma unroll 1
    for (int t = 0; t < count; t += 4) {
      unsigned v[4][4];
      bool ok[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int cu = c_first + (t + u) * c_step;
        int Xs = 0, Ys = 0;
        const bool live = t + u < count && cu < pchunks && window_start(D, min(cu, pchunks - 1), &Xs, &Ys);
        const int X = Xs + 2 * k, Y = Ys + 2 * j;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int Xc = X + (q & 1), Yc = Y + (q >> 1);
          ok[u][q] = live && Xc < gw && Yc < grows;
          v[u][q] = qimage[ok[u][q] ? Yc * gw + Xc : 0];
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q) sum[q] += ok[u][q] ? static_cast<int>(v[u][q]) : 0;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) sum[q] = WaveSum(sum[q]);
  };
  const auto lower_bound_of = [&](int e, const int (&sum)[4]) {
    const int s = e / (NB * NB), b = e - s * (NB * NB);
    const int j = b / NB, k = b - j * NB;
    float lb = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int dxi = 2 * k + (q & 1), dyi = 2 * j + (q >> 1);
      if (dxi < side && dyi < side) {
        const float base = 0.1f + per_q * static_cast<float>(sum[q]);
        lb = fmaxf(lb, (base - slack) * TileWeight(P, s, dxi - nl, dyi - nl) * (1.f - 1e-5f));
      }
    }
    return lb;
  };
  const auto store_sums = [&](int e, const int (&sum)[4]) {      // (one lane)
    const int s = e / (NB * NB), b = e - s * (NB * NB);
    const int j = b / NB, k = b - j * NB;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int dxi = 2 * k + (q & 1), dyi = 2 * j + (q >> 1);
      if (dxi < side && dyi < side)
        AsGlobal(P.qsum)[static_cast<size_t>(s) * cands + dxi * side + dyi] = sum[q];
    }
  };
  {
    // the best block: its chunks dealt over the wavefronts, the sums meet in LDS
    int sum[4];
    block_sums(best_e, wave, kWaves, (pchunks - wave + kWaves - 1) / kWaves, sum);
    if (lane == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) atomicAdd(&best_sum[q], sum[q]);
    }
  }
  __syncthreads();                           // (and the zeros of qsum are behind every later store)
  Stamp(tl, tl_block, 4);                    // the best block summed
  float lb;
  {
    const int sum[4] = {best_sum[0], best_sum[1], best_sum[2], best_sum[3]};
    lb = lower_bound_of(best_e, sum);
    if (verify && tid == 0) store_sums(best_e, sum);
  }
  // ---- phase C: every other block that reaches the bound ----------------------------------------
  if (verify) {
    // (debug switch rt2d_bounds_verify: EVERY block is summed, a wavefront per block, and a block
    // whose weighted bound lies below the weighted value of one of its own candidates is
    // reported -- the invariant the pruning rests on, tests/test_gpu_r2_paths.py)
    bool violated = false;
    int summed = 0;
#pragma unroll 1
    for (int e = wave; e < nblk; e += kWaves) {
      int sum[4];
      block_sums(e, 0, 1, pchunks, sum);
      if (e != best_e) {
        if (lane == 0) store_sums(e, sum);
        ++summed;
      }
      const int s = e / (NB * NB), b = e - s * (NB * NB);
      const int j = b / NB, k = b - j * NB;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int dxi = 2 * k + (q & 1), dyi = 2 * j + (q >> 1);
        if (dxi < side && dyi < side &&
            ubw[e] < (0.1f + per_q * static_cast<float>(sum[q])) * TileWeight(P, s, dxi - nl, dyi - nl))
          violated = true;
      }
    }
    if (lane == 0 && summed) atomicAdd(&ctl[1], summed);
    if (violated && lane == 0) atomicOr(&P.misc[0], kBoundViolated);
    __syncthreads();
  } else {
    // the blocks that reach the bound, listed; their chunks in units of four dealt over the
    // wavefronts, the sums meet in LDS.  (More than the list holds -- a flat landscape: the
    // host repeats the match on the per-candidate kernels, as it does when the finish kernel
    // meets more finalists than it lists.)
    for (int e = tid; e < nblk; e += kThreads) {
      if (e != best_e && ubw[e] >= lb) {
        const int at = atomicAdd(&ctl[1], 1);
        if (at < kBoundListCap) {
          list[at] = e;
          sums[4 * at] = sums[4 * at + 1] = sums[4 * at + 2] = sums[4 * at + 3] = 0;
        }
      }
    }
    __syncthreads();
    const int listed = ctl[1];
    if (listed > kBoundListCap) {
      if (tid == 0) atomicOr(&P.misc[0], kBoundFlat);
    } else {
      const int groups = (pchunks + 3) >> 2;
#pragma unroll 1
      for (int u = wave; u < listed * groups; u += kWaves) {
        const int at = u / groups, grp = u - at * groups;
        int sum[4];
        block_sums(list[at], 4 * grp, 1, min(4, pchunks - 4 * grp), sum);
        if (lane == 0) {
#pragma unroll
          for (int q = 0; q < 4; ++q) atomicAdd(&sums[4 * at + q], sum[q]);
        }
      }
      __syncthreads();
    }
  }
  if (outside) atomicOr(&P.misc[0], kOutOfBox);
  Stamp(tl, tl_block, 5);                    // phase C: the surviving blocks summed
  const int listed = min(ctl[1], kBoundListCap);
  if (tid == 0) {
    P.bstat[0] = static_cast<unsigned>((verify ? ctl[1] : listed) + 1);   // blocks summed exactly
    P.bstat[1] = static_cast<unsigned>(nblk);                           // block bounds evaluated
  }
  if (verify) return;                      // (the finish kernel takes it from the sums in HBM)
  if (ctl[1] > kBoundListCap) {
    // nothing to finish: the flag travels with the match's words, as the finish would send them
    __syncthreads();
    if (tid < 128)
      host_out[static_cast<size_t>(match) * 128 + tid] =
          __hip_atomic_load(&P.misc[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  // ---- the finish of the match, here: the candidates of the summed blocks (the best one and
  // the listed ones) with their sums take the place of the block sums, Rt2DFinishMatch selects
  // among them (every other candidate lies below the best lower bound) and lays its own LDS
  // out over the planes and the cloud, which nobody reads any more.
  int* cand_e = ub;
  int* cand_q = ub + 4 * (kBoundListCap + 1);
  if (tid == 0) ctl[3] = 0;
  __syncthreads();                           // (the bounds in `ub` have been read by everyone)
  if (tid <= listed) {
    const int e = tid == 0 ? best_e : list[tid - 1];
    const int s = e / (NB * NB), b = e - s * (NB * NB);
    const int j = b / NB, k = b - j * NB;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int dxi = 2 * k + (q & 1), dyi = 2 * j + (q >> 1);
      if (dxi < side && dyi < side) {
        const int at = atomicAdd(&ctl[3], 1);
        cand_e[at] = s * cands + dxi * side + dyi;
        cand_q[at] = tid == 0 ? best_sum[q] : sums[4 * (tid - 1) + q];
      }
    }
  }
  __syncthreads();
  Rt2DFinishMatch<kThreads, true, false>(P, fin_smem, group, host_out, match, cand_e, cand_q, ctl[3]);
}

// ---------------------------------------------------------------------------------------------
// grid (persistent: two workgroups of 512 threads per CU), work items = (match, rotation group g
// of G) from the fused path's list, pulled through a counter.  An item bounds the blocks of its
// rotations g, g + G, ... (phase A); with G > 1 the byte sums travel through HBM (agent-scope
// stores) and the item that draws the match's LAST ticket runs phases B and C for the whole
// match -- it holds the match's planes and cloud like every other.  Large batches: G = 1.
// Dynamic LDS: planes[4][b_lh][b_lpb] | zeros[NB * b_lpb + 16] | ax[n_pad] | ay[n_pad] | (up to
//   b_tail_at: what the fused finish needs) rots[num_scans] | ub[BoundSumWords] |
//   list[kBoundListCap] | sums[kBoundListCap][4]
// ---------------------------------------------------------------------------------------------

// LOG = 1: blocks of 2 x 2 translations (four parity planes, NB <= 8 block columns in three
// dwords); LOG = 2 (round 6): blocks of 4 x 4 translations -- sixteen phase planes
// plane(Y & 3, X & 3)[Y >> 2][X >> 2] of the 4 x 4 max-pooled image (Rt2DPool4Kernel), NB <= 4
// block columns in two dwords, always split: the tail kernel (Rt2DBoundTail4Kernel) refines the
// few blocks that survive through their 2 x 2 sub-blocks.  C1: 4 rows of two dwords per (point,
// rotation) instead of 7 rows of three.
template <int NB, int LOG>
__global__ void __launch_bounds__(kBoundThreads, 4)       // (HIP: min WAVES per SIMD -- two workgroups per CU, at most 128 VGPRs)
Rt2DBoundKernel(const Rt2DTileParams* __restrict__ params, const int4* __restrict__ work,
                const int* __restrict__ work_count, int* __restrict__ next_item,
                int* __restrict__ tickets, int* __restrict__ ub_global, int group,
                unsigned* __restrict__ host_out, int split) {
  extern __shared__ __attribute__((aligned(16))) unsigned char bound_smem[];
  __shared__ Rt2DTileParams P;
  __shared__ int fetched;
  __shared__ int ctl[4];                       // [0] next rotation of this item, [1] listed blocks, [2] ticket, [3] next unit
  __shared__ unsigned long long red[kBoundThreads / 64];
  __shared__ int best_sum[4];
  constexpr int kWaves = kBoundThreads / 64;
  constexpr int kRowWords = NB > 4 ? 2 : 1;    // dwords of a block row that hold blocks
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int num_items = *work_count;
  for (int item_index = 0;; ++item_index) {
    __syncthreads();                           // the previous item's LDS is done with
    if (tid == 0) fetched = atomicAdd(next_item, 1);
    __syncthreads();
    const int item_at = fetched;
    if (item_at >= num_items) break;
    const int4 item = work[3 * item_at];
    const int match = item.x, g = item.z, G = item.w;
    CopyParams(&P, params + match, tid);
    __syncthreads();
    // (in-kernel timeline of the profiling tools, debug switch `timeline`: slots = workgroup x its
    // first four items, as in the tile kernel)
    unsigned long long* const tl = item_index < 4 ? P.timeline : nullptr;
    const int tl_block = blockIdx.x * 4 + item_index;
    Stamp(tl, tl_block, 0);
    if (tl && tid == 0) {                      // (tools: which item, and where it ran: XCC | HW_ID)
      tl[static_cast<size_t>(tl_block) * kTimelineStamps + 14] = static_cast<unsigned>(item_at);
      tl[static_cast<size_t>(tl_block) * kTimelineStamps + 15] =
          (static_cast<unsigned long long>(__builtin_amdgcn_s_getreg((31 << 11) | 20)) << 32) |
          static_cast<unsigned>(__builtin_amdgcn_s_getreg((31 << 11) | 4));
    }
    const int n = P.n, n_pad = P.n_pad, S = P.num_scans, nl = P.nl;
    static_assert(LOG == 1 || (LOG == 2 && NB <= 4), "4 x 4 blocks: at most 16 x 16 windows");
    constexpr int kPlanes = 1 << (2 * LOG), kPhase = (1 << LOG) - 1;
    const int lpb = LOG == 1 ? P.b_lpb : P.b4_lpb, lh = LOG == 1 ? P.b_lh : P.b4_lh;
    const int plane_bytes = LOG == 1 ? lh * lpb : P.b4_pstride;   // (4 x 4: plane p starts two banks behind plane p - 1)
    const int c0 = LOG == 1 ? P.b_c0 : P.b4_c0, r0 = LOG == 1 ? P.b_r0 : P.b4_r0;
    unsigned char* planes = bound_smem;
    const int lds_planes = static_cast<int>(reinterpret_cast<uintptr_t>(
        (const __attribute__((address_space(3))) unsigned char*)bound_smem));
    const int zero_at = kPlanes * plane_bytes;            // NB rows of zeros at the planes' pitch
    const int zero_bytes = (NB * lpb + 16 + 15) & ~15;
    float* ax = reinterpret_cast<float*>(bound_smem + zero_at + zero_bytes);
    float* ay = ax + n_pad;
    const int lds_ax = lds_planes + static_cast<int>(reinterpret_cast<unsigned char*>(ax) - bound_smem);
    const int lds_ay = lds_ax + 4 * n_pad;
    float2* rots = reinterpret_cast<float2*>(bound_smem + (LOG == 1 ? P.b_tail_at : P.b4_tail_at));
    int* ub = reinterpret_cast<int*>(rots + ((S + 1) & ~1));
    const int nblk = S * NB * NB;
    int* list = ub + BoundSumWords(nblk);                 // (LOG = 1 only: the fused tail's lists)
    int* sums = list + kBoundListCap;
    int* ub_match = ub_global + (LOG == 1 ? P.b_ub_at : P.b4_ub_at);   // this match's byte sums in HBM
    // (LOG = 2: the per-rotation constants of the discretisation -- a few dozen f64 operations --
    // once per item instead of once per (wavefront, rotation): behind the block sums)
    BoundDisc* discs = reinterpret_cast<BoundDisc*>(ub + ((nblk + 3) & ~3));

    // ---- staging: cloud rotated by the initial yaw, rotations, planes ---------------------------
    {
      const auto* xyz = AsGlobal(P.xyz);
      for (int i = tid; i < n_pad; i += kBoundThreads) {
        float x = 0.f, y = 0.f;
        if (i < n) RotateZ(P.init_qw, P.init_qz, xyz[3 * i], xyz[3 * i + 1], &x, &y);
        ax[i] = x;
        ay[i] = y;
      }
      const auto* rot = AsGlobal(reinterpret_cast<const float*>(P.scan_rot));
      for (int s = tid; s < S; s += kBoundThreads) {
        rots[s] = make_float2(rot[2 * s], rot[2 * s + 1]);
        if (LOG == 2) discs[s] = MakeBoundDisc(P, rots[s]);
      }
      // the planes: 8-byte pieces (b_lpb is a multiple of 8 -- and not of 16: thirty-four dwords
      // from row to row spread the rows of a wall over the banks, a pitch of 128 bytes put them
      // all on two -- b_c0 and m2_pitch are multiples of 4), eight in flight per thread.  A piece
      // that would end beyond its source row holds no column any window reads (m2_pitch leaves
      // room behind the last one): it is fetched from the row's last 8 bytes instead, rows below
      // the planes from the last row.
      typedef unsigned U2 __attribute__((ext_vector_type(2)));
      const int ppr = lpb >> 3;
      const int pieces = kPlanes * lh * ppr;
      const auto* src = (const __attribute__((address_space(1))) unsigned char*)(LOG == 1 ? P.m2 : P.m4);
      const int src_pitch = LOG == 1 ? P.m2_pitch : P.m4_pitch, src_rows = LOG == 1 ? P.m2_rows : P.m4_rows;
      const int src_plane = src_rows * src_pitch;
      const int plane_gap = plane_bytes - lh * lpb;       // (LDS bytes between the planes)
      constexpr int kInFlight = 8;
      // (piece -> (plane, row, piece of the row) by multiplication: the quotients are exact for
      // dividends below 2^16, and two integer divisions per piece were a tenth of the kernel's
      // vector instructions)
      const unsigned ppr_magic = 0xffffffffu / static_cast<unsigned>(ppr) + 1u;
      const unsigned lh_magic = 0xffffffffu / static_cast<unsigned>(lh) + 1u;
      for (int p0 = tid; p0 < pieces; p0 += kInFlight * kBoundThreads) {
        U2 v[kInFlight];
        int dst_gap[kInFlight];
#pragma unroll
        for (int q = 0; q < kInFlight; ++q) {
          const int p = min(p0 + q * kBoundThreads, pieces - 1);
          const int pr = static_cast<int>(__umulhi(static_cast<unsigned>(p), ppr_magic)), piece = p - pr * ppr;
          const int plane = static_cast<int>(__umulhi(static_cast<unsigned>(pr), lh_magic)), row = pr - plane * lh;
          const int col = min(c0 + (piece << 3), src_pitch - 8);
          const int srow = min(r0 + row, src_rows - 1);
          v[q] = *reinterpret_cast<const __attribute__((address_space(1))) U2*>(
              src + plane * src_plane + srow * src_pitch + col);
          if (LOG == 2) dst_gap[q] = plane * plane_gap;
        }
#pragma unroll
        for (int q = 0; q < kInFlight; ++q) {
          const int p = p0 + q * kBoundThreads;
          if (p < pieces) *reinterpret_cast<U2*>(planes + (p << 3) + (LOG == 2 ? dst_gap[q] : 0)) = v[q];
        }
      }
      for (int w = tid; w < (zero_bytes >> 2); w += kBoundThreads)
        reinterpret_cast<uint32_t*>(planes + zero_at)[w] = 0;
      for (int e = tid; e < nblk; e += kBoundThreads) ub[e] = 0;      // (the slices of a rotation meet by atomics)
      if (tid < 4) { ctl[tid] = tid == 0 ? kWaves : 0; best_sum[tid] = 0; }
    }
    __syncthreads();
    Stamp(tl, tl_block, 1);                    // staged: cloud, rotations, planes

    const Rt2DFrame F = FrameOf(P);
    const int off_x = P.hl - nl, off_y = P.ht - nl;       // window start in image coordinates
    const int box_x0 = P.box_x0, box_y0 = P.box_y0, T = P.T;
    const int pchunks = n_pad >> 6;
    bool outside = false;

    // ---- phase A: the byte sums of every block of this item's rotations, a wavefront per
    // rotation.  A lane adds the dwords of its points' block rows as they are -- kBoundFlush
    // values of at most kBoundMax fit a byte -- and spreads them into 16-bit fields (bytes 0 / 2
    // and 1 / 3 of a dword: two registers) every kBoundFlush chunks; a lane's fields hold at most
    // 32 chunks x kBoundMax, the wavefront's totals 64 times that: < 2^16 (kFusedMaxPoints).
    // An item with fewer rotations than wavefronts (a single match spread over the chip) slices
    // every rotation's chunks over several of them.
    const int my_rotations = (S - g + G - 1) / G;
    // (C1's 27 rotations over 8 wavefronts leave the last round three of eight busy; halves of
    // rotations -- 54 units -- measured: 86.5 against 81.7 us per 1024 matches, the second
    // workgroup of the CU fills the gap better than twice the reductions do)
    const int slices = max(1, min(kWaves / my_rotations, pchunks));
    const int slice_chunks = (pchunks + slices - 1) / slices;
    int rotations_done = 0;
#pragma unroll 1
    for (int unit = wave; unit < my_rotations * slices;
         unit = __builtin_amdgcn_readfirstlane(lane == 0 ? atomicAdd(&ctl[0], 1) : 0)) {
      const int ri = unit / slices, slice = unit - ri * slices;
      const int chunk_first = slice * slice_chunks, chunk_end = min(pchunks, chunk_first + slice_chunks);
      const int s = g + ri * G;
      const BoundDisc D = LOG == 2 ? discs[s] : MakeBoundDisc(P, rots[s]);
      uint32_t even[NB][kRowWords], odd[NB][kRowWords];   // 16-bit fields: blocks (4 w, 4 w + 2) / (4 w + 1, 4 w + 3)
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int w = 0; w < kRowWords; ++w) even[j][w] = odd[j][w] = 0;
      // The address of a point's first block row (the zero rows for no point).  Software pipeline:
      // the rows of chunk c are in flight while the cells of chunk c + 1 are worked out, from
      // coordinates that were requested a chunk earlier -- nothing between the row requests and
      // their wait reads LDS.
      const auto row_address = [&](float x, float y, int c) {
        const int i = (c << 6) + lane;
        const bool valid = i < n;
        int ix, iy;
        BoundCellOf(D, F, x, y, valid, &ix, &iy);
        const int Xs = ix + off_x, Ys = iy + off_y;
        const bool inside = static_cast<unsigned>(Xs - box_x0) < static_cast<unsigned>(T) &&
                            static_cast<unsigned>(Ys - box_y0) < static_cast<unsigned>(T);
        if (valid && !inside) outside = true;
        const int plane = ((Ys & kPhase) << LOG) | (Xs & kPhase);
        // (24-bit multiplies: full rate, where v_mul_lo_u32 takes four issue slots -- a plane
        // index and a row times strides far below 2^24)
        return valid && inside ? __mul24(plane, plane_bytes) + __mul24((Ys >> LOG) - r0, lpb) + ((Xs >> LOG) - c0)
                               : zero_at;
      };
      const int last = chunk_end - 1;
      int at = row_address(ax[(chunk_first << 6) + lane], ay[(chunk_first << 6) + lane], chunk_first);
      float xn = ax[(min(chunk_first + 1, last) << 6) + lane], yn = ay[(min(chunk_first + 1, last) << 6) + lane];
#pragma unroll 1
      for (int c0 = chunk_first; c0 < chunk_end; c0 += kBoundFlush) {
        uint32_t packed[NB][kRowWords];
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
          for (int w = 0; w < kRowWords; ++w) packed[j][w] = 0;
        const int c1 = min(c0 + kBoundFlush, chunk_end);
#pragma unroll 1
        for (int c = c0; c < c1; ++c) {
          // (a ds_read_b64 that is not 8-byte aligned takes ~40 cycles of the CU's LDS pipeline
          // instead of ~7: tools/probes/lds_unaligned.hip.  The row is read as aligned dwords and
          // shifted into place: v_alignbyte_b32, the byte shift in a register)
          const int a4 = at & ~3;
          const unsigned sh = static_cast<unsigned>(at) & 3u;
          // Every row is requested before the first is used: reads and the one wait are asm (the
          // compiler's own schedule reads two rows, drains the queue, adds, reads the next two:
          // four LDS round trips per chunk); the empty asm behind the wait ties every register to
          // it, so no use can move in front of it.
          uint2v w01[NB];
          uint32_t w2[NB];
#pragma unroll
          for (int j = 0; j < NB; ++j) {
            const int row_at = lds_planes + a4 + j * lpb;
            asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(w01[j]) : "v"(row_at));
            if (kRowWords > 1) asm volatile("ds_read_b32 %0, %1 offset:8" : "=v"(w2[j]) : "v"(row_at));
          }
          // the coordinates of chunk c + 2 requested, the cells of chunk c + 1 worked out (the last
          // chunks repeat the last one: their addresses are not used)
          const float x1 = xn, y1 = yn;
          const int c2 = min(c + 2, last);
          float x2, y2;
          asm volatile("ds_read_b32 %0, %1" : "=v"(x2) : "v"(lds_ax + (((c2 << 6) + lane) << 2)));
          asm volatile("ds_read_b32 %0, %1" : "=v"(y2) : "v"(lds_ay + (((c2 << 6) + lane) << 2)));
          at = row_address(x1, y1, min(c + 1, last));
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
          for (int j = 0; j < NB; ++j) {
            asm volatile("" : "+v"(w01[j]));
            if (kRowWords > 1) asm volatile("" : "+v"(w2[j]));
          }
          asm volatile("" : "+v"(x2), "+v"(y2));
          xn = x2;
          yn = y2;
#pragma unroll
          for (int j = 0; j < NB; ++j) {
            packed[j][0] += __builtin_amdgcn_alignbyte(w01[j].y, w01[j].x, sh);
            if (kRowWords > 1) packed[j][kRowWords - 1] += __builtin_amdgcn_alignbyte(w2[j], w01[j].y, sh);
          }
        }
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
          for (int w = 0; w < kRowWords; ++w) {
            even[j][w] += packed[j][w] & 0x00ff00ffu;
            odd[j][w] += (packed[j][w] >> 8) & 0x00ff00ffu;
          }
      }
      // (both fields of a register are summed across the lanes at once: no carry between them.
      // Lane l collects the total of register l -- one select per register -- and every lane then
      // adds its two blocks' sums into the match's array: two LDS operations per rotation instead
      // of a predicated one per block)
      uint32_t mine = 0;
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int w = 0; w < kRowWords; ++w) {
          const uint32_t te = static_cast<uint32_t>(WaveSum(static_cast<int>(even[j][w])));
          const uint32_t to = static_cast<uint32_t>(WaveSum(static_cast<int>(odd[j][w])));
          if (j == 0 && w == 0 && wave == 0 && rotations_done < 8) Stamp(tl, tl_block, 6 + rotations_done++);   // (wave 0: its rotations)
          mine = lane == 2 * (j * kRowWords + w) ? te : mine;
          mine = lane == 2 * (j * kRowWords + w) + 1 ? to : mine;
        }
      if (lane < 2 * NB * kRowWords) {
        const int reg = lane >> 1, j = reg / kRowWords, w = reg - j * kRowWords, par = lane & 1;
        int* row = ub + (s * NB + j) * NB;
        if (4 * w + par < NB) atomicAdd(&row[4 * w + par], static_cast<int>(mine & 0xffffu));
        if (4 * w + 2 + par < NB) atomicAdd(&row[4 * w + 2 + par], static_cast<int>(mine >> 16));
      }
    }
    __syncthreads();
    Stamp(tl, tl_block, 2);                    // phase A: the byte sums of this item's blocks
    if (outside) atomicOr(&P.misc[0], kOutOfBox);
    if (split || LOG == 2) {
      // (round 6) the sums of this item's rotations to HBM: the tail kernel, behind this one on
      // the stream, takes the match from there -- this workgroup's LDS and its place on the CU go
      // to the next item's phase A instead of ~30 us of dependent round trips
      for (int e = tid; e < my_rotations * NB * NB; e += kBoundThreads) {
        const int ri = e / (NB * NB), b = e - ri * (NB * NB);
        const int at = (g + ri * G) * (NB * NB) + b;
        ub_match[at] = ub[at];
      }
      continue;
    }
    if (G > 1) {
      // the sums of this item's rotations to HBM, written through (other workgroups -- other
      // XCDs -- read them); then the match's ticket: the last item carries on
      for (int e = tid; e < my_rotations * NB * NB; e += kBoundThreads) {
        const int ri = e / (NB * NB), b = e - ri * (NB * NB);
        const int at = (g + ri * G) * (NB * NB) + b;
        __hip_atomic_store(&ub_match[at], ub[at], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __syncthreads();                         // (vmcnt(0): every store of the workgroup has been acknowledged)
      if (tid == 0)
        ctl[2] = __hip_atomic_fetch_add(&tickets[match], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      if (ctl[2] != G - 1) continue;           // (uniform)
      for (int e = tid; e < nblk; e += kBoundThreads)
        ub[e] = __hip_atomic_load(&ub_match[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
    }
    if constexpr (LOG == 1)
      Rt2DBoundTail<NB, kBoundThreads>(P, bound_smem, ax, ay, rots, ub, list, sums, ctl, red, best_sum, outside,
                                       group, host_out, match, tl, tl_block);
  }
}

// ---------------------------------------------------------------------------------------------
// grid (matches), 512 threads: the tail of a match whose block sums the bound kernel (split = 1)
// left in HBM.  ~33 KB of LDS: four or five workgroups per CU, every match of a part in flight at
// once -- the tail is a chain of a dozen dependent round trips (weighted bounds, the best block,
// the listed blocks, the finalists' gathers and f32 chains), latency that only other matches hide.
// Dynamic LDS: [Rt2DFinishMatch's region, over] ax[n_pad] | ay[n_pad] | (at BoundTailRegion)
//   rots[num_scans] | ub[BoundSumWords] | list[kBoundListCap] | sums[kBoundListCap][4]
// ---------------------------------------------------------------------------------------------
constexpr int kBoundTailThreads = 512;
constexpr int kBoundTailGroup = 2;           // finalists per round of f32 chains (LDS rows)
__host__ __device__ constexpr size_t BoundTailRegion(int n_pad, int num_scans) {
  const size_t fin = 4 * static_cast<size_t>(n_pad) + static_cast<size_t>(kBoundTailGroup) * 4 * (static_cast<size_t>(n_pad) + 4) +
                     4 * ((static_cast<size_t>(num_scans) + 3) & ~size_t{3}) + 12 * static_cast<size_t>(kStage1Cap);
  const size_t cloud = 8 * static_cast<size_t>(n_pad);
  return ((fin > cloud ? fin : cloud) + 15) & ~size_t{15};
}
__host__ __device__ constexpr size_t BoundTailLds(int n_pad, int num_scans, int nb) {
  return BoundTailRegion(n_pad, num_scans) + 8 * ((static_cast<size_t>(num_scans) + 1) & ~size_t{1}) +
         4 * BoundSumWords(num_scans * nb * nb) + 20 * size_t{kBoundListCap};
}

template <int NB>
__global__ void __launch_bounds__(kBoundTailThreads)
Rt2DBoundTailKernel(const Rt2DTileParams* __restrict__ params, const int* __restrict__ ub_global,
                    unsigned* __restrict__ host_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tail_smem[];
  __shared__ Rt2DTileParams P;
  __shared__ int ctl[4];
  __shared__ unsigned long long red[kBoundTailThreads / 64];
  __shared__ int best_sum[4];
  const int tid = threadIdx.x;
  const int match = blockIdx.x;
  CopyParams(&P, params + match, tid);
  __syncthreads();
  unsigned long long* const tl = P.timeline;
  const int tl_block = P.timeline_finish_base + match;
  Stamp(tl, tl_block, 0);
  const int n = P.n, n_pad = P.n_pad, S = P.num_scans;
  float* ax = reinterpret_cast<float*>(tail_smem);
  float* ay = ax + n_pad;
  float2* rots = reinterpret_cast<float2*>(tail_smem + BoundTailRegion(n_pad, S));
  int* ub = reinterpret_cast<int*>(rots + ((S + 1) & ~1));
  const int nblk = S * NB * NB;
  int* list = ub + BoundSumWords(nblk);
  int* sums = list + kBoundListCap;
  {
    const auto* xyz = AsGlobal(P.xyz);
    for (int i = tid; i < n_pad; i += kBoundTailThreads) {
      float x = 0.f, y = 0.f;
      if (i < n) RotateZ(P.init_qw, P.init_qz, xyz[3 * i], xyz[3 * i + 1], &x, &y);
      ax[i] = x;
      ay[i] = y;
    }
    const auto* rot = AsGlobal(reinterpret_cast<const float*>(P.scan_rot));
    for (int s = tid; s < S; s += kBoundTailThreads) rots[s] = make_float2(rot[2 * s], rot[2 * s + 1]);
    const auto* ubm = AsGlobal(ub_global + P.b_ub_at);
    for (int e = tid; e < nblk; e += kBoundTailThreads) ub[e] = ubm[e];
    if (tid < 4) { ctl[tid] = 0; best_sum[tid] = 0; }
  }
  __syncthreads();
  Stamp(tl, tl_block, 7);                      // staged: cloud, rotations, block sums
  Rt2DBoundTail<NB, kBoundTailThreads>(P, tail_smem, ax, ay, rots, ub, list, sums, ctl, red, best_sum,
                                       /*outside=*/false, kBoundTailGroup, host_out, match, tl, tl_block);
}

// ---------------------------------------------------------------------------------------------
// grid (matches), kBoundTail4Threads threads: the tail of a match behind Rt2DBoundKernel<NB4, 2>
// (round 6).  From the byte sums of the match's 4 x 4 blocks:
//   B  weighted upper bounds; the best block's SIXTEEN candidates summed: the best weighted lower
//      bound among them;
//   C  every other block whose bound reaches it (C1: eight or nine of 432) has its sixteen
//      candidates summed the same way, a wavefront per block;
//   the finish (Rt2DFinishMatch) over the candidates of the summed blocks.
// The candidates are summed out of LDS: the match's box of the BYTE image q8 = u >> 7
// (Rt2DQuantKernel writes it next to the 10-bit one), the four candidates of a window row one
// shifted dword, added as packed 16-bit fields.  A wave-wide gather from HBM / L2 with 64 distinct
// lines costs 100 - 130 cycles of the CU's address path (profiles/r03_gather_ceiling.txt), and the
// first version of this kernel -- 2 x 2 sub-block bounds from the planes in HBM, then the
// candidates' cells -- spent 480 - 580 of them per match: 140 us per 1024 matches against the
// bound kernel's 80.  The bytes are four times coarser than the 10-bit cells: a few more
// candidates reach the finish's exact stages (its bounds carry the width of the quantisation).
// Why the result cannot change: as in the header of this file -- a block is dropped only if its
// bound (>= every member's weighted f32-chain score) lies strictly below the lower bound of a
// candidate that IS summed.  Debug switch rt2d_bounds_verify: EVERY block's sixteen candidates are
// summed and checked against the block's bound.
// Dynamic LDS: [Rt2DFinishMatch's region, over] box[b8_lh][b8_lp] | zeros[4 b8_lp + 16] |
//   ax[n_pad] | ay[n_pad] | (at Tail4Region) discs[num_scans] | ub[nblk] |
//   best16[16] | list4[kList4Cap] | cand_e[kCand4Cap] | cand_q[same]
// ---------------------------------------------------------------------------------------------
#ifndef CMX_RT2D_TAIL4_THREADS
#define CMX_RT2D_TAIL4_THREADS 512
#endif
constexpr int kBoundTail4Threads = CMX_RT2D_TAIL4_THREADS;
constexpr int kList4Cap = 96;                // 4 x 4 blocks summed per match besides the best; more: the per-candidate kernels
                                             // (a wavefront per block: a match with hundreds of them holds its
                                             // whole launch up for longer than its repeat takes)
constexpr int kCand4Cap = 512;               // candidates handed to the finish; more (a landscape of ties): the same
__host__ __device__ constexpr size_t Tail4Region(int lp, int lh, int n_pad, int num_scans) {
  const size_t fin = BoundTailRegion(n_pad, num_scans);
  const size_t own = static_cast<size_t>(lh) * lp + ((4 * static_cast<size_t>(lp) + 16 + 15) & ~size_t{15}) +
                     8 * static_cast<size_t>(n_pad);
  return ((fin > own ? fin : own) + 15) & ~size_t{15};
}
__host__ __device__ constexpr size_t BoundTail4Lds(int lp, int lh, int n_pad, int num_scans, int nb4) {
  return Tail4Region(lp, lh, n_pad, num_scans) + sizeof(BoundDisc) * static_cast<size_t>(num_scans) +
         4 * ((static_cast<size_t>(num_scans) * nb4 * nb4 + 3) & ~size_t{3}) +
         4 * static_cast<size_t>(16 + kList4Cap + 2 * kCand4Cap) + 64;
}

template <int NB4>
__global__ void __launch_bounds__(kBoundTail4Threads, 4)
Rt2DBoundTail4Kernel(const Rt2DTileParams* __restrict__ params, const int* __restrict__ ub_global,
                     unsigned* __restrict__ host_out) {
  typedef unsigned U2 __attribute__((ext_vector_type(2)));
  extern __shared__ __attribute__((aligned(16))) unsigned char tail_smem[];
  __shared__ Rt2DTileParams P;
  __shared__ int ctl[4];                       // [1] listed blocks, [3] candidates for the finish
  __shared__ unsigned long long red[kBoundTail4Threads / 64];
  constexpr int kThreads = kBoundTail4Threads, kWaves = kThreads / 64;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int match = blockIdx.x;
  CopyParams(&P, params + match, tid);
  __syncthreads();
  unsigned long long* const tl = P.timeline;
  const int tl_block = P.timeline_finish_base + match;
  Stamp(tl, tl_block, 0);
  const int n = P.n, n_pad = P.n_pad, S = P.num_scans, nl = P.nl;
  const int side = 2 * nl + 1, cands = side * side;
  const int nblk = S * NB4 * NB4;
  const int lp = P.b8_lp, lh = P.b8_lh;
  const int zero_at = lh * lp;                 // four rows of zeros at the box's pitch: what a lane without a point reads
  const int zero_bytes = (4 * lp + 16 + 15) & ~15;
  float* ax = reinterpret_cast<float*>(tail_smem + zero_at + zero_bytes);
  float* ay = ax + n_pad;
  BoundDisc* discs = reinterpret_cast<BoundDisc*>(tail_smem + Tail4Region(lp, lh, n_pad, S));
  int* ub = reinterpret_cast<int*>(discs + S);
  float* ubw = reinterpret_cast<float*>(ub);
  int* raw = ub + ((nblk + 3) & ~3);           // the best block's sixteen sums
  int* list4 = raw + 16;
  int* cand_e = list4 + kList4Cap;             // the candidates handed to the finish: index, byte sum
  int* cand_q = cand_e + kCand4Cap;
  {
    const auto* xyz = AsGlobal(P.xyz);
    for (int i = tid; i < n_pad; i += kThreads) {
      float x = 0.f, y = 0.f;
      if (i < n) RotateZ(P.init_qw, P.init_qz, xyz[3 * i], xyz[3 * i + 1], &x, &y);
      ax[i] = x;
      ay[i] = y;
    }
    const auto* rot = AsGlobal(reinterpret_cast<const float*>(P.scan_rot));
    for (int s = tid; s < S; s += kThreads) discs[s] = MakeBoundDisc(P, make_float2(rot[2 * s], rot[2 * s + 1]));
    const auto* ubm = AsGlobal(ub_global + P.b4_ub_at);
    for (int e = tid; e < nblk; e += kThreads) ub[e] = ubm[e];
    if (tid < 4) ctl[tid] = 0;
    if (tid < 16) raw[tid] = 0;
    // the box of the byte image: 8-byte pieces (box_x0 and the image's pitch are multiples of 8),
    // eight in flight per thread; beyond the image: zeros
    const int ppr = lp >> 3, pieces = lh * ppr;
    const int gw8 = P.gpitch >> 1;
    const auto* src = (const __attribute__((address_space(1))) unsigned char*)P.q8;
    const unsigned ppr_magic = 0xffffffffu / static_cast<unsigned>(ppr) + 1u;   // (exact below 2^16 pieces)
    constexpr int kInFlight = 8;
    for (int p0 = tid; p0 < pieces; p0 += kInFlight * kThreads) {
      U2 v[kInFlight];
#pragma unroll
      for (int q = 0; q < kInFlight; ++q) {
        const int p = min(p0 + q * kThreads, pieces - 1);
        const int row = static_cast<int>(__umulhi(static_cast<unsigned>(p), ppr_magic)), piece = p - row * ppr;
        const int X = P.box_x0 + (piece << 3), Y = P.box_y0 + row;
        const bool in = X < gw8 && Y < P.grows;
        v[q] = *reinterpret_cast<const __attribute__((address_space(1))) U2*>(src + (in ? Y * gw8 + X : 0));
        if (!in) v[q] = U2{0u, 0u};
      }
#pragma unroll
      for (int q = 0; q < kInFlight; ++q) {
        const int p = p0 + q * kThreads;
        if (p < pieces) reinterpret_cast<U2*>(tail_smem)[p] = v[q];
      }
    }
    for (int w = tid; w < (zero_bytes >> 2); w += kThreads) reinterpret_cast<uint32_t*>(tail_smem + zero_at)[w] = 0;
  }
  __syncthreads();
  Stamp(tl, tl_block, 7);                      // staged: box, cloud, discretisation constants, block sums
  const Rt2DFrame F = FrameOf(P);
  const int off_x = P.hl - nl, off_y = P.ht - nl;       // window start in image coordinates
  const int box_x0 = P.box_x0, box_y0 = P.box_y0, T = P.T;
  const int pchunks = n_pad >> 6;
  bool outside = false;
  const bool verify = P.b_verify != 0;
  const float kScale = ((1.f - 0.1f) - (1.f - (1.f - 0.1f))) / 32766.f;   // (kMaxCC - kMinCC) / 32766
  const float slack = Rt2DBoundSlack(n);
  const float per_m = kScale * static_cast<float>(kBoundUnit) / static_cast<float>(n);
  const float per_q = kScale * static_cast<float>(1 << kQ8Shift) / static_cast<float>(n);
  // (the weight falls with the distance from the window's centre -- sqrt, the products and the
  // exponential are monotone, and what a last-bit wobble of __expf could do lies far inside the
  // 1e-5 the bounds carry: the block's largest weight is that of its candidate nearest the centre)
  const auto weight_max = [&](int s, int x0, int y0) {
    if (x0 >= side || y0 >= side) return 0.f;      // (no candidate: a smaller window in a launch for a larger one)
    const int dxi = min(max(nl, x0), min(x0 + 4, side) - 1), dyi = min(max(nl, y0), min(y0 + 4, side) - 1);
    return TileWeight(P, s, dxi - nl, dyi - nl);
  };

  // ---- B: weighted upper bounds of the blocks, the best one ------------------------------------
  {
    unsigned long long key = 0;
    for (int e = tid; e < nblk; e += kThreads) {
      const int s = e / (NB4 * NB4), b = e - s * (NB4 * NB4);
      const int j = b / NB4, k = b - j * NB4;
      const float bound = (0.1f + per_m * static_cast<float>(ub[e]) + slack) * weight_max(s, 4 * k, 4 * j) * (1.f + 1e-5f);
      ubw[e] = bound;
      const unsigned long long mine =
          (static_cast<unsigned long long>(__float_as_uint(fmaxf(bound, 0.f))) << 32) |
          static_cast<unsigned>(0x7fffffff - e);
      key = mine > key ? mine : key;
    }
    key = WaveMaxU64(key);
    if (lane == 0) red[wave] = key;
  }
  __syncthreads();
  unsigned long long best_key = 0;
#pragma unroll
  for (int w = 0; w < kWaves; ++w) best_key = red[w] > best_key ? red[w] : best_key;
  const int best_e = 0x7fffffff - static_cast<int>(static_cast<unsigned>(best_key));
  Stamp(tl, tl_block, 3);                      // weighted bounds, the best block known

  // The byte sums of the sixteen candidates of block e over the chunks c_first, c_first +
  // c_step, ...: lane q (< 16) returns the wavefront's sum of candidate q = 4 dy + dx.  Four chunks
  // at a time; the four candidates of a window row are four neighbouring bytes of the box: two
  // aligned dwords shifted into one (v_alignbyte_b32), added as two registers of 16-bit fields
  // (a lane's thirty-two chunks of at most 255 stay far below 2^16).
  const uint32_t* box32 = reinterpret_cast<const uint32_t*>(tail_smem);
  const auto block_sums16 = [&](int e, int c_first, int c_step) -> int {
    const int s = e / (NB4 * NB4), b = e - s * (NB4 * NB4);
    const int j = b / NB4, k = b - j * NB4;
    const BoundDisc D = discs[s];
    const int rel = (4 * j - box_y0) * lp + 4 * k - box_x0;
    unsigned acc[4][2];
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r][0] = acc[r][1] = 0;
#pragma unroll 1
    for (int c0 = c_first; c0 < pchunks; c0 += 4 * c_step) {
      int at[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = c0 + u * c_step;
        const int i = (min(c, pchunks - 1) << 6) + lane;
        const bool valid = c < pchunks && i < n;
        int ix, iy;
        BoundCellOf(D, F, ax[i], ay[i], valid, &ix, &iy);
        const int Xs = ix + off_x, Ys = iy + off_y;
        const bool inside = static_cast<unsigned>(Xs - box_x0) < static_cast<unsigned>(T) &&
                            static_cast<unsigned>(Ys - box_y0) < static_cast<unsigned>(T);
        if (valid && !inside) outside = true;
        at[u] = valid && inside ? __mul24(Ys, lp) + Xs + rel : zero_at;
      }
      uint32_t w0[4][4], w1[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int word = (at[u] + r * lp) >> 2;
          w0[u][r] = box32[word];
          w1[u][r] = box32[word + 1];
        }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          // (the pitch is a multiple of 4: the byte shift is that of the block's first row)
          const uint32_t w = __builtin_amdgcn_alignbyte(w1[u][r], w0[u][r], static_cast<unsigned>(at[u]) & 3u);
          acc[r][0] += w & 0x00ff00ffu;            // candidates dx = 0 and 2
          acc[r][1] += (w >> 8) & 0x00ff00ffu;     // candidates dx = 1 and 3
        }
    }
    int mine = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        const unsigned word = acc[r][h & 1];
        const int total = WaveSum(static_cast<int>((h >> 1) ? word >> 16 : word & 0xffffu));
        mine = lane == 4 * r + h ? total : mine;
      }
    return mine;
  };
  // candidate q of block e: its index in the search space, -1: outside the window
  const auto candidate_of = [&](int e, int q) {
    const int s = e / (NB4 * NB4), b = e - s * (NB4 * NB4);
    const int j = b / NB4, k = b - j * NB4;
    const int dxi = 4 * k + (q & 3), dyi = 4 * j + (q >> 2);
    return dxi < side && dyi < side ? s * cands + dxi * side + dyi : -1;
  };
  const auto weight_of = [&](int candidate) {
    const int s = candidate / cands, c = candidate - s * cands;
    const int dxi = c / side, dyi = c - dxi * side;
    return TileWeight(P, s, dxi - nl, dyi - nl);
  };

  if (verify) {
    // EVERY block: its sixteen candidates against its bound; the sums go to qsum for the finish
    // KERNEL (every candidate of the search space lies in exactly one block)
    auto* qsum = AsGlobal(P.qsum);
    bool violated = false;
#pragma unroll 1
    for (int e = wave; e < nblk; e += kWaves) {
      const int sum = block_sums16(e, 0, 1);
      const int candidate = lane < 16 ? candidate_of(e, lane) : -1;
      if (candidate >= 0) {
        if (ubw[e] < (0.1f + per_q * static_cast<float>(sum)) * weight_of(candidate)) violated = true;
        qsum[candidate] = sum;
      }
    }
    if (violated) atomicOr(&P.misc[0], kBoundViolated);
    if (outside) atomicOr(&P.misc[0], kOutOfBox);
    if (tid == 0) {
      P.bstat[0] = static_cast<unsigned>(S * cands);     // candidates summed: all
      P.bstat[1] = static_cast<unsigned>(nblk);          // bounds evaluated
    }
    return;
  }

  // ---- the best block's sixteen candidates: its chunks dealt over the wavefronts -----------------
  {
    const int sum = block_sums16(best_e, wave, kWaves);
    if (lane < 16) atomicAdd(&raw[lane], sum);
  }
  __syncthreads();
  Stamp(tl, tl_block, 4);                      // the best block summed
  float lb = 0.f;
  {
    // (sixteen lanes of every wavefront: the same sixteen values, no barrier for a broadcast)
    const int candidate = lane < 16 ? candidate_of(best_e, lane) : -1;
    if (candidate >= 0)
      lb = (0.1f + per_q * static_cast<float>(raw[lane]) - slack) * weight_of(candidate) * (1.f - 1e-5f);
    lb = __uint_as_float(static_cast<unsigned>(WaveMaxDpp(static_cast<int>(__float_as_uint(fmaxf(lb, 0.f))))));
  }
  // ---- C: the other blocks that reach the bound, a wavefront per block.  What a block's
  // candidates are worth is known as soon as they are summed: the finish's own bounds (byte
  // quantisation: `width`) give every candidate an interval, the best lower end seen so far is
  // kept in LDS (it only rises), and a candidate is handed to the finish only if its upper end
  // reaches it -- the finish would drop the others at its first step, against a bound at least as
  // high.  So what is listed for the finish are the candidates near the top, however many blocks
  // had to be looked at: a flat landscape costs time, not a repeat on the per-candidate kernels
  // (that is left to landscapes of TIES: more near-best candidates than the list holds).
  const float width = kScale * static_cast<float>((1 << kQ8Shift) - 1);
  for (int e = tid; e < nblk; e += kThreads) {
    if (e != best_e && ubw[e] >= lb) {
      const int at = atomicAdd(&ctl[1], 1);
      if (at < kList4Cap) list4[at] = e;
    }
  }
  if (tid == 0) ctl[2] = static_cast<int>(__float_as_uint(lb));
  if (tid < 16) {                              // (the best block's candidates: all of them)
    const int candidate = candidate_of(best_e, tid);
    if (candidate >= 0) {
      const int at = atomicAdd(&ctl[3], 1);
      cand_e[at] = candidate;
      cand_q[at] = raw[tid];
      atomicAdd(&ctl[0], 1);
    }
  }
  __syncthreads();
  bool flat = ctl[1] > kList4Cap;
  const int listed4 = min(ctl[1], kList4Cap);
  int summed = 0;
  if (!flat) {
#pragma unroll 1
    for (int at = wave; at < listed4; at += kWaves) {
      const int e = list4[at];
      const int sum = block_sums16(e, 0, 1);
      const int candidate = lane < 16 ? candidate_of(e, lane) : -1;
      float lo = 0.f, hi = -1.f;
      if (candidate >= 0) {
        const float base = 0.1f + per_q * static_cast<float>(sum);
        const float w = weight_of(candidate);
        lo = (base - slack) * w * (1.f - 1e-5f);
        hi = (base + width + slack) * w * (1.f + 1e-5f);
      }
      const float seen = __uint_as_float(static_cast<unsigned>(__hip_atomic_load(&ctl[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)));
      if (hi >= seen) {
        const int slot = atomicAdd(&ctl[3], 1);
        if (slot < kCand4Cap) { cand_e[slot] = candidate; cand_q[slot] = sum; }
      }
      const int best_lo = WaveMaxDpp(static_cast<int>(__float_as_uint(fmaxf(lo, 0.f))));
      if (lane == 0) atomicMax(&ctl[2], best_lo);
      summed += __popcll(__ballot(candidate >= 0));
    }
  }
  if (outside) atomicOr(&P.misc[0], kOutOfBox);
  if (lane == 0 && summed) atomicAdd(&ctl[0], summed);
  __syncthreads();
  Stamp(tl, tl_block, 5);                      // the surviving blocks summed
  flat = flat || ctl[3] > kCand4Cap;
  if (tid == 0) {
    P.bstat[0] = static_cast<unsigned>(ctl[0]);                                    // candidates summed
    P.bstat[1] = static_cast<unsigned>(nblk);                                      // bounds evaluated
    if (flat) atomicOr(&P.misc[0], kBoundFlat);
  }
  if (flat) {
    // nothing to finish: the flag travels with the match's words, as the finish would send them
    __syncthreads();
    if (tid < 128)
      host_out[static_cast<size_t>(match) * 128 + tid] =
          __hip_atomic_load(&P.misc[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  Rt2DFinishMatch<kThreads, true, false, kQ8Shift>(P, tail_smem, kBoundTailGroup, host_out, match, cand_e, cand_q,
                                                    ctl[3]);
}

#endif  // CMX_RT_2D_BOUNDS_H_
