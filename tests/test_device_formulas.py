"""CPU checks (numpy float32 / float64 emulation, same IEEE operations) of two device-side
shortcuts whose exactness the GPU parity tests rely on but cannot enumerate:

* `SumUpperBound` (fast_2d.hip): the integer recovered from a node's f32 score must never be
  below the integer sum the score was computed from, or the early exit of `ExpandWaveKernel`
  could drop a child the reference keeps.
* `FastCellIndex` (rt_3d.hip) / `CellIndexF64` (cmx_device.h): when the shortcut path is taken
  its result must equal lround of the IEEE quotient the reference computes.
"""
import numpy as np
import pytest


def _to_score(total, n, min_s, scale):
    # ToScore: min_s + (float(sum) / float(n)) * score_scale, all f32
    return (np.float32(min_s) + (total.astype(np.float32) / np.float32(n)) * np.float32(scale)) \
        .astype(np.float32)


@pytest.mark.parametrize("n", [1, 7, 200, 1000, 5000, 65536])
def test_sum_upper_bound_is_conservative(n):
    min_s = np.float32(1) - (np.float32(1) - np.float32(0.1))          # 1 - max_cc
    max_s = np.float32(1) - (np.float32(1) - (np.float32(1) - np.float32(0.1)))
    scale = (max_s - min_s) / np.float32(255)
    hi = 255 * n
    sums = np.unique(np.concatenate([np.arange(0, min(hi, 70000) + 1),
                                     np.linspace(0, hi, 200001).astype(np.int64),
                                     hi - np.arange(0, min(hi, 70000) + 1)]))
    score = _to_score(sums, n, min_s, scale)
    s = ((score - min_s) / scale * np.float32(n)).astype(np.float32)
    ub = np.ceil((s * (np.float32(1) + np.float32(1e-5))).astype(np.float32)) + np.float32(2)
    ub = np.minimum(np.maximum(ub, 0), np.float32(255) * np.float32(n)).astype(np.int64)
    assert np.all(ub >= sums), (n, sums[ub < sums][:5])
    # and it is not uselessly loose: within ~2e-5 relative + 3
    assert np.all(ub - sums <= 4 + 3e-5 * sums)


def _lround(v):
    return np.where(v >= 0, np.floor(v + 0.5), -np.floor(-v + 0.5)).astype(np.int64)


@pytest.mark.parametrize("res", [0.05, 0.1, 0.45, 0.2, 1.0 / 3.0])
def test_fast_cell_index_f32_matches_ieee_division(res):
    res = np.float32(res)
    inv = np.float32(1) / res
    rng = np.random.default_rng(7)
    k = rng.integers(-5000, 5000, 1_000_000)
    near = ((k + 0.5) * np.float64(res)).astype(np.float32)
    near = (near.view(np.int32) + rng.integers(-8, 9, near.size).astype(np.int32)).view(np.float32)
    c = np.concatenate([near, rng.uniform(-500, 500, 1_000_000).astype(np.float32)])
    c = c[np.isfinite(c)]
    q0 = (c * inv).astype(np.float32)
    n = np.rint(q0).astype(np.float32)
    margin = (np.float32(0.5) - np.abs((q0 - n).astype(np.float32))).astype(np.float32)
    fast = margin > (np.abs(q0) * np.float32(2.0 ** -20)).astype(np.float32)
    exact = _lround((c / res).astype(np.float32).astype(np.float64))
    assert fast.mean() > 0.45                      # the shortcut is the common case ...
    assert np.array_equal(n[fast].astype(np.int64), exact[fast])   # ... and is exact when taken


@pytest.mark.parametrize("res", [0.05, 0.1, 0.03, 0.05 / 1000])
def test_cell_index_f64_matches_ieee_division(res):
    inv = 1.0 / res
    rng = np.random.default_rng(11)
    k = rng.integers(-30000, 30000, 1_000_000)
    near = (k * res)
    near = (near.view(np.int64) + rng.integers(-6, 7, near.size)).view(np.float64)
    t = np.concatenate([near, rng.uniform(-3000, 3000, 1_000_000)])
    t = t[np.isfinite(t)]
    q0 = t * inv
    v0 = q0 - 0.5
    n = np.rint(v0)
    margin = 0.5 - np.abs(v0 - n)
    fast = margin > np.abs(q0) * 2.0 ** -46 + 2.0 ** -46
    exact = _lround(t / res - 0.5)
    assert fast.mean() > 0.45
    assert np.array_equal(n[fast].astype(np.int64), exact[fast])


@pytest.mark.parametrize("res,max_v", [(0.05, 10.0), (0.05, -3.7), (0.1, 100.0), (0.025, 0.0),
                                       (0.05, 512.35), (1.0 / 3.0, 7.0)])
def test_cell_index_fast_f32_estimate_is_exact_when_taken(res, max_v):
    """CellIndexFast (cmx_device.h): lround((max - v) / res - 0.5) -- MapLimits::GetCellIndex,
    f64 in the reference -- decided from an f32 estimate whenever the estimate is further from
    every half-integer than its error bound.  Values on, next to and far from rounding
    boundaries, both signs, large offsets."""
    res64, max64 = np.float64(res), np.float64(max_v)
    inv64 = np.float64(1.0) / res64
    rng = np.random.default_rng(11)
    k = rng.integers(-4000, 4000, 1_500_000)
    # v such that (max - v) / res - 0.5 is (nearly) a half-integer / an integer / anything
    near_half = (max64 - (k + 1.0) * res64).astype(np.float32)
    near_half = (near_half.view(np.int32) + rng.integers(-6, 7, near_half.size).astype(np.int32)) \
        .view(np.float32)
    near_int = (max64 - (k + 0.5) * res64).astype(np.float32)
    anything = rng.uniform(max_v - 300.0, max_v + 300.0, 1_500_000).astype(np.float32)
    v = np.concatenate([near_half, near_int, anything])
    v = v[np.isfinite(v)]
    f = np.float32
    maxf, invf = f(max64), f(inv64)
    a = (maxf - v).astype(np.float32)
    b = (a * invf).astype(np.float32)
    t = (b - f(0.5)).astype(np.float32)
    r = np.rint(t).astype(np.float32)
    margin = (f(0.5) - np.abs((t - r).astype(np.float32))).astype(np.float32)
    bound = (((np.abs(maxf) + np.abs(a)).astype(np.float32) * invf).astype(np.float32)
             * f(2.0 ** -23)).astype(np.float32)
    bound = (bound + (np.abs(b) * f(2.0 ** -21)).astype(np.float32)).astype(np.float32)
    bound = (bound + f(2.0 ** -20)).astype(np.float32)
    fast = (margin > bound) & (np.abs(t) < f(1e6))
    exact = _lround((max64 - v.astype(np.float64)) / res64 - 0.5)
    assert np.array_equal(r[fast].astype(np.int64), exact[fast])
    # the estimate decides all but a sliver of ordinary points
    assert fast[-anything.size:].mean() > 0.99


def test_rotate_z_equals_the_full_quaternion_rotation():
    """RotateZ (cmx_device.h) drops the products by the zero components of a yaw quaternion.
    Its x / y must equal Eigen's `_transformVector` order of operations (Rotate) bit for bit on
    finite inputs, zeros compared by value (a zero's sign disappears in the next addition)."""
    f = np.float32
    rng = np.random.default_rng(3)
    m = 2_000_000
    ang = rng.uniform(-np.pi, np.pi, m)
    w, z = np.cos(ang / 2).astype(f), np.sin(ang / 2).astype(f)
    vx = rng.uniform(-40, 40, m).astype(f)
    vy = rng.uniform(-40, 40, m).astype(f)
    vz = rng.uniform(-3, 3, m).astype(f)
    # some exact zeros / axis-aligned points / tiny values
    vx[:1000] = 0; vy[1000:2000] = 0; vz[2000:3000] = 0
    vx[3000:4000] *= f(1e-30); z[4000:5000] = 0; w[4000:5000] = 1
    zero = np.zeros(m, f)

    def cross(a, b):
        return (a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])
    qv = (zero, zero, z)
    v = (vx, vy, vz)
    uv = cross(qv, v)
    uv = tuple((c + c).astype(f) for c in uv)
    c = cross(qv, uv)
    full_x = ((vx + w * uv[0]).astype(f) + c[0]).astype(f)
    full_y = ((vy + w * uv[1]).astype(f) + c[1]).astype(f)
    uvx = (-(z * vy)).astype(f); uvy = (z * vx).astype(f)
    uvx = (uvx + uvx).astype(f); uvy = (uvy + uvy).astype(f)
    cx = (-(z * uvy)).astype(f); cy = (z * uvx).astype(f)
    fast_x = ((vx + w * uvx).astype(f) + cx).astype(f)
    fast_y = ((vy + w * uvy).astype(f) + cy).astype(f)
    assert np.array_equal(full_x, fast_x) and np.array_equal(full_y, fast_y)   # -0 == +0


def test_parent_cell_is_max_of_child_cells(oracle, synth):
    """The identity behind the early exit of node expansions (fast_2d.hip): a width-2h
    precomputation cell equals the maximum of the four width-h cells its children read
    (cells outside a level read 0), on the oracle's own PrecomputationGrid2D levels -- which
    the device stack equals bit for bit (tests/test_gpu_2d.py)."""
    cells, _, _ = synth.make_submap(5, 70, 53, 0.05, 6, 200, 30.0, 0.01)
    ny, nx = cells.shape
    for h in (1, 2, 4, 8, 16):
        small = oracle.precompute2d(cells, h).astype(np.int32)       # [y0 + h - 1][x0 + h - 1]
        big = oracle.precompute2d(cells, 2 * h).astype(np.int32)     # [y0 + 2h - 1][x0 + 2h - 1]

        def child(x0, y0):
            ix, iy = x0 + h - 1, y0 + h - 1
            ok = (ix >= 0) & (ix < small.shape[1]) & (iy >= 0) & (iy < small.shape[0])
            return np.where(ok, small[np.clip(iy, 0, small.shape[0] - 1),
                                      np.clip(ix, 0, small.shape[1] - 1)], 0)
        y0, x0 = np.meshgrid(np.arange(-2 * h + 1, ny), np.arange(-2 * h + 1, nx), indexing="ij")
        want = np.maximum(np.maximum(child(x0, y0), child(x0 + h, y0)),
                          np.maximum(child(x0, y0 + h), child(x0 + h, y0 + h)))
        np.testing.assert_array_equal(big, want)


def test_cropping_round_trip_is_the_identity_on_cell_values(oracle):
    """ProbabilityGrid::ComputeCroppedGrid copies cells as
    SetProbability(GetProbability(cell)) (probability_grid.cc:90-106), i.e. value -> cost ->
    probability = 1 - cost -> cost' = 1 - probability -> value'.  For every value in [1, 32767]
    value' == value, so the device crop (cmx_grid2d_crop) may copy the cells as they are."""
    _, value_to_cost, _ = oracle.value_tables()
    L = oracle.lib()
    for v in range(1, 32768):
        cost = np.float32(value_to_cost[v])
        probability = np.float32(1) - cost               # CorrespondenceCostToProbability
        back = np.float32(1) - probability               # ProbabilityToCorrespondenceCost
        assert L.orc_correspondence_cost_to_value(float(back)) == v


def test_product_probability_odds_table_equals_the_reference(oracle, tmp_path):
    """cartographer_amd/csrc/cmx_odds_table.h -- the host table cmx_grid3d_insert uploads --
    compiled on its own with g++ and compared, entry by entry, with the reference's own
    ComputeLookupTableToApplyOdds (oracle/_ref)."""
    import ctypes
    import os
    import subprocess
    ref = oracle.ref_lib()
    if ref is None:
        pytest.skip("reference tree not available and oracle/_ref not prebuilt")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "table.cc"
    src.write_text('#include "cmx_odds_table.h"\n'
                   'extern "C" void table(float p, uint16_t* out) {'
                   ' cmx::ProbabilityOddsTable(p, out); }\n')
    lib = tmp_path / "libtable.so"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
                           "-I", os.path.join(root, "cartographer_amd", "csrc"), "-o", str(lib),
                           str(src)])
    fn = ctypes.CDLL(str(lib)).table
    fn.argtypes = [ctypes.c_float, np.ctypeslib.ndpointer(np.uint16, flags="C_CONTIGUOUS")]
    for probability in (0.7, 0.4, 0.55, 0.49, 0.9, 0.1, 0.5, 0.62):
        got = np.empty(32768, np.uint16)
        fn(probability, got)
        cost_table = np.empty(32768, np.uint16)
        probability_table = np.empty(32768, np.uint16)
        ref.ref_odds_tables(probability, cost_table, probability_table)
        np.testing.assert_array_equal(got, probability_table)


class _Grid3DModel:
    """The ALGORITHM of cartographer_amd/csrc/grid_3d.hip in numpy (dense brick, extent pass with
    the first / last free-space sample of every ray, 16-voxel brick growth, DynamicGrid's doubling
    rule, all hits then all (ray, k) miss samples with first-writer-wins, marker clearing).  It
    checks the decomposition the kernels implement -- not their HIP syntax -- against the host
    voxel builder, which is pinned on the reference's own RangeDataInserter3D."""

    MARKER = 1 << 15

    def __init__(self, resolution, table_fn):
        self.res = np.float32(resolution)
        self.table_fn = table_fn
        self.grid_size = 128
        self.lo = None
        self.cells = None                      # [nz, ny, nx] uint16

    def _cell(self, p):
        q = (np.asarray(p, np.float32) / self.res).astype(np.float64)      # f32 divide
        return (np.sign(q) * np.floor(np.abs(q) + 0.5)).astype(np.int64)   # lround

    @staticmethod
    def _trunc_div(a, b):
        return np.sign(a) * (np.abs(a) // b)

    def insert(self, origin, returns, hit_p, miss_p, free):
        returns = np.asarray(returns, np.float32).reshape(-1, 3)
        if returns.shape[0] == 0:
            return
        hit_table, miss_table = self.table_fn(hit_p), self.table_fn(miss_p)
        o = self._cell(origin)
        h = self._cell(returns)
        d = h - o
        ns = np.abs(d).max(axis=1)
        assert (ns < (1 << 15)).all()
        # pass 0: extent from the hits and the first / last touched sample of every ray
        first = np.maximum(0, ns - free)
        has = first < ns
        safe = np.maximum(ns, 1)[:, None]
        c_first = o + self._trunc_div(d * first[:, None], safe)
        c_last = o + self._trunc_div(d * (ns - 1)[:, None], safe)
        pts = np.concatenate([h, c_first[has], c_last[has]])
        lo, hi = pts.min(axis=0), pts.max(axis=0)
        while not ((lo >= -(self.grid_size // 2)).all() and (hi < self.grid_size // 2).all()):
            self.grid_size *= 2
        nlo, nhi = lo // 16 * 16, hi // 16 * 16 + 15
        if self.cells is not None:
            cur_hi = self.lo + np.array(self.cells.shape[::-1]) - 1
            nlo, nhi = np.minimum(nlo, self.lo), np.maximum(nhi, cur_hi)
        dims = nhi - nlo + 1
        grown = np.zeros((dims[2], dims[1], dims[0]), np.uint16)
        if self.cells is not None:
            off = self.lo - nlo
            nz, ny, nx = self.cells.shape
            grown[off[2]:off[2] + nz, off[1]:off[1] + ny, off[0]:off[0] + nx] = self.cells
        self.cells, self.lo = grown, nlo

        def apply(cells_xyz, table):
            idx = cells_xyz - self.lo
            assert (idx >= 0).all() and (idx < dims).all(), "a voxel fell outside the extent"
            for x, y, z in idx:                                   # first writer wins
                old = self.cells[z, y, x]
                if old < self.MARKER:
                    self.cells[z, y, x] = table[old]
        apply(h, hit_table)
        for k in range(free):                                     # thread (i, k)
            position = ns - free + k
            ok = (position >= 0) & (position < ns)
            if ok.any():
                c = o + self._trunc_div(d[ok] * position[ok][:, None], ns[ok][:, None])
                apply(c, miss_table)
        self.cells[self.cells >= self.MARKER] -= self.MARKER

    def voxels(self):
        z, y, x = np.nonzero(self.cells) if self.cells is not None else ([], [], [])
        return [(int(a + self.lo[0]), int(b + self.lo[1]), int(c + self.lo[2]),
                 int(self.cells[c, b, a])) for a, b, c in zip(x, y, z)]


@pytest.mark.parametrize("seed", range(5))
def test_grid3d_device_algorithm_model_equals_the_host_builder(oracle, synth, tmp_path, seed):
    import ctypes
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "table.cc"
    src.write_text('#include "cmx_odds_table.h"\n'
                   'extern "C" void table(float p, uint16_t* out) {'
                   ' cmx::ProbabilityOddsTable(p, out); }\n')
    lib = tmp_path / "libtable.so"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
                           "-I", os.path.join(root, "cartographer_amd", "csrc"), "-o", str(lib),
                           str(src)])
    fn = ctypes.CDLL(str(lib)).table
    fn.argtypes = [ctypes.c_float, np.ctypeslib.ndpointer(np.uint16, flags="C_CONTIGUOUS")]
    cache = {}

    def table(p):
        if p not in cache:
            cache[p] = np.empty(32768, np.uint16)
            fn(p, cache[p])
        return cache[p]

    rng = np.random.default_rng(800 + seed)                 # test_random_insertions_3d's generator
    res = float(rng.choice([0.05, 0.1, 0.45, 1.0]))
    host, model = synth.HybridGrid(res), _Grid3DModel(res, table)
    for _ in range(8):
        origin = rng.uniform(-20, 20, 3).astype(np.float32) * np.float32(res)
        n = int(rng.integers(0, 60))
        d = rng.normal(size=(n, 3))
        d /= np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-9)
        rad = rng.uniform(0, 90 * res, (n, 1)) * (rng.uniform(size=(n, 1)) > 0.1)
        pts = (origin + d * rad).astype(np.float32)
        if n > 3:
            pts[1] = pts[0]
        hit, miss = float(rng.uniform(0.51, 0.95)), float(rng.uniform(0.05, 0.49))
        free = int(rng.choice([0, 1, 2, 10, 60]))
        host.insert(origin, pts, hit, miss, free)
        model.insert(origin, pts, hit, miss, free)
        assert model.grid_size == host.grid_size
        hv = host.voxels()
        want = sorted((int(r["x"]), int(r["y"]), int(r["z"]), int(r["value"])) for r in hv)
        assert sorted(model.voxels()) == want
