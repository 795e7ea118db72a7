"""oracle/_ref, 3D: the REFERENCE'S OWN real_time_correlative_scan_matcher_3d.cc,
fast_correlative_scan_matcher_3d.cc, precomputation_grid_3d.cc, rotational_scan_matcher.cc,
low_resolution_matcher.cc and the header-only mapping/3d/hybrid_grid.h, compiled unmodified from
/root/reference (oracle/Makefile `ref`) against the oracle's restatement (oracle/oracle_3d.cc).
Every comparison is bit-exact.

What this pins: the search loops, window / step-size arithmetic, candidate generation and
ordering, DiscretizeScan's shifted indices, ScoreCandidates, the branch-and-bound recursion and
its tie handling, the precomputation stack (shift schedule, half-resolution levels, 8-bit
quantisation), HybridGrid's growth / lookup / iteration, the rotational histogram matcher and
the low-resolution matcher.  What it does NOT pin: the four Eigen kernels the stand-in headers
restate in the oracle's order (oracle/ref_shims/Eigen/Geometry) -- see ref_shims/README.md.

Skipped where neither /root/reference nor a prebuilt oracle/_ref/libref.so exists.
"""
import math

import numpy as np
import pytest

from test_oracle_reference_pins_3d import (FAST3D_CLOUD, RT_CLOUD, RT_INITIAL_POSES,
                                           fast3d_fixture, quat_from_angle_axis, rt3d_fixture)


@pytest.fixture(scope="module")
def ref(oracle):
    lib = oracle.ref_lib()
    if lib is None:
        pytest.skip("reference tree not available and oracle/_ref not prebuilt")
    return lib


# ----------------------------------------------------------------------------
# HybridGrid: growth, iteration, GetCellIndex
# ----------------------------------------------------------------------------
@pytest.mark.parametrize("extent", [3, 60, 64, 65, 200, 700])
def test_hybrid_grid_size_and_contents_equal_the_reference(ref, oracle, extent):
    """DynamicGrid::Grow doubles until every written voxel fits (hybrid_grid.h:381-398); the
    oracle's HybridGridView restates the resulting grid_size() and the read side."""
    from cartographer_amd._lib import VOXEL_DTYPE
    rng = np.random.default_rng(extent)
    n = 400
    vox = np.zeros(n, VOXEL_DTYPE)
    vox["x"] = rng.integers(-extent, extent + 1, n)
    vox["y"] = rng.integers(-extent // 2, extent // 2 + 1, n)
    vox["z"] = rng.integers(-extent // 4, extent // 4 + 1, n)
    vox["value"] = rng.integers(1, 32768, n)
    # unique cells only: a later write of the same cell would overwrite in both
    _, first = np.unique(np.stack([vox["x"], vox["y"], vox["z"]], 1), axis=0, return_index=True)
    vox = vox[np.sort(first)]
    assert oracle.grid3d_size(0.1, vox) == oracle.ref_grid3d_size(0.1, vox)
    got = oracle.ref_grid3d_iterate(0.1, vox)
    want = np.stack([vox["x"], vox["y"], vox["z"], vox["value"]], 1).astype(np.int32)
    order = lambda a: a[np.lexsort((a[:, 0], a[:, 1], a[:, 2]))]       # noqa: E731
    np.testing.assert_array_equal(order(got), order(want))


def test_hybrid_grid_iteration_order_is_z_major_within_blocks(ref, oracle):
    """The iterator walks meta cells, then 8^3 blocks, then cells, each z-major
    (hybrid_grid.h:38-52); within one 8x8x8 block that is plain (z, y, x) order."""
    from cartographer_amd._lib import VOXEL_DTYPE
    cells = [(1, 2, 3), (0, 0, 0), (7, 7, 7), (3, 2, 1), (0, 1, 0), (1, 0, 0)]
    vox = np.zeros(len(cells), VOXEL_DTYPE)
    for i, (x, y, z) in enumerate(cells):
        vox[i] = (x, y, z, 100 + i, 0)
    got = oracle.ref_grid3d_iterate(0.05, vox)
    want = sorted(cells, key=lambda c: (c[2], c[1], c[0]))
    assert [tuple(r[:3]) for r in got] == want


def test_get_cell_index_equals_the_reference(ref, oracle):
    """hybrid_grid.h:428-433: f32 divide by the resolution, then lround -- including exact
    half-way points, where a multiply by the reciprocal would round differently."""
    rng = np.random.default_rng(5)
    for res in (0.05, 0.1, 0.45, 0.2):
        pts = rng.uniform(-30, 30, (20000, 3)).astype(np.float32)
        k = rng.integers(-400, 400, (2000, 3))
        half = ((k + 0.5) * np.float32(res)).astype(np.float32)
        pts = np.concatenate([pts, half, np.nextafter(half, np.float32(1e9)),
                              np.nextafter(half, np.float32(-1e9))])
        got = oracle.ref_grid3d_cell_index(res, pts)
        r = np.float32(res)
        want = np.stack([[int(math.floor(abs(float(v)) + 0.5)) * (1 if v >= 0 else -1)
                          for v in (pts[:, a] / r)] for a in range(3)], 1)
        np.testing.assert_array_equal(got, want)


# ----------------------------------------------------------------------------
# Real-time 3D
# ----------------------------------------------------------------------------
def _same_rt(a, b):
    assert np.float32(a["score"]) == np.float32(b["score"])
    np.testing.assert_array_equal(a["pose"], b["pose"])


@pytest.mark.parametrize("t,q", RT_INITIAL_POSES)
def test_rt3d_fixture_equals_the_reference(ref, oracle, synth, t, q):
    """The reference test's own grid, cloud and poses (real_time_..._3d_test.cc:36-117)."""
    vox = rt3d_fixture(synth).voxels()
    args = (0.1, vox, list(t) + list(q), RT_CLOUD, 0.3, math.radians(1.0), 1e-1, 1.0)
    _same_rt(oracle.rt3d_match(*args), oracle.ref_rt3d_match(*args))


@pytest.mark.parametrize("seed,window,ang,tw,rw", [(3, 0.2, 1.0, 0.1, 0.1), (5, 0.1, 2.0, 0.5, 3.0),
                                                   (8, 0.3, 0.5, 0.0, 0.0)])
def test_rt3d_synthetic_equals_the_reference(ref, oracle, synth, seed, window, ang, tw, rw):
    grid, world = synth.make_submap_3d(seed, 0.1, (8.0, 8.0, 4.0), 4, 8, 96)
    vox = grid.voxels()
    pos = world.free_position(seed + 1, 0.5)
    cloud = world.scan(pos, 0.3, 6, 64, seed=9)
    # a tilted initial orientation, so the quaternion products are not about one axis
    q = quat_from_angle_axis(0.31, [0.1, -0.2, 0.97])
    init = list(pos + np.array([0.07, -0.04, 0.02])) + q
    args = (0.1, vox, init, cloud, window, math.radians(ang), tw, rw)
    _same_rt(oracle.rt3d_match(*args), oracle.ref_rt3d_match(*args))


def test_rt3d_empty_grid_equals_the_reference(ref, oracle):
    from cartographer_amd._lib import VOXEL_DTYPE
    cloud = np.array([[1, 0, 0], [0, 2, 0.5]], np.float32)
    args = (0.1, np.zeros(0, VOXEL_DTYPE), [0, 0, 0, 1, 0, 0, 0], cloud, 0.1, 0.01, 0.5, 0.5)
    _same_rt(oracle.rt3d_match(*args), oracle.ref_rt3d_match(*args))


# ----------------------------------------------------------------------------
# Rotational scan matcher
# ----------------------------------------------------------------------------
@pytest.mark.parametrize("size", [1, 7, 16, 120])
def test_rotational_match_equals_the_reference(ref, oracle, size):
    rng = np.random.default_rng(size)
    submap = rng.uniform(0, 3, size).astype(np.float32)
    scan = rng.uniform(0, 3, size).astype(np.float32)
    angles = np.concatenate([np.linspace(-3.2, 3.2, 57), [0.0, math.pi, -math.pi,
                                                         math.pi / size]]).astype(np.float32)
    for initial in (0.0, 0.37, -2.9, 5.0):
        np.testing.assert_array_equal(oracle.rotational_match(submap, scan, initial, angles),
                                      oracle.ref_rotational_match(submap, scan, initial, angles))
    zero = np.zeros(size, np.float32)      # normalisation < 1e-3 -> score 1
    np.testing.assert_array_equal(oracle.rotational_match(zero, scan, 0.1, angles),
                                  oracle.ref_rotational_match(zero, scan, 0.1, angles))


# ----------------------------------------------------------------------------
# Fast 3D: precomputation stack, Match, MatchFullSubmap
# ----------------------------------------------------------------------------
REF_OPTIONS = dict(depth=6, frd=6, min_rot=0.1, min_low=0.15, lin_xy=0.8, lin_z=0.8, ang=0.3)


def _pair(oracle, res, vox, low_res, low_vox, hist, depth, frd, min_rot, min_low, lin_xy, lin_z,
          ang):
    args = (res, vox, low_res, low_vox, hist, depth, frd, min_rot, min_low, lin_xy, lin_z, ang)
    return (oracle.FastCorrelativeScanMatcher3D(*args),
            oracle.ReferenceFastCorrelativeScanMatcher3D(*args))


def _same(a, b):
    assert a["found"] == b["found"]
    if not a["found"]:
        return
    for key in ("score", "rotational_score", "low_resolution_score"):
        assert np.float32(a[key]) == np.float32(b[key]), key
    np.testing.assert_array_equal(a["pose"], b["pose"])


@pytest.mark.parametrize("depth,frd", [(6, 6), (8, 3), (5, 1), (1, 1), (4, 9)])
def test_precomputation_stack_equals_the_reference(ref, oracle, synth, depth, frd):
    """PrecomputationGridStack3D (fast_..._3d.cc:57-77) over precomputation_grid_3d.cc: every
    level's non-zero cells and 8-bit values."""
    grid, _ = synth.make_submap_3d(11, 0.1, (6.0, 5.0, 3.0), 3, 8, 64)
    vox = grid.voxels()
    om, rm = _pair(oracle, 0.1, vox, 0.1, vox, np.zeros(8, np.float32),
                   **dict(REF_OPTIONS, depth=depth, frd=frd))
    for d in range(depth):
        np.testing.assert_array_equal(om.level(d), rm.level(d), err_msg=f"depth {d}")


def test_fast3d_fixture_equals_the_reference(ref, oracle, synth):
    """The reference test's own cloud, grids and options (fast_..._3d_test.cc:40-204)."""
    rng = np.random.default_rng(42)
    hist = np.zeros(10, np.float32)
    ident = [0, 0, 0, 1, 0, 0, 0]
    for _ in range(4):
        t = 0.7 * rng.uniform(-1, 1, 3)
        theta = 0.2 * rng.uniform(-1, 1)
        vox = fast3d_fixture(synth, t, theta).voxels()
        om, rm = _pair(oracle, 0.05, vox, 0.05, vox, hist, **REF_OPTIONS)
        a = om.match(ident, ident, [1, 0, 0, 0], FAST3D_CLOUD, FAST3D_CLOUD, hist, 0.1)
        b = rm.match(ident, ident, [1, 0, 0, 0], FAST3D_CLOUD, FAST3D_CLOUD, hist, 0.1)
        assert a["found"]
        _same(a, b)
        far = np.array([[42, 42, 42]], np.float32)        # low-resolution matcher rejects
        _same(om.match(ident, ident, [1, 0, 0, 0], FAST3D_CLOUD, far, hist, 0.1),
              rm.match(ident, ident, [1, 0, 0, 0], FAST3D_CLOUD, far, hist, 0.1))
        q = [1, 0, 0, 0]
        _same(om.match_full_submap(q, q, q, FAST3D_CLOUD, FAST3D_CLOUD, hist, 0.1),
              rm.match_full_submap(q, q, q, FAST3D_CLOUD, FAST3D_CLOUD, hist, 0.1))


@pytest.mark.parametrize("seed,depth,frd", [(21, 6, 3), (22, 5, 2), (23, 7, 1)])
def test_fast3d_synthetic_equals_the_reference(ref, oracle, synth, seed, depth, frd):
    """Hi-res 0.1 m / low-res 0.45 m grids, half-resolution levels, a selective yaw filter,
    non-identity node / submap poses and a tilted gravity alignment."""
    grid, world = synth.make_submap_3d(seed, 0.1, (9.0, 8.0, 4.0), 5, 10, 128)
    low, _ = synth.make_submap_3d(seed, 0.45, (9.0, 8.0, 4.0), 5, 10, 128)
    vox, low_vox = grid.voxels(), low.voxels()
    rng = np.random.default_rng(seed)
    hist = rng.uniform(0.0, 1.0, 120).astype(np.float32)
    hist[10:14] += 6.0
    scan_hist = np.roll(hist, -19).copy()
    pos = world.free_position(seed + 3, 0.6)
    yaw = 0.4
    hi = world.scan(pos, yaw, 8, 96, seed=1)
    lo = hi[::7].copy()
    om, rm = _pair(oracle, 0.1, vox, 0.45, low_vox, hist, depth, frd, 0.9, 0.3, 1.5, 0.5,
                   math.radians(20.0))
    submap_pose = [0.3, -0.2, 0.1] + quat_from_angle_axis(0.2, [0, 0, 1])
    c, s = math.cos(0.2), math.sin(0.2)
    local = np.array([pos[0] + 0.35, pos[1] - 0.25, pos[2] + 0.1])
    node_t = [submap_pose[0] + c * local[0] - s * local[1],
              submap_pose[1] + s * local[0] + c * local[1], submap_pose[2] + local[2]]
    node_pose = node_t + quat_from_angle_axis(0.2 + yaw + 0.1, [0, 0, 1])
    gravity = quat_from_angle_axis(0.01, [1, 0, 0])
    found = []
    for min_score in (0.05, 0.2, 0.95):
        a = om.match(node_pose, submap_pose, gravity, hi, lo, scan_hist, min_score)
        b = rm.match(node_pose, submap_pose, gravity, hi, lo, scan_hist, min_score)
        _same(a, b)
        found.append(a["found"])
    assert found[0] and found[1] and not found[-1]


def test_fast3d_c5_shaped_submap_equals_the_reference(ref, oracle, synth):
    """One submap of BASELINE.json config C5 (tools/time_configs.py c5): depth 8 /
    full-resolution depth 3, pose_graph.lua windows, ~2.7 k / ~200 points."""
    size = (15.0, 15.0, 7.5)
    grid, world = synth.make_submap_3d(42, 0.1, size, 8, 32, 512)
    low, _ = synth.make_submap_3d(42, 0.45, size, 8, 32, 512)
    vox, low_vox = grid.voxels(), low.voxels()
    rng = np.random.default_rng(1)
    hist = rng.uniform(0.0, 1.0, 120).astype(np.float32)
    hist[10:14] += 6.0
    pos = world.free_position(77, 0.6)
    yaw = 0.4
    full = world.scan(pos, yaw, 32, 512, seed=1)
    hi, lo = full[::6].copy(), full[::80].copy()
    scan_hist = np.roll(hist, -19).copy()
    om, rm = _pair(oracle, 0.1, vox, 0.45, low_vox, hist, 8, 3, 0.77, 0.35, 5.0, 1.0,
                   math.radians(15.0))
    node = [pos[0] + 0.8, pos[1] - 0.6, pos[2] + 0.2] + quat_from_angle_axis(yaw + 0.1, [0, 0, 1])
    ident = [0, 0, 0, 1, 0, 0, 0]
    a = om.match(node, ident, [1, 0, 0, 0], hi, lo, scan_hist, 0.2)
    b = rm.match(node, ident, [1, 0, 0, 0], hi, lo, scan_hist, 0.2)
    assert a["found"]
    _same(a, b)


def test_fast3d_full_submap_synthetic_equals_the_reference(ref, oracle, synth):
    grid, world = synth.make_submap_3d(31, 0.2, (8.0, 8.0, 3.0), 4, 8, 96)
    vox = grid.voxels()
    hist = np.zeros(16, np.float32)
    pos = world.free_position(5, 0.6)
    hi = world.scan(pos, 0.0, 6, 64, seed=2)
    lo = hi[::5].copy()
    om, rm = _pair(oracle, 0.2, vox, 0.2, vox, hist, 5, 2, 0.5, 0.25, 1.0, 1.0, 0.1)
    q = [1, 0, 0, 0]
    node_q = quat_from_angle_axis(0.2, [0, 0, 1])
    for nq in (q, node_q):
        a = om.match_full_submap(nq, q, q, hi, lo, hist, 0.4)
        b = rm.match_full_submap(nq, q, q, hi, lo, hist, 0.4)
        _same(a, b)
    assert a["found"]


# ----------------------------------------------------------------------------
# HybridGrid write side + RangeDataInserter3D: the host builder every 3D
# fixture of the tests and tools comes from, against the reference's own
# range_data_inserter_3d.cc driving its own HybridGrid.
# ----------------------------------------------------------------------------
def _same_grid(host, reference):
    assert host.grid_size == reference.grid_size
    np.testing.assert_array_equal(host.voxels(), reference.voxels())


def test_range_data_inserter_3d_fixture_equals_the_reference(ref, oracle, synth):
    """RangeDataInserter3DTest::InsertPointCloud (range_data_inserter_3d_test.cc:28-53): 1 m
    grid, origin (0, 0, -4), four returns at z = 4, hit 0.7 / miss 0.4, 1000 free-space voxels;
    inserted twice like its InsertPointCloud / ProbabilityProgression tests."""
    returns = np.array([[-3, -1, 4], [-2, 0, 4], [-1, 1, 4], [0, 2, 4]], np.float32)
    origin = np.array([0, 0, -4], np.float32)
    host, reference = synth.HybridGrid(1.0), oracle.ReferenceHybridGrid(1.0)
    for _ in range(2):
        host.insert(origin, returns, 0.7, 0.4, 1000)
        reference.insert(origin, returns, 0.7, 0.4, 1000)
        _same_grid(host, reference)
    assert len(reference.voxels()) > 10


@pytest.mark.parametrize("seed,free", [(2, 2), (7, 0), (9, 40), (11, 5)])
def test_range_data_inserter_3d_scans_equal_the_reference(ref, oracle, synth, seed, free):
    """Eight scans of a synthetic room (as synth.make_submap_3d inserts them): re-updates through
    the odds tables, hits before misses, the last `num_free_space_voxels` voxels of each ray,
    DynamicGrid growth from 128 to 256 voxels."""
    world = synth.World3D(seed, (15.0, 15.0, 7.5))
    host, reference = synth.HybridGrid(0.1), oracle.ReferenceHybridGrid(0.1)
    for p in range(8):
        pos = world.free_position(seed * 1009 + p, 0.5)
        yaw = 0.37 * p
        sensor = world.scan(pos, yaw, 8, 96, seed=seed * 31 + p).astype(np.float64)
        c, s = math.cos(yaw), math.sin(yaw)
        in_map = np.stack([pos[0] + c * sensor[:, 0] - s * sensor[:, 1],
                           pos[1] + s * sensor[:, 0] + c * sensor[:, 1],
                           pos[2] + sensor[:, 2]], 1).astype(np.float32)
        hit, miss = (0.55, 0.49) if p % 3 == 2 else (0.7, 0.4)
        host.insert(pos.astype(np.float32), in_map, hit, miss, free)
        reference.insert(pos.astype(np.float32), in_map, hit, miss, free)
        _same_grid(host, reference)
    assert reference.grid_size == 256


def test_make_submap_3d_equals_a_reference_built_one(ref, oracle, synth):
    """The C4/C5 fixture of tools/time_configs.py at reduced scan density, rebuilt by the
    reference's inserter from the same scans."""
    seed, res, size, poses, rings, az, free = 42, 0.1, (15.0, 15.0, 7.5), 8, 8, 128, 2
    grid, world = synth.make_submap_3d(seed, res, size, poses, rings, az, free)
    reference = oracle.ReferenceHybridGrid(res)
    for p in range(poses):
        pos = world.free_position(seed * 1009 + p, 0.5)
        yaw = 0.37 * p
        sensor = world.scan(pos, yaw, rings, az, seed=seed * 31 + p).astype(np.float64)
        c, s = np.cos(yaw), np.sin(yaw)
        in_map = np.stack([pos[0] + c * sensor[:, 0] - s * sensor[:, 1],
                           pos[1] + s * sensor[:, 0] + c * sensor[:, 1],
                           pos[2] + sensor[:, 2]], 1).astype(np.float32)
        reference.insert(pos.astype(np.float32), in_map, 0.7, 0.4, free)
    _same_grid(grid, reference)


def test_hybrid_grid_set_get_probability_equal_the_reference(ref, oracle, synth):
    host, reference = synth.HybridGrid(0.05), oracle.ReferenceHybridGrid(0.05)
    rng = np.random.default_rng(3)
    cells = rng.integers(-90, 90, (200, 3))
    for c in cells:
        p = float(rng.uniform(0, 1))
        host.set_probability(c, p)
        reference.set_probability(c, p)
    _same_grid(host, reference)
    for c in list(cells[:20]) + [np.array([1000, 0, 0]), np.array([0, 0, -3000])]:
        assert np.float32(host.get_probability(c)) == np.float32(reference.get_probability(c))


@pytest.mark.parametrize("seed", range(4))
def test_random_insertions_3d_equal_the_reference(ref, oracle, synth, seed):
    """Seeded random range data into the voxel grid: random resolutions, origins, ranges that
    force DynamicGrid growth, duplicate returns, zero-length rays, 0 .. 60 free-space voxels."""
    rng = np.random.default_rng(800 + seed)
    res = float(rng.choice([0.05, 0.1, 0.45, 1.0]))
    host, reference = synth.HybridGrid(res), oracle.ReferenceHybridGrid(res)
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
        reference.insert(origin, pts, hit, miss, free)
        _same_grid(host, reference)


# ----------------------------------------------------------------------------
# IntensityHybridGrid: RangeDataInserter3D::Insert with intensities (f3)
# ----------------------------------------------------------------------------
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_intensity_insertion_equals_the_reference(ref, oracle, seed):
    """InsertIntensitiesIntoGrid + IntensityHybridGrid::AddIntensity (range_data_inserter_3d.cc:
    54-70, hybrid_grid.h:552-556) through the reference's OWN inserter and grid: the restatement
    returns the same voxels, counts and -- bit for bit -- the same f32 sums (they depend on the
    point order); returns above the threshold are skipped, a NaN intensity is not (`>`), a cloud
    without intensities inserts nothing, several points of one scan share voxels."""
    rng = np.random.default_rng(seed)
    res = 0.1
    hybrid, intensity = oracle.ReferenceHybridGrid(res), oracle.ReferenceIntensityHybridGrid(res)
    vox = np.zeros(0, oracle.INTENSITY_VOXEL_DTYPE)
    for scan in range(4):
        n = 700
        ret = rng.uniform(-1.5, 1.5, (n, 3)).astype(np.float32)
        ret[:50] = ret[50:100] + rng.uniform(-0.01, 0.01, (50, 3)).astype(np.float32)   # shared voxels
        ints = rng.uniform(0.0, 60.0, n).astype(np.float32)
        ints[7] = np.nan
        use = None if scan == 2 else ints
        hybrid.insert_with_intensities(intensity, [0.1, -0.2, 0.0], ret, use,
                                       intensity_threshold=40.0)
        vox = oracle.insert_intensities(res, vox, ret, use, 40.0)
    want = intensity.voxels()
    assert len(want) > 1000 and want["count"].max() >= 2
    assert want.tobytes() == vox.tobytes()
