"""GPU parity tests of the 3D matchers: the HIP path (through the C ABI) against
the CPU oracle on identical point clouds + hybrid grids.

Bars: precomputation levels and integer candidate sums are bit-exact (checked
through bit-equal f32 scores); returned scores, rotational / low-resolution
scores are bit-equal f32; poses are bit-equal (f32 values widened to f64).
"""
import math

import numpy as np
import pytest

from test_oracle_reference_pins_3d import (FAST3D_CLOUD, RT_CLOUD, RT_INITIAL_POSES,
                                           fast3d_fixture, quat_from_angle_axis, rt3d_fixture)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sm3():
    from cartographer_amd import _lib, scan_matching_3d
    assert _lib.lib().cmx_device_count() >= 1, "no HIP device: these tests need the GPU"
    return scan_matching_3d


def _pose7(p):
    return np.array(list(p.translation) + list(p.rotation))


# ----------------------------------------------------------------------------
# Real-time 3D
# ----------------------------------------------------------------------------
@pytest.mark.parametrize("t,q", RT_INITIAL_POSES)
def test_rt3d_reference_fixture(sm3, oracle, synth, t, q):
    """RealTimeCorrelativeScanMatcher3DTest (real_time_..._3d_test.cc:36-117)."""
    g = rt3d_fixture(synth)
    vox = g.voxels()
    ref = oracle.rt3d_match(0.1, vox, list(t) + list(q), RT_CLOUD, 0.3, math.radians(1.0), 1e-1,
                            1.0)
    m = sm3.RealTimeCorrelativeScanMatcher3D(0.3, math.radians(1.0), 1e-1, 1.0)
    score, pose = m.match(sm3.Rigid3d(tuple(t), tuple(q)), RT_CLOUD, 0.1, vox)
    assert m.last_stats["candidates_scored"] == ref["num_candidates"]
    assert np.float32(score) == np.float32(ref["score"])
    np.testing.assert_array_equal(_pose7(pose), ref["pose"])


@pytest.mark.parametrize("seed,window,ang", [(3, 0.2, 1.0), (5, 0.1, 2.0)])
def test_rt3d_synthetic_world(sm3, oracle, synth, seed, window, ang):
    grid, world = synth.make_submap_3d(seed, 0.1, (8.0, 8.0, 4.0), 4, 8, 96)
    vox = grid.voxels()
    pos = world.free_position(seed + 1, 0.5)
    yaw = 0.3
    cloud = world.scan(pos, yaw, 6, 64, seed=9)
    q = quat_from_angle_axis(yaw + 0.01, [0, 0, 1])
    init = list(pos + np.array([0.07, -0.04, 0.02])) + q
    ref = oracle.rt3d_match(0.1, vox, init, cloud, window, math.radians(ang), 0.1, 0.1)
    m = sm3.RealTimeCorrelativeScanMatcher3D(window, math.radians(ang), 0.1, 0.1)
    score, pose = m.match(sm3.Rigid3d(tuple(init[:3]), tuple(init[3:])), cloud, 0.1, vox)
    assert m.last_stats["candidates_scored"] == ref["num_candidates"]
    assert np.float32(score) == np.float32(ref["score"])
    np.testing.assert_array_equal(_pose7(pose), ref["pose"])


def test_rt3d_empty_grid_and_invalid(sm3, oracle):
    from cartographer_amd._lib import CmxError, INVALID_ARGUMENT, VOXEL_DTYPE
    empty = np.zeros(0, VOXEL_DTYPE)
    cloud = np.array([[1, 0, 0], [0, 2, 0.5]], np.float32)
    ref = oracle.rt3d_match(0.1, empty, [0, 0, 0, 1, 0, 0, 0], cloud, 0.1, 0.01, 0.5, 0.5)
    m = sm3.RealTimeCorrelativeScanMatcher3D(0.1, 0.01, 0.5, 0.5)
    score, pose = m.match(sm3.Rigid3d(), cloud, 0.1, empty)
    assert np.float32(score) == np.float32(ref["score"])
    np.testing.assert_array_equal(_pose7(pose), ref["pose"])
    with pytest.raises(CmxError) as e:
        m.match(sm3.Rigid3d(), np.zeros((0, 3), np.float32), 0.1, empty)
    assert e.value.status == INVALID_ARGUMENT


# ----------------------------------------------------------------------------
# Fast 3D: precomputation stack
# ----------------------------------------------------------------------------
def _both(sm3, oracle, res, vox, grid_size, low_res, low_vox, hist, **opt):
    om = oracle.FastCorrelativeScanMatcher3D(
        res, vox, low_res, low_vox, hist, opt["branch_and_bound_depth"],
        opt["full_resolution_depth"], opt["min_rotational_score"],
        opt["min_low_resolution_score"], opt["linear_xy_search_window"],
        opt["linear_z_search_window"], opt["angular_search_window"])
    gm = sm3.FastCorrelativeScanMatcher3D(res, vox, grid_size, low_res, low_vox, hist, **opt)
    return om, gm


REF_OPTIONS = dict(branch_and_bound_depth=6, full_resolution_depth=6, min_rotational_score=0.1,
                   min_low_resolution_score=0.15, linear_xy_search_window=0.8,
                   linear_z_search_window=0.8, angular_search_window=0.3)


@pytest.mark.parametrize("depth,frd", [(6, 6), (8, 3), (5, 1), (1, 1)])
def test_fast3d_stack_matches_oracle(sm3, oracle, synth, depth, frd):
    grid, _ = synth.make_submap_3d(11, 0.1, (6.0, 5.0, 3.0), 3, 8, 64)
    vox = grid.voxels()
    opt = dict(REF_OPTIONS, branch_and_bound_depth=depth, full_resolution_depth=frd)
    om, gm = _both(sm3, oracle, 0.1, vox, grid.grid_size, 0.1, vox, np.zeros(8, np.float32), **opt)
    for d in range(depth):
        np.testing.assert_array_equal(gm.level(d), om.level(d), err_msg=f"depth {d}")


# ----------------------------------------------------------------------------
# Fast 3D: Match / MatchFullSubmap
# ----------------------------------------------------------------------------
def _assert_result(ref, got):
    assert (got is not None) == ref["found"]
    if got is None:
        return
    assert np.float32(got["score"]) == np.float32(ref["score"])
    assert np.float32(got["rotational_score"]) == np.float32(ref["rotational_score"])
    assert np.float32(got["low_resolution_score"]) == np.float32(ref["low_resolution_score"])
    np.testing.assert_array_equal(_pose7(got["pose_estimate"]), ref["pose"])


def test_fast3d_reference_fixture(sm3, oracle, synth):
    """FastCorrelativeScanMatcher3DTest.CorrectPoseForMatch (fast_..._3d_test.cc:146-178)."""
    rng = np.random.default_rng(42)
    hist = np.zeros(10, np.float32)
    ident = [0, 0, 0, 1, 0, 0, 0]
    for _ in range(6):
        t = 0.7 * rng.uniform(-1, 1, 3)
        theta = 0.2 * rng.uniform(-1, 1)
        g = fast3d_fixture(synth, t, theta)
        vox = g.voxels()
        om, gm = _both(sm3, oracle, 0.05, vox, g.grid_size, 0.05, vox, hist, **REF_OPTIONS)
        ref = om.match(ident, ident, [1, 0, 0, 0], FAST3D_CLOUD, FAST3D_CLOUD, hist, 0.1)
        data = sm3.TrajectoryNodeData(FAST3D_CLOUD, FAST3D_CLOUD, hist)
        got = gm.match(sm3.Rigid3d(), sm3.Rigid3d(), data, 0.1)
        assert ref["found"]
        _assert_result(ref, got)
        assert gm.last_stats["coarse_candidates"] == ref["coarse_candidates"]
        far = np.array([[42, 42, 42]], np.float32)
        ref2 = om.match(ident, ident, [1, 0, 0, 0], FAST3D_CLOUD, far, hist, 0.1)
        got2 = gm.match(sm3.Rigid3d(), sm3.Rigid3d(),
                        sm3.TrajectoryNodeData(FAST3D_CLOUD, far, hist), 0.1)
        assert not ref2["found"]
        _assert_result(ref2, got2)


def test_fast3d_reference_full_submap(sm3, oracle, synth):
    """CorrectPoseForMatchFullSubmap (:180-204)."""
    rng = np.random.default_rng(7)
    t = 0.7 * rng.uniform(-1, 1, 3)
    theta = 0.2 * rng.uniform(-1, 1)
    g = fast3d_fixture(synth, t, theta)
    vox = g.voxels()
    hist = np.zeros(10, np.float32)
    om, gm = _both(sm3, oracle, 0.05, vox, g.grid_size, 0.05, vox, hist, **REF_OPTIONS)
    ref = om.match_full_submap([1, 0, 0, 0], [1, 0, 0, 0], [1, 0, 0, 0], FAST3D_CLOUD,
                               FAST3D_CLOUD, hist, 0.1)
    got = gm.match_full_submap([1, 0, 0, 0], [1, 0, 0, 0],
                               sm3.TrajectoryNodeData(FAST3D_CLOUD, FAST3D_CLOUD, hist), 0.1)
    assert ref["found"]
    _assert_result(ref, got)


@pytest.mark.parametrize("seed,depth,frd", [(21, 6, 3), (22, 5, 2)])
def test_fast3d_synthetic_world(sm3, oracle, synth, seed, depth, frd):
    """Hi-res 0.1 m / low-res 0.45 m grids, half-resolution levels, a histogram
    pre-filter that removes most yaws, non-identity node / submap poses."""
    grid, world = synth.make_submap_3d(seed, 0.1, (9.0, 8.0, 4.0), 5, 10, 128)
    low, _ = synth.make_submap_3d(seed, 0.45, (9.0, 8.0, 4.0), 5, 10, 128)
    vox, low_vox = grid.voxels(), low.voxels()
    rng = np.random.default_rng(seed)
    hist = rng.uniform(0.0, 1.0, 120).astype(np.float32)
    hist[10:14] += 6.0      # a dominant direction so the yaw filter is selective
    # The node's histogram is the submap's turned by the node's yaw in the submap
    # frame (~0.5 rad = 19 buckets), so only yaws near zero pass the filter.
    scan_hist = np.roll(hist, -19).copy()
    pos = world.free_position(seed + 3, 0.6)
    yaw = 0.4
    hi = world.scan(pos, yaw, 8, 96, seed=1)
    lo = hi[::7].copy()
    opt = dict(branch_and_bound_depth=depth, full_resolution_depth=frd, min_rotational_score=0.9,
               min_low_resolution_score=0.3, linear_xy_search_window=1.5,
               linear_z_search_window=0.5, angular_search_window=math.radians(20.0))
    om, gm = _both(sm3, oracle, 0.1, vox, grid.grid_size, 0.45, low_vox, hist, **opt)
    submap_pose = [0.3, -0.2, 0.1] + quat_from_angle_axis(0.2, [0, 0, 1])
    # global node pose = submap pose * (true pose in the submap frame, perturbed)
    c, s = math.cos(0.2), math.sin(0.2)
    local = np.array([pos[0] + 0.35, pos[1] - 0.25, pos[2] + 0.1])
    node_t = [submap_pose[0] + c * local[0] - s * local[1],
              submap_pose[1] + s * local[0] + c * local[1], submap_pose[2] + local[2]]
    node_pose = node_t + quat_from_angle_axis(0.2 + yaw + 0.1, [0, 0, 1])
    gravity = quat_from_angle_axis(0.01, [1, 0, 0])
    for min_score in (0.3, 0.95):     # best scores here are 0.25-0.29 (the found case, with
        # min_score 0.15, is test_gpu_zz_new.py::test_fast3d_synthetic_world_found)
        ref = om.match(node_pose, submap_pose, gravity, hi, lo, scan_hist, min_score)
        data = sm3.TrajectoryNodeData(hi, lo, scan_hist, tuple(gravity))
        got = gm.match(sm3.Rigid3d(tuple(node_pose[:3]), tuple(node_pose[3:])),
                       sm3.Rigid3d(tuple(submap_pose[:3]), tuple(submap_pose[3:])), data,
                       min_score)
        _assert_result(ref, got)
        assert gm.last_stats["num_scans"] == ref["num_scans"]
        assert gm.last_stats["coarse_candidates"] == ref["coarse_candidates"]
    assert 0 < ref["num_scans"] < 40      # the filter is selective but not empty


def test_fast3d_full_submap_synthetic(sm3, oracle, synth):
    grid, world = synth.make_submap_3d(31, 0.2, (8.0, 8.0, 3.0), 4, 8, 96)
    vox = grid.voxels()
    hist = np.zeros(16, np.float32)     # zero histograms: every yaw passes (score 1)
    pos = world.free_position(5, 0.6)
    hi = world.scan(pos, 0.0, 6, 64, seed=2)
    lo = hi[::5].copy()
    opt = dict(branch_and_bound_depth=5, full_resolution_depth=2, min_rotational_score=0.5,
               min_low_resolution_score=0.25, linear_xy_search_window=1.0,
               linear_z_search_window=1.0, angular_search_window=0.1)
    om, gm = _both(sm3, oracle, 0.2, vox, grid.grid_size, 0.2, vox, hist, **opt)
    ref = om.match_full_submap([1, 0, 0, 0], [1, 0, 0, 0], [1, 0, 0, 0], hi, lo, hist, 0.4)
    got = gm.match_full_submap([1, 0, 0, 0], [1, 0, 0, 0], sm3.TrajectoryNodeData(hi, lo, hist),
                               0.4)
    _assert_result(ref, got)
    assert ref["found"]


def test_fast3d_invalid_arguments(sm3, synth):
    from cartographer_amd._lib import CmxError, INVALID_ARGUMENT
    g = fast3d_fixture(synth, [0, 0, 0], 0.0)
    vox = g.voxels()
    with pytest.raises(CmxError) as e:
        sm3.FastCorrelativeScanMatcher3D(0.05, vox, g.grid_size, 0.05, vox, np.zeros(4, np.float32),
                                         branch_and_bound_depth=0)
    assert e.value.status == INVALID_ARGUMENT
    gm = sm3.FastCorrelativeScanMatcher3D(0.05, vox, g.grid_size, 0.05, vox,
                                          np.zeros(4, np.float32), **REF_OPTIONS)
    with pytest.raises(CmxError) as e:   # histogram size mismatch
        gm.match(sm3.Rigid3d(), sm3.Rigid3d(),
                 sm3.TrajectoryNodeData(FAST3D_CLOUD, FAST3D_CLOUD, np.zeros(7, np.float32)), 0.1)
    assert e.value.status == INVALID_ARGUMENT


# ----------------------------------------------------------------------------
# BASELINE-sized inputs
# ----------------------------------------------------------------------------
def test_rt3d_c4_sized_cloud_and_grid(sm3, oracle, synth):
    """C4's data sizes (64 x 1024 cloud, 15 x 15 x 7.5 m room at 0.1 m) with a window the
    oracle finishes in seconds: 27 translations x 27 rotations over all 65 536 points."""
    grid, world = synth.make_submap_3d(42, 0.1, (15.0, 15.0, 7.5), 8, 32, 512)
    vox = grid.voxels()
    pos = world.free_position(77, 0.5)
    cloud = world.scan(pos, 0.3, 64, 1024, seed=9)
    assert cloud.shape[0] > 60000
    q = quat_from_angle_axis(0.31, [0, 0, 1])
    init = list(pos + np.array([0.07, -0.04, 0.02])) + q
    r_max = float(np.linalg.norm(cloud, axis=1).max())
    step = (1 - 1e-3) * math.acos(1 - 0.1 ** 2 / (2 * r_max ** 2))
    ang = 1.2 * step                                # lround(ang / step) == 1
    ref = oracle.rt3d_match(0.1, vox, init, cloud, 0.1, ang, 0.1, 0.1)
    assert ref["num_candidates"] == 27 * 27
    m = sm3.RealTimeCorrelativeScanMatcher3D(0.1, ang, 0.1, 0.1)
    score, pose = m.match(sm3.Rigid3d(tuple(init[:3]), tuple(init[3:])), cloud, 0.1, vox)
    assert m.last_stats["candidates_scored"] == ref["num_candidates"]
    assert np.float32(score) == np.float32(ref["score"])
    np.testing.assert_array_equal(_pose7(pose), ref["pose"])


def test_fast3d_c5_sized_submap(sm3, oracle, synth):
    """One submap of C5: hi 0.1 m / low 0.45 m, depth 8 / full_resolution_depth 3, the
    pose_graph.lua windows (5 m, 1 m, 15 deg), 120-bin histograms."""
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
    opt = dict(branch_and_bound_depth=8, full_resolution_depth=3, min_rotational_score=0.77,
               min_low_resolution_score=0.35, linear_xy_search_window=5.0,
               linear_z_search_window=1.0, angular_search_window=math.radians(15.0))
    om, gm = _both(sm3, oracle, 0.1, vox, grid.grid_size, 0.45, low_vox, hist, **opt)
    node = [pos[0] + 0.8, pos[1] - 0.6, pos[2] + 0.2] + quat_from_angle_axis(yaw + 0.1, [0, 0, 1])
    ident = [0, 0, 0, 1, 0, 0, 0]
    for min_score in (0.2, 0.6):
        ref = om.match(node, ident, [1, 0, 0, 0], hi, lo, scan_hist, min_score)
        got = gm.match(sm3.Rigid3d(tuple(node[:3]), tuple(node[3:])), sm3.Rigid3d(),
                       sm3.TrajectoryNodeData(hi, lo, scan_hist), min_score)
        _assert_result(ref, got)
        assert gm.last_stats["num_scans"] == ref["num_scans"] > 0


def test_rt3d_ragged_and_large_clouds(sm3, oracle, synth):
    """Point counts around the 64-point LDS chunk of the scorer, and a cloud of several
    thousand points."""
    grid, world = synth.make_submap_3d(5, 0.1, (8.0, 8.0, 4.0), 4, 8, 96)
    vox = grid.voxels()
    pos = world.free_position(6, 0.5)
    full = world.scan(pos, 0.3, 16, 256, seed=9)
    assert full.shape[0] > 3000
    q = quat_from_angle_axis(0.31, [0, 0, 1])
    init = list(pos + np.array([0.03, -0.02, 0.01])) + q
    for n in (63, 64, 65, 127, 3001):
        cloud = full[:n]
        ref = oracle.rt3d_match(0.1, vox, init, cloud, 0.1, math.radians(1.0), 0.1, 0.1)
        m = sm3.RealTimeCorrelativeScanMatcher3D(0.1, math.radians(1.0), 0.1, 0.1)
        score, pose = m.match(sm3.Rigid3d(tuple(init[:3]), tuple(init[3:])), cloud, 0.1, vox)
        assert m.last_stats["candidates_scored"] == ref["num_candidates"]
        assert np.float32(score) == np.float32(ref["score"]), n
        np.testing.assert_array_equal(_pose7(pose), ref["pose"])


# ----------------------------------------------------------------------------
# RT-3D integer bulk pass (round 2): same result as the per-candidate kernel and the oracle
# ----------------------------------------------------------------------------
@pytest.mark.parametrize("case", ["yaw", "tilted", "far_points", "offset_box", "weights"])
def test_rt3d_bulk_pass_equals_per_candidate_kernel(sm3, oracle, synth, case, debug):
    """The bulk pass only selects finalists; score and pose come from the reference's own
    arithmetic.  Tilted initial orientations (the translation lattice is then not aligned
    with the voxel axes), points far outside the stored box (clamped while staged), a box away
    from the origin and strong weights (finalists not at the unweighted maximum)."""
    grid, world = synth.make_submap_3d(11, 0.1, (8.0, 8.0, 4.0), 4, 8, 96)
    vox = grid.voxels().copy()
    pos = world.free_position(6, 0.5)
    cloud = world.scan(pos, 0.3, 16, 256, seed=9)[:2500].copy()
    axis, angle = [0, 0, 1], 0.31
    shift = np.zeros(3)
    weights = (0.1, 0.1)
    if case == "tilted":
        axis, angle = [0.3, -0.5, 0.8], 0.45
    if case == "far_points":
        # tens of metres outside the grid (not further: the angular step shrinks with the
        # longest range and the oracle has to finish in seconds)
        cloud[::7] += np.array([21.0, -17.0, 9.0], np.float32)
        cloud[5::11] += np.float32(3.0)
    if case == "offset_box":
        vox["x"] += 700
        vox["y"] -= 350
        vox["z"] += 90
        shift = np.array([70.0, -35.0, 9.0])
    if case == "weights":
        weights = (4.0, 6.0)
    q = quat_from_angle_axis(angle, axis)
    init = list(pos + shift + np.array([0.13, -0.08, 0.06])) + q
    ref = oracle.rt3d_match(0.1, vox, init, cloud, 0.2, math.radians(1.0), *weights)
    m = sm3.RealTimeCorrelativeScanMatcher3D(0.2, math.radians(1.0), *weights)
    rigid = sm3.Rigid3d(tuple(init[:3]), tuple(init[3:]))
    got = {}
    from cartographer_amd import _lib
    # "tiles": the LDS-tiled bulk passes, cross-checked element by element against the gather
    # kernels (rt3d_crosscheck: every group bound bitwise, every candidate sum); "1": the
    # gather kernels alone; "0": every candidate scored exhaustively
    # "fixed": the tiled passes as shipped (group centres in packed fixed-point, a different but
    # equally valid centre cell next to boundaries: checked with EVERY group expanded, so that
    # rt3d_verify compares every group bound with every member's own bounds)
    # "staged": the shipped configuration under rt3d_verify (second candidate round by point
    # segments: every intermediate bound checked against the candidate's final sum);
    # "shipped": the same without the verification mode (candidates really leave the lists)
    # Since round 4 "fixed" / "staged" / "shipped" include the rotation-block level above the
    # group pass ("fixed": every (rotation, group) pair computed and every block bound checked
    # against all of its pairs' bounds; "staged": the pairs the thresholds keep); "dense": the
    # group pass over all pairs as before; "best_block": the first group round takes only the
    # pairs of the best block.
    for bulk in ("tiles", "fixed", "staged", "shipped", "dense", "best_block", "1", "0"):
        _lib.debug_reset()
        tiled = bulk in ("tiles", "fixed", "staged", "shipped", "dense", "best_block")
        debug(rt3d_verify=0 if bulk == "shipped" else 1,     # group bounds checked against member bounds
              rt3d_legacy=1 if bulk == "0" else 0,
              rt3d_no_tiles=0 if tiled else 1,
              rt3d_crosscheck=1 if bulk == "tiles" else 0,
              rt3d_expand_all=1 if bulk == "fixed" else 0,
              rt3d_no_rotblocks=1 if bulk == "dense" else 0,
              rt3d_rotblock_permille=1000 if bulk == "best_block" else 0)
        score, pose = m.match(rigid, cloud, 0.1, vox)
        got[bulk] = (np.float32(score), _pose7(pose), dict(m.last_stats))
        assert m.last_stats["candidates_scored"] == ref["num_candidates"]
        assert np.float32(score) == np.float32(ref["score"]), (case, bulk)
        np.testing.assert_array_equal(_pose7(pose), ref["pose"])
    # the bulk pass ran (it reports its finalists) and narrowed the search down
    assert 1 <= got["1"][2]["nodes_expanded"] <= 4096
    assert 1 <= got["tiles"][2]["nodes_expanded"] <= 4096
    assert 1 <= got["fixed"][2]["nodes_expanded"] <= 4096
    assert got["0"][2]["nodes_expanded"] == 0
    # same group bounds, same candidate sums: the two bulk paths evaluate (almost) the same
    # number of bounds (the ambiguity counts group lookups differently, which moves the second
    # round's threshold by rounding-level amounts)
    a, b = got["tiles"][2]["coarse_candidates"], got["1"][2]["coarse_candidates"]
    assert abs(a - b) <= 0.02 * b


def test_rt3d_bulk_pass_flat_landscape_falls_back(sm3, oracle):
    """An empty grid scores every candidate alike: more finalists than the list holds, the
    per-candidate kernel takes over and the first candidate wins as in the reference."""
    from cartographer_amd._lib import VOXEL_DTYPE
    empty = np.zeros(0, VOXEL_DTYPE)
    rng = np.random.default_rng(3)
    cloud = rng.uniform(-2, 2, (300, 3)).astype(np.float32)
    ref = oracle.rt3d_match(0.1, empty, [0, 0, 0, 1, 0, 0, 0], cloud, 0.5, 0.02, 0.0, 0.0)
    m = sm3.RealTimeCorrelativeScanMatcher3D(0.5, 0.02, 0.0, 0.0)
    score, pose = m.match(sm3.Rigid3d(), cloud, 0.1, empty)
    assert np.float32(score) == np.float32(ref["score"])
    np.testing.assert_array_equal(_pose7(pose), ref["pose"])


# ----------------------------------------------------------------------------
# fast-3D device batch (round 2): one chain of launches for all pairs of a node
# ----------------------------------------------------------------------------
def _fast3d_batch_scene(sm3, synth, depths):
    hist = np.zeros(16, np.float32)
    matchers, worlds = [], []
    for k, depth in enumerate(depths):
        opt = dict(branch_and_bound_depth=depth, full_resolution_depth=2, min_rotational_score=0.0,
                   min_low_resolution_score=0.2, linear_xy_search_window=1.0,
                   linear_z_search_window=0.4, angular_search_window=math.radians(10.0))
        grid, world = synth.make_submap_3d(70 + k % 3, 0.2, (8.0, 8.0, 3.0), 4, 8, 96)
        vox = grid.voxels()
        matchers.append(sm3.FastCorrelativeScanMatcher3D(0.2, vox, grid.grid_size, 0.2, vox, hist,
                                                         **opt))
        worlds.append(world)
    pos = worlds[0].free_position(200, 0.6)
    hi = worlds[0].scan(pos, 0.0, 6, 64, seed=0)
    data = sm3.TrajectoryNodeData(hi, hi[::5].copy(), hist,
                                  tuple(quat_from_angle_axis(0.01, [1, 0, 0])))
    return matchers, pos, data


def _assert_same_results(expected, got):
    for k, (e, g) in enumerate(zip(expected, got)):
        assert (e is None) == (g is None), (k, e, g)
        if e is None:
            continue
        for key in ("score", "rotational_score", "low_resolution_score"):
            assert np.float32(e[key]) == np.float32(g[key]), (k, key, e, g)
        assert e["pose_estimate"] == g["pose_estimate"], (k, e, g)


@pytest.mark.parametrize("capacity", [None, "4096"])
def test_fast3d_device_batch_mixed_depths_poses_and_overflow(sm3, synth, oracle, debug,
                                                            capacity):
    """Twelve pairs in one chain of launches: stacks of different depth (3 .. 6, one of depth 1:
    that batch falls back to single searches), a different node pose per pair, windowed and
    full-submap searches mixed -- pair by pair what the single calls return.  With the frontier
    shrunk to 4096 nodes the shared lists overflow and every pair is repeated on its own
    (strict retry inside)."""
    depths = [5, 4, 6, 3, 5, 4, 6, 3, 5, 5, 4, 6]
    matchers, pos, data = _fast3d_batch_scene(sm3, synth, depths)
    ident = sm3.Rigid3d()
    rng = np.random.default_rng(5)
    nodes, fulls, thresholds = [], [], []
    for k in range(len(depths)):
        d = rng.uniform(-0.3, 0.3, 3) * np.array([1.0, 1.0, 0.3])
        nodes.append(sm3.Rigid3d(tuple(pos + d),
                                 tuple(quat_from_angle_axis(rng.uniform(-0.1, 0.1), [0, 0, 1]))))
        fulls.append(k % 5 == 3)
        thresholds.append([0.12, 0.3, 0.99][k % 3] if k % 4 else 0.12)
    expected = []
    for m, node, full, t in zip(matchers, nodes, fulls, thresholds):
        expected.append(m.match_full_submap(node.rotation, ident.rotation, data, t) if full
                        else m.match(node, ident, data, t))
    assert any(e is not None for e in expected) and any(e is None for e in expected)
    if capacity:
        debug(frontier_capacity=int(capacity))
    got, stats = sm3.fast3d_match_batch(matchers, nodes, [ident] * len(depths), fulls, thresholds,
                                        data)
    _assert_same_results(expected, got)
    assert stats["num_scans"] > 0
    if capacity is None:
        # a depth-1 stack in the batch: leaf verification path, searched one by one
        matchers1, _, _ = _fast3d_batch_scene(sm3, synth, [1, 4])
        exp1 = [m.match(nodes[0], ident, data, 0.12) for m in matchers1]
        got1, _ = sm3.fast3d_match_batch(matchers1, [nodes[0]] * 2, [ident] * 2, [False] * 2,
                                         [0.12] * 2, data)
        _assert_same_results(exp1, got1)


def test_fast3d_repeated_searches_return_the_same_result(sm3, synth):
    """Regression for a round-1 race: every thread of an expansion block read the moving bound
    for itself, so part of a block could skip a node the rest expanded -- about one search in
    fifty returned a leaf that does not exist (tools/stress_fast3d.py).  Weak matches with low
    thresholds (many nodes alive while the bound rises) repeated: single and batched searches
    must return the same thing every time."""
    matchers, pos, data = _fast3d_batch_scene(sm3, synth, [5, 5, 5])
    node = sm3.Rigid3d(tuple(pos + np.array([0.3, -0.2, 0.1])),
                       tuple(quat_from_angle_axis(0.05, [0, 0, 1])))
    ident = sm3.Rigid3d()
    pairs = [(0, False, 0.12), (1, False, 0.12), (0, True, 0.12), (1, True, 0.3), (2, False, 0.12)]

    def key(r):
        return None if r is None else (np.float32(r["score"]),
                                       np.float32(r["low_resolution_score"]),
                                       tuple(r["pose_estimate"].translation))

    def single():
        return [key(matchers[k].match_full_submap(node.rotation, ident.rotation, data, t) if full
                    else matchers[k].match(node, ident, data, t)) for k, full, t in pairs]

    def batch():
        got, _ = sm3.fast3d_match_batch([matchers[k] for k, _, _ in pairs], [node] * len(pairs),
                                        [ident] * len(pairs), [f for _, f, _ in pairs],
                                        [t for _, _, t in pairs], data)
        return [key(g) for g in got]

    first = single()
    assert any(f is not None for f in first)
    for r in range(80):
        assert single() == first, r
        assert batch() == first, r


# ----------------------------------------------------------------------------
# CeresScanMatcher3D on the device vs the oracle's restatement
# ----------------------------------------------------------------------------
def _ceres3d_compare(sm3, oracle, pairs, target, init, weights, tw, rw, **kw):
    ref = oracle.ceres3d_match(pairs, target, init, weights, translation_weight=tw,
                               rotation_weight=rw, **kw)
    m = sm3.CeresScanMatcher3D(weights, tw, rw, **kw)
    pose, summary = m.match(target, sm3.Rigid3d(tuple(init[:3]), tuple(init[3:])), pairs)
    # f64 on both sides; the device reduces its sums in a fixed but different order, and its
    # sin / cos / pow are not libm's
    np.testing.assert_allclose(_pose7(pose), ref["pose"], rtol=0, atol=1e-6)
    assert abs(summary["initial_cost"] - ref["initial_cost"]) <= 1e-9 * max(1.0, ref["initial_cost"])
    assert abs(summary["final_cost"] - ref["final_cost"]) <= 1e-6 * max(1.0, ref["final_cost"])
    assert summary["num_successful_steps"] == ref["num_successful_steps"]
    assert summary["termination"] == ref["termination"]
    return ref, pose, summary


@pytest.mark.parametrize("t", [(-1.0, 0.0, 0.0), (-0.8, 0.0, 0.0), (-1.0, 0.0, -0.2),
                               (-0.9, -0.2, 0.2)])
def test_ceres3d_reference_fixture(sm3, oracle, synth, t):
    """CeresScanMatcher3DTest (ceres_scan_matcher_3d_test.cc:36-111, without the intensity
    term): the reference's expectations and the oracle's iterates."""
    from test_ceres_3d import POINTS, fixture, is_nearly
    vox = fixture(synth, POINTS)
    init = list(t) + [1.0, 0.0, 0.0, 0.0]
    ref, pose, summary = _ceres3d_compare(sm3, oracle, [(POINTS, 1.0, vox)], t, init, [1.0], 0.01,
                                          0.1, use_nonmonotonic_steps=True, max_num_iterations=10)
    assert abs(summary["final_cost"]) < 1e-2
    assert is_nearly(_pose7(pose), [-1.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0], 3e-2)


@pytest.mark.parametrize("yaw_only,nonmono,seed", [(False, False, 7), (True, False, 7),
                                                   (False, True, 9), (True, True, 11)])
def test_ceres3d_equals_the_oracle(sm3, oracle, synth, yaw_only, nonmono, seed):
    """Two (cloud, grid) pairs like the local trajectory builder's (high and low resolution),
    tilted initial rotation, both parameterizations, monotonic and non-monotonic steps."""
    grid, world = synth.make_submap_3d(seed, 0.1, (8.0, 8.0, 4.0), 4, 8, 96)
    low, _ = synth.make_submap_3d(seed, 0.3, (8.0, 8.0, 4.0), 4, 8, 96)
    pos = world.free_position(seed + 1, 0.5)
    yaw = 0.4
    full = world.scan(pos, yaw, 16, 128, seed=3)
    hi, lo = full[::2].copy(), full[::9].copy()
    pairs = [(hi, 0.1, grid.voxels()), (lo, 0.3, low.voxels())]
    init_t = pos + np.array([0.04, -0.03, 0.02])
    init = list(init_t) + quat_from_angle_axis(yaw + 0.02, [0.05, -0.02, 1.0])
    ref, pose, summary = _ceres3d_compare(sm3, oracle, pairs, init_t, init, [1.0, 6.0], 5.0, 4e2,
                                          only_optimize_yaw=yaw_only,
                                          use_nonmonotonic_steps=nonmono, max_num_iterations=12)
    assert summary["final_cost"] < summary["initial_cost"]


@pytest.mark.parametrize("huber_scale,yaw_only", [(0.3, False), (1e6, False), (0.3, True)])
def test_ceres3d_with_the_intensity_cost_function_equals_the_oracle(sm3, oracle, synth,
                                                                    huber_scale, yaw_only):
    """IntensityCostFunction3D under ceres::HuberLoss (ceres_scan_matcher_3d.cc:118-137,
    intensity_cost_function_3d.h:37-91) on the device: the first pair carries an
    intensity_hybrid_grid, huber_scale 0.3 puts its block in the outlier region (rho' < 1) and
    1e6 leaves it quadratic; cells with count 0 and points above the threshold included.  The
    oracle side of this is pinned on the reference's own compiled sources
    (tests/test_reference_ref_ceres.py)."""
    from test_reference_ref_ceres import _intensity_world
    vox, iv, pos, cloud, intensities = _intensity_world(synth, oracle, 7)
    low, _ = synth.make_submap_3d(7, 0.3, (8.0, 8.0, 4.0), 4, 8, 96)
    pairs = [(cloud, 0.1, vox, intensities, iv, (0.5, huber_scale, 100.0)),
             (cloud[::4].copy(), 0.3, low.voxels())]
    init_t = np.array([0.04, -0.03, 0.02])
    init = list(init_t) + quat_from_angle_axis(0.02, [0.05, -0.02, 1.0])
    kw = dict(only_optimize_yaw=yaw_only, max_num_iterations=12)
    ref = oracle.ceres3d_match_intensity(pairs, init_t, init, [1.0, 6.0], translation_weight=5.0,
                                         rotation_weight=4e2, **kw)
    m = sm3.CeresScanMatcher3D([1.0, 6.0], 5.0, 4e2, **kw)
    pose, summary = m.match(init_t, sm3.Rigid3d(tuple(init[:3]), tuple(init[3:])), pairs)
    np.testing.assert_allclose(_pose7(pose), ref["pose"], rtol=0, atol=1e-6)
    assert abs(summary["initial_cost"] - ref["initial_cost"]) <= 1e-9 * max(1.0, ref["initial_cost"])
    assert abs(summary["final_cost"] - ref["final_cost"]) <= 1e-6 * max(1.0, ref["final_cost"])
    assert summary["num_successful_steps"] == ref["num_successful_steps"]
    assert summary["termination"] == ref["termination"]
    # the block is there: the match without it ends elsewhere
    plain, _ = m.match(init_t, sm3.Rigid3d(tuple(init[:3]), tuple(init[3:])),
                       [p[:3] for p in pairs])
    assert np.abs(np.array(_pose7(plain)) - ref["pose"]).max() > 1e-6


def test_ceres3d_intensity_options_are_checked(sm3, oracle, synth):
    from cartographer_amd._lib import CmxError, INVALID_ARGUMENT
    from test_reference_ref_ceres import _intensity_world
    vox, iv, pos, cloud, intensities = _intensity_world(synth, oracle, 7)
    m = sm3.CeresScanMatcher3D([1.0], 5.0, 4e2)
    for bad in ((0.0, 0.3, 100.0), (0.5, 0.0, 100.0), (0.5, 0.3, 0.0)):
        with pytest.raises(CmxError) as e:      # CHECK_GT(weight / huber_scale / threshold, 0)
            m.match((0, 0, 0), sm3.Rigid3d(), [(cloud, 0.1, vox, intensities, iv, bad)])
        assert e.value.status == INVALID_ARGUMENT


def test_ceres3d_invalid_arguments(sm3, synth):
    from cartographer_amd._lib import CmxError, INVALID_ARGUMENT
    from test_ceres_3d import POINTS, fixture
    vox = fixture(synth, POINTS)
    with pytest.raises(CmxError) as e:      # CHECK_GT(options_.translation_weight(), 0.)
        sm3.CeresScanMatcher3D([1.0], 0.0, 0.1).match((0, 0, 0), sm3.Rigid3d(),
                                                      [(POINTS, 1.0, vox)])
    assert e.value.status == INVALID_ARGUMENT
    with pytest.raises(CmxError) as e:      # CHECK_GT(options_.occupied_space_weight(i), 0.)
        sm3.CeresScanMatcher3D([0.0], 0.01, 0.1).match((0, 0, 0), sm3.Rigid3d(),
                                                       [(POINTS, 1.0, vox)])
    assert e.value.status == INVALID_ARGUMENT


@pytest.mark.parametrize("oct", ["1", "0"])
def test_fast3d_oct_layout_equals_the_oracle(sm3, oracle, synth, debug, oct):
    """Child cells from the 8-byte "oct" words (one gather per point and node) and from the level
    bricks themselves (debug switch fast3d_no_oct at matcher creation): the oracle's result either way, on
    the C5-shaped submap (full and half resolution levels, strides 1, 2, 4) and on a shallow
    stack with a window reaching far outside the grid."""
    debug(fast3d_no_oct=1 if oct == "0" else 0)
    size = (15.0, 15.0, 7.5)
    grid, world = synth.make_submap_3d(42, 0.1, size, 8, 32, 512)
    low, _ = synth.make_submap_3d(42, 0.45, size, 8, 32, 512)
    vox, low_vox = grid.voxels(), low.voxels()
    rng = np.random.default_rng(1)
    hist = rng.uniform(0.0, 1.0, 120).astype(np.float32)
    hist[10:14] += 6.0
    pos = world.free_position(77, 0.6)
    full = world.scan(pos, 0.4, 32, 512, seed=1)
    hi, lo = full[::6].copy(), full[::80].copy()
    scan_hist = np.roll(hist, -19).copy()
    opt = dict(branch_and_bound_depth=8, full_resolution_depth=3, min_rotational_score=0.77,
               min_low_resolution_score=0.35, linear_xy_search_window=5.0,
               linear_z_search_window=1.0, angular_search_window=math.radians(15.0))
    om, gm = _both(sm3, oracle, 0.1, vox, grid.grid_size, 0.45, low_vox, hist, **opt)
    node = [pos[0] + 0.8, pos[1] - 0.6, pos[2] + 0.2] + quat_from_angle_axis(0.5, [0, 0, 1])
    ident = [0, 0, 0, 1, 0, 0, 0]
    ref = om.match(node, ident, [1, 0, 0, 0], hi, lo, scan_hist, 0.2)
    got = gm.match(sm3.Rigid3d(tuple(node[:3]), tuple(node[3:])), sm3.Rigid3d(),
                   sm3.TrajectoryNodeData(hi, lo, scan_hist), 0.2)
    _assert_result(ref, got)
    # shallow stack, full-submap search (window beyond the grid on every side)
    g2, w2 = synth.make_submap_3d(5, 0.2, (6.0, 6.0, 3.0), 4, 8, 96)
    v2 = g2.voxels()
    h2 = np.zeros(8, np.float32)
    opt2 = dict(branch_and_bound_depth=4, full_resolution_depth=2, min_rotational_score=0.0,
                min_low_resolution_score=0.1, linear_xy_search_window=1.0,
                linear_z_search_window=0.4, angular_search_window=math.radians(10.0))
    om2, gm2 = _both(sm3, oracle, 0.2, v2, g2.grid_size, 0.2, v2, h2, **opt2)
    p2 = w2.free_position(9, 0.5)
    c2 = w2.scan(p2, 0.0, 6, 64, seed=2)
    ref2 = om2.match_full_submap([1, 0, 0, 0], [1, 0, 0, 0], [1, 0, 0, 0], c2, c2[::4].copy(), h2, 0.1)
    got2 = gm2.match_full_submap((1.0, 0.0, 0.0, 0.0), (1.0, 0.0, 0.0, 0.0),
                                 sm3.TrajectoryNodeData(c2, c2[::4].copy(), h2), 0.1)
    _assert_result(ref2, got2)


def test_fast3d_refine_batch_equals_single_ceres_matches(sm3, oracle, synth):
    """cmx_fast3d_refine_batch (ConstraintBuilder3D's ceres_scan_matcher_.Match after the search,
    constraint_builder_3d.cc:263-276) against grids resident in HBM: entry by entry what
    cmx_ceres3d_match returns for the same voxel lists (bit for bit: same kernel, same bricks),
    which in turn equals the oracle; not-found entries pass through."""
    hist = np.zeros(16, np.float32)
    opt = dict(branch_and_bound_depth=4, full_resolution_depth=2, min_rotational_score=0.0,
               min_low_resolution_score=0.2, linear_xy_search_window=1.0,
               linear_z_search_window=0.4, angular_search_window=math.radians(10.0))
    matchers, grids = [], []
    for seed in (70, 71, 70):
        grid, world = synth.make_submap_3d(seed, 0.1, (8.0, 8.0, 3.0), 4, 8, 96)
        low, _ = synth.make_submap_3d(seed, 0.3, (8.0, 8.0, 3.0), 4, 8, 96)
        matchers.append(sm3.FastCorrelativeScanMatcher3D(0.1, grid.voxels(), grid.grid_size, 0.3,
                                                         low.voxels(), hist, **opt))
        grids.append((grid.voxels(), low.voxels(), world))
    world = grids[0][2]
    pos = world.free_position(200, 0.6)
    full = world.scan(pos, 0.2, 8, 96, seed=0)
    data = sm3.TrajectoryNodeData(full[::2].copy(), full[::7].copy(), hist)
    rng = np.random.default_rng(3)
    poses = []
    for k in range(3):
        t = pos + rng.uniform(-0.05, 0.05, 3)
        poses.append(sm3.Rigid3d(tuple(t), tuple(quat_from_angle_axis(0.2 + rng.uniform(-0.02, 0.02),
                                                                       [0.02, -0.01, 1.0]))))
    found = [True, False, True]
    ceres = sm3.CeresScanMatcher3D([5.0, 20.0], 10.0, 1.0, only_optimize_yaw=False,
                                   use_nonmonotonic_steps=False, max_num_iterations=10)
    refined, summaries = ceres.refine_batch(matchers, found, poses, data)
    assert refined[1] == poses[1] and summaries[1]["num_successful_steps"] == 0
    for k in (0, 2):
        pairs = [(data.high_resolution_point_cloud, 0.1, grids[k][0]),
                 (data.low_resolution_point_cloud, 0.3, grids[k][1])]
        single, s1 = ceres.match(poses[k].translation, poses[k], pairs)
        assert refined[k] == single, (k, refined[k], single)
        assert summaries[k] == s1
        init = list(poses[k].translation) + list(poses[k].rotation)
        ref = oracle.ceres3d_match(pairs, poses[k].translation, init, [5.0, 20.0],
                                   translation_weight=10.0, rotation_weight=1.0,
                                   max_num_iterations=10)
        np.testing.assert_allclose(_pose7(refined[k]), ref["pose"], rtol=0, atol=1e-6)
        assert summaries[k]["final_cost"] <= summaries[k]["initial_cost"]


def test_constraint_builder_3d_refines_on_the_device(sm3, oracle, synth):
    """ConstraintBuilder3D with its ceres_scan_matcher_: the constraint transform is the refined
    pose (constraint_builder_3d.cc:263-281), equal to refining the search result separately."""
    from cartographer_amd import constraint_builder as cb
    hist = np.zeros(16, np.float32)
    grid, world = synth.make_submap_3d(70, 0.1, (8.0, 8.0, 3.0), 4, 8, 96)
    low, _ = synth.make_submap_3d(70, 0.3, (8.0, 8.0, 3.0), 4, 8, 96)
    pos = world.free_position(200, 0.6)
    full = world.scan(pos, 0.0, 8, 96, seed=0)
    data = sm3.TrajectoryNodeData(full[::2].copy(), full[::7].copy(), hist)
    options = cb.ConstraintBuilderOptions3D(
        sampling_ratio=1.0, max_constraint_distance=50.0, min_score=0.12,
        global_localization_min_score=0.12, branch_and_bound_depth=4, full_resolution_depth=2,
        min_rotational_score=0.0, min_low_resolution_score=0.1, linear_xy_search_window=0.5,
        linear_z_search_window=0.2, angular_search_window=math.radians(5.0))
    ceres = sm3.CeresScanMatcher3D([5.0, 20.0], 10.0, 1.0, max_num_iterations=10)
    submap = cb.Submap3D(0.1, grid.voxels(), grid.grid_size, 0.3, low.voxels(), hist)
    node = sm3.Rigid3d(tuple(pos + np.array([0.1, -0.1, 0.05])), (1.0, 0.0, 0.0, 0.0))
    plain = cb.ConstraintBuilder3D(options)
    with_ceres = cb.ConstraintBuilder3D(options, ceres=ceres)
    out = {}
    for name, builder in (("plain", plain), ("ceres", with_ceres)):
        builder.maybe_add_constraint((0, 0), submap, (0, 5), data, node, sm3.Rigid3d())
        builder.notify_end_of_node()
        builder.when_done(lambda constraints, name=name: out.__setitem__(name, constraints))
    assert len(out["plain"]) == 1 and len(out["ceres"]) == 1
    searched, refined = out["plain"][0].zbar_ij, out["ceres"][0].zbar_ij
    pairs = [(data.high_resolution_point_cloud, 0.1, grid.voxels()),
             (data.low_resolution_point_cloud, 0.3, low.voxels())]
    single, _ = ceres.match(searched.translation, searched, pairs)
    assert refined == single
    assert refined != searched
    assert with_ceres.last_refine_summaries[0]["final_cost"] < \
        with_ceres.last_refine_summaries[0]["initial_cost"]
    # ... and what the oracle's CeresScanMatcher3D makes of the same search result (wiring of the
    # refine target = initial pose = search result, constraint_builder_3d.cc:263-270)
    init7 = list(searched.translation) + list(searched.rotation)
    ref = oracle.ceres3d_match(pairs, searched.translation, init7, [5.0, 20.0],
                               translation_weight=10.0, rotation_weight=1.0, max_num_iterations=10)
    np.testing.assert_allclose(_pose7(refined), ref["pose"], rtol=0, atol=1e-6)
    # geometry: the scan was taken at `pos` with identity rotation; the search result is quantised
    # to the 0.1 m voxel lattice around the (offset) initial pose and must lie within its reach.
    # No such bound holds for the REFINED pose here: on this sparse synthetic submap (8 scans;
    # ~17 % of the scan's returns land in a known cell at the true pose, whose score ~0.20 is why
    # min_score is 0.12 and not round 1's 0.2, which sat on the last bit of the score) the
    # least-squares optimum of the reference's own cost lies 0.13 m from the truth -- the oracle's
    # CeresScanMatcher3D goes there too, which is what the equality above guards.
    err_searched = np.linalg.norm(np.array(searched.translation) - pos)
    assert err_searched < 0.1 * math.sqrt(3.0)
    assert np.linalg.norm(np.array(refined.translation) - np.array(searched.translation)) < 0.3


@pytest.mark.parametrize("families", [None, "0"])
@pytest.mark.parametrize("affinity", [None, "0"])
def test_fast3d_device_batch_of_twenty_pairs(sm3, synth, debug, affinity, families):
    """From 16 pairs on a problem's nodes stay on one XCD (placement only: the search is
    order-free); pair by pair the batch must return what the single searches return."""
    if affinity is not None:
        debug(fast3d_affinity=1)                       # nodes of a problem on any XCD
    if families is not None:         # every node expanded on its own (round 2) instead of by family
        debug(fast3d_no_families=1)
    depths = [5, 4, 6, 3] * 5
    matchers, pos, data = _fast3d_batch_scene(sm3, synth, depths)
    ident = sm3.Rigid3d()
    rng = np.random.default_rng(11)
    nodes, fulls, thresholds = [], [], []
    for k in range(len(depths)):
        d = rng.uniform(-0.3, 0.3, 3) * np.array([1.0, 1.0, 0.3])
        nodes.append(sm3.Rigid3d(tuple(pos + d),
                                 tuple(quat_from_angle_axis(rng.uniform(-0.1, 0.1), [0, 0, 1]))))
        fulls.append(k % 7 == 3)
        thresholds.append([0.12, 0.3, 0.99][k % 3] if k % 4 else 0.12)
    expected = [m.match_full_submap(node.rotation, ident.rotation, data, t) if full
                else m.match(node, ident, data, t)
                for m, node, full, t in zip(matchers, nodes, fulls, thresholds)]
    assert sum(e is not None for e in expected) >= 5
    got, stats = sm3.fast3d_match_batch(matchers, nodes, [ident] * len(depths), fulls, thresholds,
                                        data)
    _assert_same_results(expected, got)
    assert stats["expansion_launches"] >= 1 and stats["expansion_nodes"] > 0


# ----------------------------------------------------------------------------
# cmx_fast3d_match_sharded: the C5 fan-out over a communicator (north star: 256 submaps / 8 GPUs)
# ----------------------------------------------------------------------------
@pytest.mark.parametrize("virtual_ranks", [0, 2, "rccl"])
def test_fast3d_sharded_match_equals_the_batch(sm3, synth, debug, virtual_ranks):
    """cmx_fast3d_match_sharded over a communicator of every visible device (one on the test
    box): pair by pair what cmx_fast3d_match_batch returns -- windowed and full-submap pairs
    mixed, thresholds that reject some pairs -- and the RCCL all-reduce(max) of the packed key
    returns the best found pair, the LOWEST index among equal scores (matchers 0, 3, 6 and 9
    are built from the same submap and see the same node: ties)."""
    import torch
    from cartographer_amd import sharding
    ndev = torch.cuda.device_count()
    if virtual_ranks == "rccl":
        # the one device through RCCL itself (debug switch comm_force_rccl: ncclCommInitAll with
        # one device, the key through a grouped ncclAllReduce(int64, max)): the binding executed
        debug(comm_force_rccl=1)
        comm = sharding.Communicator([0])
        assert comm.uses_rccl and comm.num_devices == 1
        ndev = 1
    elif virtual_ranks:        # the one device as two ranks (see the 2D test of the same name)
        debug(comm_virtual_ranks=virtual_ranks)
        comm = sharding.Communicator([0])
        assert comm.num_devices == virtual_ranks
        ndev = 1
    else:
        comm = sharding.Communicator(list(range(ndev)))
    depths = [5, 4, 6] * 4                      # seeds 70 + k % 3: k = 0, 3, 6, 9 identical
    matchers, pos, data = _fast3d_batch_scene(sm3, synth, depths)
    if ndev > 1:                                # place every matcher on the device that owns it
        pytest.skip("multi-device placement is exercised by the driver's scaling run")
    ident = sm3.Rigid3d()
    rng = np.random.default_rng(3)
    node0 = sm3.Rigid3d(tuple(pos + np.array([0.2, -0.1, 0.05])),
                        tuple(quat_from_angle_axis(0.04, [0, 0, 1])))
    nodes, fulls, thresholds = [], [], []
    for k in range(len(depths)):
        if k % 3 == 0:
            nodes.append(node0)
            fulls.append(False)
            thresholds.append(0.12)
        else:
            d = rng.uniform(-0.3, 0.3, 3) * np.array([1.0, 1.0, 0.3])
            nodes.append(sm3.Rigid3d(tuple(pos + d), tuple(quat_from_angle_axis(
                rng.uniform(-0.1, 0.1), [0, 0, 1]))))
            fulls.append(k % 5 == 2)
            thresholds.append(0.99 if k % 4 == 1 else 0.12)
    expected, stats1 = sm3.fast3d_match_batch(matchers, nodes, [ident] * len(depths), fulls,
                                              thresholds, data)
    got, best, stats = comm.match_batch_3d(matchers, nodes, [ident] * len(depths), fulls,
                                           thresholds, data)
    _assert_same_results(expected, got)
    assert any(g is None for g in got) and sum(g is not None for g in got) >= 4
    # (the number of nodes a branch-and-bound search expands depends on when the bound rises:
    # candidates_scored differs by a few hundred of 7.6 M between two runs of the same batch)
    assert abs(stats["candidates_scored"] - stats1["candidates_scored"]) < \
        0.01 * stats1["candidates_scored"]
    scores = np.array([-1.0 if g is None else np.float32(g["score"]) for g in got], np.float32)
    assert got[0] is not None and scores[0] == scores[3] == scores[6] == scores[9]
    assert best[0] == int(np.argmax(scores)) and np.float32(best[1]) == scores.max()
    # nothing found anywhere: the sentinel
    none, best_none, _ = comm.match_batch_3d(matchers[:3], nodes[:3], [ident] * 3, [False] * 3,
                                             [0.999] * 3, data)
    assert all(g is None for g in none) and best_none[0] == -1
