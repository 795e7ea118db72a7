"""GPU parity tests of the 2D matchers: the HIP path (through the C ABI) against
the CPU oracle on identical scans + grids.

Bars: integer work (precomputation grids, discretised scans, search bounds,
candidate sums) is bit-exact; returned f32 scores are bit-equal; poses are
equal to 1e-12 (they are computed from integer offsets in f64), also where
several leaves share the best score: the device path reproduces the order in
which the reference's depth-first search (and its std::sort) meets them.
"""
import math
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sm():
    from cartographer_amd import _lib, scan_matching
    assert _lib.lib().cmx_device_count() >= 1, "no HIP device: these tests need the GPU"
    return scan_matching


def _grid(sm, cells, lim):
    return sm.Grid2D(cells, lim["resolution"], lim["max_x"], lim["max_y"])


def _leaf_sum(level0, scans, bounds_unused, scan, dx, dy):
    x = scans[scan, :, 0] + dx
    y = scans[scan, :, 1] + dy
    ok = (x >= 0) & (y >= 0) & (x < level0.shape[1]) & (y < level0.shape[0])
    return int(level0[y[ok], x[ok]].astype(np.int64).sum())


def _assert_match_parity(oracle_m, gpu_m, res, init, cloud, min_score, full, sm):
    if full:
        ref = oracle_m.match_full_submap(cloud, min_score)
        found, score, pose = gpu_m.match_full_submap(cloud, min_score)
    else:
        ref = oracle_m.match(init, cloud, min_score)
        found, score, pose = gpu_m.match(sm.Rigid2d(*init), cloud, min_score)
    assert found == ref["found"]
    if not found:
        return ref, None
    assert np.float32(score) == np.float32(ref["score"]), (score, ref["score"])
    # Ties included: the device path reproduces the reference's depth-first /
    # std::sort order among equal-score leaves, so the pose is always the same.
    got = np.array([pose.x, pose.y, pose.theta])
    np.testing.assert_allclose(got, ref["pose"], rtol=0, atol=1e-12)
    return ref, gpu_m.last_stats


def oracle_m_center(oracle_m, res):
    return oracle_m._center


def _oracle(oracle, cells, lim, depth, lin=7.0, ang=math.radians(30.0)):
    m = oracle.FastCorrelativeScanMatcher2D(cells, lim["resolution"], lim["max_x"], lim["max_y"],
                                            depth, lin, ang)
    m._center = (lim["max_x"] - 0.5 * lim["resolution"] * lim["num_y_cells"],
                 lim["max_y"] - 0.5 * lim["resolution"] * lim["num_x_cells"])
    return m


# ----------------------------------------------------------------------------
# Precomputation grid stack (K3)
# ----------------------------------------------------------------------------
@pytest.mark.parametrize("nx,ny,depth,seed", [(400, 400, 7, 42), (200, 200, 7, 1), (37, 53, 5, 2),
                                              (1, 1, 3, 3), (5, 3, 6, 4), (130, 64, 8, 5)])
def test_stack_matches_oracle(sm, oracle, synth, nx, ny, depth, seed):
    if nx >= 100:
        cells, lim, _ = synth.make_submap(seed, nx, ny, 0.05, 12, 500, 30.0, 0.01)
    else:
        rng = np.random.default_rng(seed)
        cells = rng.integers(0, 32768, (ny, nx)).astype(np.uint16)
        cells[rng.random((ny, nx)) < 0.3] = 0
        lim = dict(resolution=0.05, max_x=1.0, max_y=2.0, num_x_cells=nx, num_y_cells=ny)
    om = _oracle(oracle, cells, lim, depth)
    gm = sm.FastCorrelativeScanMatcher2D(_grid(sm, cells, lim), depth)
    for level in range(depth):
        np.testing.assert_array_equal(gm.level(level), om.level(level), err_msg=f"level {level}")


def test_stack_over_a_tsdf_cost_range(sm, oracle, synth):
    """cmx_fast2d_create over a Grid2D whose correspondence costs span [-truncation_distance,
    truncation_distance] -- a TSDF2D's tsd plane, fast_correlative_scan_matcher_2d.cc:91-108 takes
    any Grid2D: 1 - |cost| -- every level of the stack equals the oracle's, which
    tests/test_reference_ref.py pins on the reference's PrecomputationGrid2D over its own TSDF2D."""
    from tsdf_helpers import tsdf_from_probability_grid
    cells, lim, _ = synth.make_submap(11, 130, 97, 0.05, 8, 300, 30.0, 0.01)
    tsd, _ = tsdf_from_probability_grid(oracle, cells, 0.05, 0.3, 10.0, 5)
    tsd = tsd.copy()
    tsd[::7, ::5] |= 0x8000
    grid = sm.Grid2D(tsd, 0.05, lim["max_x"], lim["max_y"], -0.3, 0.3)
    gm = sm.FastCorrelativeScanMatcher2D(grid, 6)
    for level in range(6):
        np.testing.assert_array_equal(gm.level(level),
                                      oracle.precompute2d_range(tsd, 1 << level, -0.3, 0.3),
                                      err_msg=f"level {level}")
    values = np.unique(gm.level(0))
    assert values.min() == 0 and 100 < values.max() <= 128 and len(values) > 10   # 1 - |tsd| in [0.7, 1.0] of [0.7, 1.3]


def test_stack_reference_fixture(sm, oracle, synth):
    """PrecomputationGridTest.CorrectValues fixture (fast_..._2d_test.cc:37-57):
    uint8-exact probabilities; widths 1,2,8 are stack levels 0,1,3."""
    from test_oracle_reference_pins import _uint8_grid
    g = _uint8_grid(oracle, synth, 250, 250, 0.05, (5.0, 5.0), ((50, 50), (249, 249)))
    lim = g.limits
    gm = sm.FastCorrelativeScanMatcher2D(_grid(sm, g.cells, lim), 4)
    for level, width in [(0, 1), (1, 2), (3, 8)]:
        np.testing.assert_array_equal(gm.level(level), oracle.precompute2d(g.cells, width))


# ----------------------------------------------------------------------------
# Scan preparation + lowest-resolution sums (K1, a2, K4)
# ----------------------------------------------------------------------------
@pytest.fixture(scope="module")
def c2(synth):
    cells, lim, world = synth.make_submap(42, 400, 400, 0.05, 30, 1000, 30.0, 0.01)
    pose = world.free_pose(1234, 0.5)
    scan = world.scan(pose, 1000, 30.0, 0.01, 7)
    assert scan.shape[0] == 1000
    return cells, lim, world, pose, scan


def test_prepare_full_submap_bit_exact(sm, oracle, c2):
    cells, lim, _, _, scan = c2
    om = _oracle(oracle, cells, lim, 7)
    gm = sm.FastCorrelativeScanMatcher2D(_grid(sm, cells, lim), 7)
    ref = om.prepare([0, 0, 0], scan, True)
    got = gm.debug_prepare(None, scan, True)
    assert got["num_scans"] == ref["num_scans"]
    assert got["step"] == ref["step"]
    np.testing.assert_array_equal(got["scans"], ref["scans"])
    np.testing.assert_array_equal(got["bounds"], ref["bounds"])
    np.testing.assert_array_equal(got["sums"], ref["sums"])


@pytest.mark.parametrize("init", [(0.3, -0.2, 0.4), (-3.0, 2.5, -2.9), (0.0, 0.0, 0.0)])
def test_prepare_windowed_bit_exact(sm, oracle, c2, init):
    cells, lim, _, _, scan = c2
    om = _oracle(oracle, cells, lim, 5, 2.0, math.radians(20.0))
    gm = sm.FastCorrelativeScanMatcher2D(_grid(sm, cells, lim), 5, 2.0, math.radians(20.0))
    ref = om.prepare(list(init), scan, False)
    got = gm.debug_prepare(sm.Rigid2d(*init), scan, False)
    assert got["num_scans"] == ref["num_scans"] and got["step"] == ref["step"]
    np.testing.assert_array_equal(got["scans"], ref["scans"])
    np.testing.assert_array_equal(got["bounds"], ref["bounds"])
    np.testing.assert_array_equal(got["sums"], ref["sums"])


# ----------------------------------------------------------------------------
# Match / MatchFullSubmap
# ----------------------------------------------------------------------------
def test_c2_full_submap_match_parity(sm, oracle, c2):
    cells, lim, _, truth, scan = c2
    om = _oracle(oracle, cells, lim, 7)
    gm = sm.FastCorrelativeScanMatcher2D(_grid(sm, cells, lim), 7)
    ref, stats = _assert_match_parity(om, gm, lim["resolution"], [0, 0, 0], scan, 0.6, True, sm)
    assert ref["found"]
    assert abs(ref["pose"][0] - truth[0]) < 0.1 and abs(ref["pose"][1] - truth[1]) < 0.1
    assert stats["coarse_candidates"] == ref["coarse_candidates"]
    assert stats["num_scans"] == ref["num_scans"]


@pytest.mark.parametrize("seed", [3, 11, 29])
def test_full_submap_other_worlds(sm, oracle, synth, seed):
    cells, lim, world = synth.make_submap(seed, 300, 260, 0.05, 20, 700, 30.0, 0.01)
    scan = world.scan(world.free_pose(seed + 5, 0.5), 777, 30.0, 0.01, seed)
    om = _oracle(oracle, cells, lim, 6)
    gm = sm.FastCorrelativeScanMatcher2D(_grid(sm, cells, lim), 6)
    _assert_match_parity(om, gm, lim["resolution"], [0, 0, 0], scan, 0.55, True, sm)


@pytest.mark.parametrize("min_score", [0.1, 0.55, 0.99])
def test_windowed_match_parity(sm, oracle, c2, min_score):
    cells, lim, _, truth, scan = c2
    init = [truth[0] + 0.4, truth[1] - 0.3, truth[2] + 0.15]
    om = _oracle(oracle, cells, lim, 7, 7.0, math.radians(30.0))
    gm = sm.FastCorrelativeScanMatcher2D(_grid(sm, cells, lim), 7, 7.0, math.radians(30.0))
    ref, _ = _assert_match_parity(om, gm, lim["resolution"], init, scan, min_score, False, sm)
    assert ref["found"] == (min_score < 0.9)


def test_reference_correct_pose_fixture(sm, oracle, synth):
    """FastCorrelativeScanMatcherTest.CorrectPose (fast_..._2d_test.cc:144-192):
    6-point cloud, 200x200 grid, window 3 m / 1 rad, depth 3, min_score 0.1."""
    from test_oracle_reference_pins import FAST_CLOUD, _rotate
    rng = np.random.default_rng(42)
    for _ in range(12):
        ex, ey, et = 2 * rng.uniform(-1, 1), 2 * rng.uniform(-1, 1), 0.5 * rng.uniform(-1, 1)
        g = synth.ProbabilityGrid(0.05, (5.0, 5.0), 200, 200)
        pts = _rotate(FAST_CLOUD, et) + np.array([ex, ey, 0], np.float32)
        g.insert([ex, ey], pts.astype(np.float32), None, 0.7, 0.4, True)
        lim = g.limits
        om = _oracle(oracle, g.cells, lim, 3, 3.0, 1.0)
        gm = sm.FastCorrelativeScanMatcher2D(_grid(sm, g.cells, lim), 3, 3.0, 1.0)
        ref, _ = _assert_match_parity(om, gm, 0.05, [0, 0, 0], FAST_CLOUD, 0.1, False, sm)
        assert ref["found"]


def test_reference_full_submap_fixture(sm, oracle, synth):
    """FastCorrelativeScanMatcherTest.FullSubmapMatching (:194-246), depth 6."""
    from test_oracle_reference_pins import FULL_CLOUD, _rotate
    rng = np.random.default_rng(7)
    for _ in range(6):
        px, py, pt = 10 * rng.uniform(-1, 1), 10 * rng.uniform(-1, 1), 1.6 * rng.uniform(-1, 1)
        cloud = (_rotate(FULL_CLOUD, pt) + np.array([px, py, 0])).astype(np.float32)
        qx, qy, qt = 2 * rng.uniform(-1, 1), 2 * rng.uniform(-1, 1), 0.5 * rng.uniform(-1, 1)
        et = qt - pt
        ipx = -(math.cos(-pt) * px - math.sin(-pt) * py)
        ipy = -(math.sin(-pt) * px + math.cos(-pt) * py)
        ex = qx + math.cos(qt) * ipx - math.sin(qt) * ipy
        ey = qy + math.sin(qt) * ipx + math.cos(qt) * ipy
        in_map = (_rotate(cloud, et) + np.array([ex, ey, 0])).astype(np.float32)
        g = synth.ProbabilityGrid(0.05, (5.0, 5.0), 200, 200)
        g.insert([qx, qy], in_map, None, 0.7, 0.4, True)
        lim = g.limits
        om = _oracle(oracle, g.cells, lim, 6, 3.0, 1.0)
        gm = sm.FastCorrelativeScanMatcher2D(_grid(sm, g.cells, lim), 6, 3.0, 1.0)
        _assert_match_parity(om, gm, 0.05, [0, 0, 0], cloud, 0.1, True, sm)


# ---- edge cases --------------------------------------------------------------
def test_edge_single_point_and_ragged_sizes(sm, oracle, c2):
    cells, lim, _, _, scan = c2
    om = _oracle(oracle, cells, lim, 4, 1.0, 0.3)
    gm = sm.FastCorrelativeScanMatcher2D(_grid(sm, cells, lim), 4, 1.0, 0.3)
    for n in (1, 2, 63, 64, 65, 129, 999):
        _assert_match_parity(om, gm, lim["resolution"], [0.5, 0.5, 0.1], scan[:n], 0.05, False, sm)


def test_edge_cloud_outside_grid(sm, oracle, c2):
    cells, lim, _, _, _ = c2
    far = np.array([[500.0, 500.0, 0], [501.0, 500.0, 0], [500.0, 502.0, 0]], np.float32)
    om = _oracle(oracle, cells, lim, 4, 1.0, 0.2)
    gm = sm.FastCorrelativeScanMatcher2D(_grid(sm, cells, lim), 4, 1.0, 0.2)
    # Every candidate scores ToScore(0) = 1.f - kMaxCorrespondenceCost (all lookups
    # outside the grid): not > 0.11 -> no match ...
    ref, _ = _assert_match_parity(om, gm, lim["resolution"], [0, 0, 0], far, 0.11, False, sm)
    assert not ref["found"]
    # ... but it is > 0.05.
    ref, _ = _assert_match_parity(om, gm, lim["resolution"], [0, 0, 0], far, 0.05, False, sm)
    assert ref["found"]


def test_edge_depth_one_and_empty_grid(sm, oracle):
    cells = np.zeros((40, 30), np.uint16)
    lim = dict(resolution=0.1, max_x=2.0, max_y=1.5, num_x_cells=30, num_y_cells=40)
    cloud = np.array([[0.3, 0.2, 0], [-0.4, 0.1, 0], [0.0, -0.5, 0]], np.float32)
    for depth in (1, 2):
        om = _oracle(oracle, cells, lim, depth, 0.5, 0.2)
        gm = sm.FastCorrelativeScanMatcher2D(_grid(sm, cells, lim), depth, 0.5, 0.2)
        ref, _ = _assert_match_parity(om, gm, 0.1, [0.1, 0.0, 0.0], cloud, 0.0, False, sm)
        assert ref["found"] and ref["score"] == pytest.approx(0.1, abs=1e-6)


def test_invalid_arguments(sm, c2):
    from cartographer_amd._lib import CmxError, INVALID_ARGUMENT
    cells, lim, _, _, scan = c2
    with pytest.raises(CmxError) as e:
        sm.FastCorrelativeScanMatcher2D(_grid(sm, cells, lim), 0)   # CHECK_GE(depth, 1)
    assert e.value.status == INVALID_ARGUMENT
    gm = sm.FastCorrelativeScanMatcher2D(_grid(sm, cells, lim), 3)
    with pytest.raises(CmxError) as e:
        gm.match_full_submap(np.zeros((0, 3), np.float32), 0.5)
    assert e.value.status == INVALID_ARGUMENT


# ---- batch + threads ----------------------------------------------------------
def test_batch_equals_individual(sm, oracle, synth):
    matchers, oracles, lims = [], [], []
    world0 = None
    for i in range(6):
        cells, lim, world = synth.make_submap(100 + i, 240, 240, 0.05, 15, 600, 30.0, 0.01)
        if i == 2:
            world0 = world
        matchers.append(sm.FastCorrelativeScanMatcher2D(_grid(sm, cells, lim), 6))
        oracles.append(_oracle(oracle, cells, lim, 6))
        lims.append(lim)
    scan = world0.scan(world0.free_pose(9, 0.5), 800, 30.0, 0.01, 3)
    found, scores, poses, stats = sm.match_full_submap_batch(matchers, scan, 0.6)
    cloud = sm.PointCloudOnDevice(scan)
    found_r, scores_r, poses_r, _ = sm.match_full_submap_batch(matchers, cloud, 0.6)
    np.testing.assert_array_equal(found, found_r)
    np.testing.assert_array_equal(scores, scores_r)
    np.testing.assert_array_equal(poses, poses_r)
    assert found[2] == 1
    total_coarse = 0
    for i, om in enumerate(oracles):
        ref = om.match_full_submap(scan, 0.6)
        assert bool(found[i]) == ref["found"]
        total_coarse += ref["coarse_candidates"]
        if ref["found"]:
            assert np.float32(scores[i]) == np.float32(ref["score"])
            f1, s1, p1 = matchers[i].match_full_submap(scan, 0.6)
            assert f1 and np.float32(s1) == np.float32(scores[i])
            np.testing.assert_allclose([p1.x, p1.y, p1.theta], poses[i], atol=1e-12)
    assert stats["coarse_candidates"] == total_coarse


def test_concurrent_matches_on_one_matcher(sm, c2):
    """All Match* are const and called concurrently from the reference's thread
    pool (constraints/constraint_builder_2d.cc:97-111)."""
    cells, lim, _, _, scan = c2
    gm = sm.FastCorrelativeScanMatcher2D(_grid(sm, cells, lim), 6)
    expected = gm.match_full_submap(scan, 0.6)
    results, errors = [None] * 4, []

    def work(i):
        try:
            for _ in range(3):
                results[i] = gm.match_full_submap(scan, 0.6)
        except Exception as exc:  # pragma: no cover
            errors.append(exc)
    threads = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not errors
    for r in results:
        assert r[0] == expected[0] and r[1] == expected[1]
        assert (r[2].x, r[2].y, r[2].theta) == (expected[2].x, expected[2].y, expected[2].theta)


# ----------------------------------------------------------------------------
# Real-time 2D
# ----------------------------------------------------------------------------
def test_rt2d_c1_parity(sm, oracle, synth):
    """BASELINE config C1: 1000 beams vs 200x200, window 0.3 m / 7 deg, weights 0.1/0.1."""
    cells, lim, world = synth.make_submap(42, 200, 200, 0.05, 25, 1000, 30.0, 0.01)
    truth = world.free_pose(77, 0.5)
    scan = world.scan(truth, 1000, 30.0, 0.01, 5)
    init = [truth[0] + 0.12, truth[1] - 0.08, truth[2] + math.radians(3.0)]
    ref = oracle.rt2d_match(cells, 0.05, lim["max_x"], lim["max_y"], init, scan, 0.3,
                            math.radians(7.0), 0.1, 0.1)
    m = sm.RealTimeCorrelativeScanMatcher2D(0.3, math.radians(7.0), 0.1, 0.1)
    score, pose = m.match(sm.Rigid2d(*init), scan, _grid(sm, cells, lim))
    assert m.last_stats["candidates_scored"] == ref["num_candidates"]
    assert score == ref["score"]          # f32 score widened to f64: bit-equal
    np.testing.assert_allclose([pose.x, pose.y, pose.theta], ref["pose"], rtol=0, atol=1e-12)
    assert abs(pose.x - truth[0]) < 0.06 and abs(pose.y - truth[1]) < 0.06


@pytest.mark.parametrize("weights", [(0.0, 0.0), (10.0, 1.0), (0.1, 5.0)])
@pytest.mark.parametrize("n", [1, 7, 200])
def test_rt2d_small_cases(sm, oracle, synth, weights, n):
    cells, lim, world = synth.make_submap(8, 120, 90, 0.05, 10, 400, 30.0, 0.01)
    truth = world.free_pose(3, 0.4)
    scan = world.scan(truth, 400, 30.0, 0.01, 1)[:n]
    init = [truth[0] - 0.07, truth[1] + 0.11, truth[2] - 0.05]
    ref = oracle.rt2d_match(cells, 0.05, lim["max_x"], lim["max_y"], init, scan, 0.2, 0.1,
                            weights[0], weights[1])
    m = sm.RealTimeCorrelativeScanMatcher2D(0.2, 0.1, weights[0], weights[1])
    score, pose = m.match(sm.Rigid2d(*init), scan, _grid(sm, cells, lim))
    assert score == ref["score"]
    np.testing.assert_allclose([pose.x, pose.y, pose.theta], ref["pose"], rtol=0, atol=1e-12)


def test_rt2d_reference_fixture(sm, oracle, synth):
    """RealTimeCorrelativeScanMatcherTest fixture (real_time_..._2d_test.cc:98-141):
    7-point L cloud in a 6x6 grid, window 0.6 m / 0.16 rad, zero weights."""
    from test_oracle_reference_pins import L_CLOUD, _rt_test_grid
    g = _rt_test_grid(synth)
    lim = g.limits
    ref = oracle.rt2d_match(g.cells, 0.05, lim["max_x"], lim["max_y"], [0, 0, 0], L_CLOUD, 0.6,
                            0.16, 0.0, 0.0)
    m = sm.RealTimeCorrelativeScanMatcher2D(0.6, 0.16, 0.0, 0.0)
    score, pose = m.match(sm.Rigid2d(0, 0, 0), L_CLOUD, _grid(sm, g.cells, lim))
    assert score == ref["score"] and score == pytest.approx(0.7, abs=1e-2)
    np.testing.assert_allclose([pose.x, pose.y, pose.theta], ref["pose"], rtol=0, atol=1e-12)


@pytest.mark.parametrize("lin,ang", [(1.0, 0.05), (0.0, 0.0), (0.05, 0.4)])
def test_rt2d_window_shapes(sm, oracle, synth, lin, ang):
    """Large linear window (41x41 offsets per rotation, points pushed outside the grid),
    the single-candidate window and a rotation-heavy one."""
    cells, lim, world = synth.make_submap(13, 96, 80, 0.05, 8, 300, 30.0, 0.01)
    truth = world.free_pose(2, 0.4)
    scan = world.scan(truth, 333, 30.0, 0.01, 4)
    init = [truth[0] + 0.4, truth[1] - 0.3, truth[2] + 0.02]
    ref = oracle.rt2d_match(cells, 0.05, lim["max_x"], lim["max_y"], init, scan, lin, ang, 0.3,
                            0.2)
    m = sm.RealTimeCorrelativeScanMatcher2D(lin, ang, 0.3, 0.2)
    score, pose = m.match(sm.Rigid2d(*init), scan, _grid(sm, cells, lim))
    assert m.last_stats["candidates_scored"] == ref["num_candidates"]
    assert score == ref["score"]
    np.testing.assert_allclose([pose.x, pose.y, pose.theta], ref["pose"], rtol=0, atol=1e-12)


# ----------------------------------------------------------------------------
# Real-time 2D on a TSDF2D (ComputeCandidateScore(TSDF2D), real_time_..._2d.cc:38-59)
# ----------------------------------------------------------------------------
def _tsdf_match_both(sm, oracle, tsd, wgt, lim, trunc, max_w, init, scan, lin, ang, tw, rw):
    ref = oracle.rt2d_match_tsdf(tsd, wgt, lim["resolution"], lim["max_x"], lim["max_y"], trunc,
                                 max_w, init, scan, lin, ang, tw, rw)
    m = sm.RealTimeCorrelativeScanMatcher2D(lin, ang, tw, rw)
    grid = sm.TSDF2D(tsd, wgt, lim["resolution"], lim["max_x"], lim["max_y"], trunc, max_w)
    score, pose = m.match(sm.Rigid2d(*init), scan, grid)
    assert m.last_stats["candidates_scored"] == ref["num_candidates"]
    assert score == ref["score"]          # f32 score widened to f64: bit-equal
    np.testing.assert_allclose([pose.x, pose.y, pose.theta], ref["pose"], rtol=0, atol=1e-12)
    return score, pose


@pytest.mark.parametrize("seed,n", [(42, 1000), (5, 77)])
def test_rt2d_tsdf_parity(sm, oracle, synth, seed, n):
    from tsdf_helpers import tsdf_from_probability_grid
    cells, lim, world = synth.make_submap(seed, 200, 200, 0.05, 25, 1000, 30.0, 0.01)
    tsd, wgt = tsdf_from_probability_grid(oracle, cells, 0.05, 0.3, 10.0, seed)
    truth = world.free_pose(77, 0.5)
    scan = world.scan(truth, 1000, 30.0, 0.01, 5)[:n]
    init = [truth[0] + 0.12, truth[1] - 0.08, truth[2] + math.radians(3.0)]
    score, pose = _tsdf_match_both(sm, oracle, tsd, wgt, lim, 0.3, 10.0, init, scan, 0.3,
                                   math.radians(7.0), 0.1, 0.1)
    assert score > 0.5
    if n == 1000:
        assert abs(pose.x - truth[0]) < 0.11 and abs(pose.y - truth[1]) < 0.11


def test_rt2d_tsdf_reference_fixture(sm, oracle):
    """The L-cloud TSDF of RealTimeCorrelativeScanMatcherTest::SetUpTSDF (:66-92)."""
    from test_oracle_reference_pins import L_CLOUD, _rt_test_tsdf
    tsd, wgt = _rt_test_tsdf(oracle)
    lim = dict(resolution=0.05, max_x=0.3, max_y=0.5)
    score, _ = _tsdf_match_both(sm, oracle, tsd, wgt, lim, 0.3, 1.0, [0, 0, 0], L_CLOUD, 0.6,
                                0.16, 0.0, 0.0)
    assert score > 0.95


def test_rt2d_tsdf_unknown_everywhere(sm, oracle):
    """Summed weight 0 -> score 0 for every candidate; the first candidate is returned."""
    from test_oracle_reference_pins import L_CLOUD
    zeros = np.zeros((20, 20), np.uint16)
    lim = dict(resolution=0.05, max_x=0.3, max_y=0.5)
    score, pose = _tsdf_match_both(sm, oracle, zeros, zeros, lim, 0.3, 1.0, [0, 0, 0], L_CLOUD,
                                   0.1, 0.05, 0.0, 0.0)
    assert score == 0.0 and pose.theta < 0


# ----------------------------------------------------------------------------
# Large clouds: beyond the on-chip staging sizes (4096-point LDS cache of the tree
# search, 1024-point register window of the real-time scorer)
# ----------------------------------------------------------------------------
def test_large_cloud_fast2d(sm, oracle, synth):
    cells, lim, world = synth.make_submap(91, 240, 200, 0.05, 15, 600, 30.0, 0.01)
    truth = world.free_pose(8, 0.5)
    scan = world.scan(truth, 5000, 30.0, 0.01, 6)
    assert scan.shape[0] > 4096
    om = _oracle(oracle, cells, lim, 6, 2.0, 0.35)
    gm = sm.FastCorrelativeScanMatcher2D(_grid(sm, cells, lim), 6, 2.0, 0.35)
    init = [truth[0] + 0.2, truth[1] - 0.15, truth[2] + 0.05]
    _assert_match_parity(om, gm, lim["resolution"], init, scan, 0.5, False, sm)
    _assert_match_parity(om, gm, lim["resolution"], init, scan, 0.5, True, sm)


@pytest.mark.parametrize("n", [1025, 2500])
def test_large_cloud_rt2d(sm, oracle, synth, n):
    cells, lim, world = synth.make_submap(92, 200, 200, 0.05, 15, 600, 30.0, 0.01)
    truth = world.free_pose(9, 0.5)
    scan = world.scan(truth, 3000, 30.0, 0.01, 6)[:n]
    assert scan.shape[0] == n
    init = [truth[0] + 0.05, truth[1] - 0.04, truth[2] + 0.02]
    ref = oracle.rt2d_match(cells, 0.05, lim["max_x"], lim["max_y"], init, scan, 0.1,
                            math.radians(3.0), 0.1, 0.1)
    m = sm.RealTimeCorrelativeScanMatcher2D(0.1, math.radians(3.0), 0.1, 0.1)
    score, pose = m.match(sm.Rigid2d(*init), scan, _grid(sm, cells, lim))
    assert m.last_stats["candidates_scored"] == ref["num_candidates"]
    assert score == ref["score"]
    np.testing.assert_allclose([pose.x, pose.y, pose.theta], ref["pose"], rtol=0, atol=1e-12)


@pytest.mark.parametrize("depth", [9, 11])
def test_deep_stack_generic_lowest_resolution_path(sm, oracle, synth, depth):
    """branch_and_bound_depth beyond the phase-plane limit (lowest-resolution width 256 / 1024
    cells > 128): the generic wave-per-candidate scorer runs instead of the plane scorer, and
    a lowest-resolution cell covers the whole grid."""
    cells, lim, world = synth.make_submap(93, 150, 130, 0.05, 12, 400, 30.0, 0.01)
    truth = world.free_pose(10, 0.5)
    scan = world.scan(truth, 300, 30.0, 0.01, 6)
    om = _oracle(oracle, cells, lim, depth, 3.0, 0.4)
    gm = sm.FastCorrelativeScanMatcher2D(_grid(sm, cells, lim), depth, 3.0, 0.4)
    init = [truth[0] + 0.3, truth[1] - 0.2, truth[2] + 0.1]
    _assert_match_parity(om, gm, lim["resolution"], init, scan, 0.4, False, sm)
    _assert_match_parity(om, gm, lim["resolution"], init, scan, 0.4, True, sm)


def test_rt2d_tsdf_large_cloud(sm, oracle, synth):
    from tsdf_helpers import tsdf_from_probability_grid
    cells, lim, world = synth.make_submap(94, 160, 160, 0.05, 12, 600, 30.0, 0.01)
    tsd, wgt = tsdf_from_probability_grid(oracle, cells, 0.05, 0.3, 10.0, 3)
    truth = world.free_pose(11, 0.5)
    scan = world.scan(truth, 1500, 30.0, 0.01, 5)
    assert scan.shape[0] > 1024
    init = [truth[0] + 0.06, truth[1] - 0.03, truth[2] + 0.01]
    _tsdf_match_both(sm, oracle, tsd, wgt, lim, 0.3, 10.0, init, scan, 0.15, math.radians(3.0),
                     0.2, 0.1)


def test_caller_owned_stream(sm, oracle, c2):
    """cmx_set_stream: the calling thread's matches run on a stream the caller owns (a torch
    stream here) and still return the oracle's result; NULL restores the library stream."""
    import ctypes
    import torch
    from cartographer_amd import _lib
    cells, lim, _, _, scan = c2
    om = _oracle(oracle, cells, lim, 5, 1.5, 0.3)
    gm = sm.FastCorrelativeScanMatcher2D(_grid(sm, cells, lim), 5, 1.5, 0.3)
    stream = torch.cuda.Stream()
    _lib.check(_lib.lib().cmx_set_stream(0, ctypes.c_void_p(stream.cuda_stream)))
    try:
        # Work queued on the caller's stream before and after the match stays ordered with it.
        with torch.cuda.stream(stream):
            marker = torch.ones(1 << 20, device="cuda").sum()
            _assert_match_parity(om, gm, lim["resolution"], [0.4, 0.3, 0.1], scan[:300], 0.2, False,
                                 sm)
            after = torch.ones(8, device="cuda").sum()
        stream.synchronize()
        assert float(marker) == float(1 << 20) and float(after) == 8.0
    finally:
        _lib.check(_lib.lib().cmx_set_stream(0, None))
    _assert_match_parity(om, gm, lim["resolution"], [0.4, 0.3, 0.1], scan[:300], 0.2, False, sm)
