"""Device-resident ProbabilityGrid: range-data insertion and matching on the
grid in HBM, against the host restatement of ProbabilityGridRangeDataInserter2D (the fixture
builder `synth.ProbabilityGrid`, itself pinned on the reference's inserter / ray-mask tests in
test_oracle_reference_pins.py) and the CPU oracle.  Bar: every cell bit-exact, limits equal.
"""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g2():
    from cartographer_amd import _lib, grid_2d
    assert _lib.lib().cmx_device_count() >= 1, "no HIP device: these tests need the GPU"
    return grid_2d


def _in_map(pose, sensor_points):
    c, s = math.cos(pose[2]), math.sin(pose[2])
    p = sensor_points.astype(np.float64)
    out = np.zeros_like(sensor_points)
    out[:, 0] = (pose[0] + c * p[:, 0] - s * p[:, 1]).astype(np.float32)
    out[:, 1] = (pose[1] + s * p[:, 0] + c * p[:, 1]).astype(np.float32)
    return out


def _assert_same(dev, host):
    assert dev.limits == host.limits
    np.testing.assert_array_equal(dev.cells, host.cells)


def test_reference_inserter_fixture(g2, synth):
    """RealTimeCorrelativeScanMatcherTest::SetUpProbabilityGrid (real_time_..._2d_test.cc:98-120):
    6x6 grid, the 7-point L cloud from the origin, hit 0.7 / miss 0.4."""
    from test_oracle_reference_pins import L_CLOUD
    host = synth.ProbabilityGrid(0.05, (0.05, 0.25), 6, 6)
    host.insert([0.0, 0.0], L_CLOUD, None, 0.7, 0.4, True)
    dev = g2.ProbabilityGridOnDevice(0.05, (0.05, 0.25), 6, 6)
    dev.insert([0.0, 0.0], L_CLOUD, None, 0.7, 0.4, True)
    _assert_same(dev, host)
    assert (dev.cells != 0).sum() > 7


@pytest.mark.parametrize("seed,free_space", [(3, True), (9, True), (4, False)])
def test_insert_parity_with_growth(g2, synth, seed, free_space):
    """Twelve scans of a synthetic room inserted into a grid that starts as 16x16 cells: the
    limits double several times (GrowLimits), later scans re-update known cells through the odds
    tables, misses (max-range returns) only clear free space."""
    _, lim, world = synth.make_submap(seed, 200, 200, 0.05, 2, 100, 30.0, 0.01)
    start = (lim["max_x"] - 4.0, lim["max_y"] - 4.0)
    host = synth.ProbabilityGrid(0.05, start, 16, 16)
    dev = g2.ProbabilityGridOnDevice(0.05, start, 16, 16)
    rng = np.random.default_rng(seed)
    for k in range(12):
        pose = world.free_pose(seed * 100 + k, 0.4)
        cloud = _in_map(pose, world.scan(pose, 257, 30.0, 0.01, k))
        # every seventh beam is a miss (no return within range): free space only
        miss_mask = np.arange(cloud.shape[0]) % 7 == 3
        returns, misses = cloud[~miss_mask], cloud[miss_mask]
        hit, miss = (0.55, 0.49) if k % 3 == 2 else (0.7, 0.4)
        host.insert(pose[:2], returns, misses, hit, miss, free_space)
        dev.insert(pose[:2], returns, misses, hit, miss, free_space)
        _assert_same(dev, host)
    assert dev.limits["num_x_cells"] > 16
    # degenerate inputs: nothing to insert, and a ray that stays inside one cell
    host.insert(pose[:2], np.zeros((0, 3), np.float32), None, 0.7, 0.4, free_space)
    dev.insert(pose[:2], np.zeros((0, 3), np.float32), None, 0.7, 0.4, free_space)
    tiny = np.array([[pose[0] + 1e-3, pose[1] + 2e-3, 0.0]], np.float32)
    host.insert(pose[:2], tiny, None, 0.7, 0.4, free_space)
    dev.insert(pose[:2], tiny, None, 0.7, 0.4, free_space)
    _assert_same(dev, host)
    del rng


def test_axis_aligned_and_corner_rays(g2, synth):
    """Rays along the grid axes and exactly through pixel corners (the cases
    ray_to_pixel_mask_test.cc exercises), in all eight octants."""
    host = synth.ProbabilityGrid(0.05, (1.0, 1.0), 40, 40)
    dev = g2.ProbabilityGridOnDevice(0.05, (1.0, 1.0), 40, 40)
    origin = [0.025, 0.025]          # a cell centre
    ends = []
    for dx, dy in [(1, 0), (-1, 0), (0, 1), (0, -1), (1, 1), (-1, 1), (1, -1), (-1, -1),
                   (2, 1), (-1, 2), (3, -2), (-2, -3)]:
        ends.append([origin[0] + 0.1 * dx * 2.5, origin[1] + 0.1 * dy * 2.5, 0.0])
    ends = np.array(ends, np.float32)
    host.insert(origin, ends, None, 0.7, 0.4, True)
    dev.insert(origin, ends, None, 0.7, 0.4, True)
    _assert_same(dev, host)


def test_rt2d_on_resident_grid(g2, synth, oracle):
    """Match -> Insert -> Match on the grid in HBM equals the host-buffer entry point and the
    oracle at every step (LocalTrajectoryBuilder2D's per-scan loop)."""
    from cartographer_amd import scan_matching as sm
    _, lim, world = synth.make_submap(21, 200, 200, 0.05, 2, 100, 30.0, 0.01)
    start = (lim["max_x"] - 2.0, lim["max_y"] - 2.0)
    dev = g2.ProbabilityGridOnDevice(0.05, start, 100, 100)
    matcher = sm.RealTimeCorrelativeScanMatcher2D(0.2, math.radians(5.0), 0.1, 0.1)
    pose = world.free_pose(5, 0.5)
    for k in range(5):
        scan = world.scan(pose, 400, 30.0, 0.01, k)
        dev.insert(pose[:2], _in_map(pose, scan))
        # next scan from a slightly moved pose, matched against the grid built so far
        pose = (pose[0] + 0.06, pose[1] - 0.04, pose[2] + 0.02)
        scan = world.scan(pose, 400, 30.0, 0.01, 50 + k)
        init = [pose[0] + 0.05, pose[1] + 0.03, pose[2] - 0.015]
        score_dev, pose_dev = matcher.match(sm.Rigid2d(*init), scan, dev)
        cells, l2 = dev.cells, dev.limits
        host_grid = sm.Grid2D(cells, l2["resolution"], l2["max_x"], l2["max_y"])
        score_host, pose_host = matcher.match(sm.Rigid2d(*init), scan, host_grid)
        ref = oracle.rt2d_match(cells, l2["resolution"], l2["max_x"], l2["max_y"], init, scan,
                                0.2, math.radians(5.0), 0.1, 0.1)
        assert score_dev == score_host == ref["score"]
        assert (pose_dev.x, pose_dev.y, pose_dev.theta) == (pose_host.x, pose_host.y,
                                                             pose_host.theta)
        np.testing.assert_allclose([pose_dev.x, pose_dev.y, pose_dev.theta], ref["pose"], rtol=0,
                                   atol=1e-12)


def test_fast_matcher_from_device_grid(g2, synth):
    from cartographer_amd import scan_matching as sm
    cells, lim, world = synth.make_submap(33, 120, 100, 0.05, 8, 300, 30.0, 0.01)
    dev = g2.ProbabilityGridOnDevice(lim["resolution"], (lim["max_x"], lim["max_y"]),
                                     lim["num_x_cells"], lim["num_y_cells"], cells=cells)
    a = dev.fast_matcher(5)
    b = sm.FastCorrelativeScanMatcher2D(
        sm.Grid2D(cells, lim["resolution"], lim["max_x"], lim["max_y"]), 5)
    for level in range(5):
        np.testing.assert_array_equal(a.level(level), b.level(level))
    pose = world.free_pose(1, 0.4)
    scan = world.scan(pose, 300, 30.0, 0.01, 2)
    assert a.match_full_submap(scan, 0.4)[:2] == b.match_full_submap(scan, 0.4)[:2]


@pytest.mark.parametrize("robots", [5, 19])     # below / above the 4-candidates-per-lane switch
def test_rt2d_batch_equals_individual(g2, synth, oracle, robots):
    """Several robots, each with its own grid, scan and pose, matched in one batch: every
    result equals the single-match entry point (and the oracle)."""
    from cartographer_amd import scan_matching as sm
    matcher = sm.RealTimeCorrelativeScanMatcher2D(0.15, math.radians(4.0), 0.1, 0.2)
    grids, inits, scans = [], [], []
    for k in range(robots):
        nx, ny = 120 + 4 * k, 100 + 2 * k             # different grid sizes ...
        cells, lim, world = synth.make_submap(70 + k, nx, ny, 0.05, 10, 300, 30.0, 0.01)
        grids.append(g2.ProbabilityGridOnDevice(0.05, (lim["max_x"], lim["max_y"]), nx, ny,
                                                cells=cells))
        pose = world.free_pose(k, 0.4)
        scans.append(world.scan(pose, 150 + 37 * k, 30.0, 0.01, k))   # ... and scan sizes
        inits.append(sm.Rigid2d(pose[0] + 0.04, pose[1] - 0.06, pose[2] + 0.01 * k))
    scores, poses, stats = sm.rt2d_match_batch(matcher, grids, inits, scans)
    total = 0
    for k in range(robots):
        s1, p1 = matcher.match(inits[k], scans[k], grids[k])
        total += matcher.last_stats["candidates_scored"]
        assert scores[k] == s1
        assert (poses[k].x, poses[k].y, poses[k].theta) == (p1.x, p1.y, p1.theta)
        l = grids[k].limits
        ref = oracle.rt2d_match(grids[k].cells, 0.05, l["max_x"], l["max_y"],
                                [inits[k].x, inits[k].y, inits[k].theta], scans[k], 0.15,
                                math.radians(4.0), 0.1, 0.2)
        assert scores[k] == ref["score"]
    assert stats["candidates_scored"] == total
