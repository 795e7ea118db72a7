"""Pins the CPU oracle against the reference's known-answer tests of the 3D hot
path (paths relative to /root/reference/cartographer/mapping/internal/3d/scan_matching)."""
import math

import numpy as np
import pytest

RT_CLOUD = np.array([[-3, 2, 0], [-4, 2, 0], [-5, 2, 0], [-6, 2, 0], [-6, 3, 1], [-6, 4, 2],
                     [-7, 3, 1]], np.float32)


def quat_from_angle_axis(angle, axis, normalize_axis=True):
    """Quaternion(AngleAxisd(angle, axis)).  Eigen does NOT normalise the axis:
    the reference's RotationAroundYZ test passes the raw axis (0,1,1), which
    yields a non-unit quaternion (a rotation of 0.8 deg * sqrt(2))."""
    axis = np.asarray(axis, np.float64)
    if normalize_axis:
        axis = axis / np.linalg.norm(axis)
    return [math.cos(angle / 2)] + list(math.sin(angle / 2) * axis)


def rt3d_fixture(synth):
    # real_time_correlative_scan_matcher_3d_test.cc:38-52: hybrid grid 0.1, every point
    # of the cloud (shifted by the expected pose (-1,0,0)) set to probability 1.
    g = synth.HybridGrid(0.1)
    for p in RT_CLOUD:
        g.set_probability(g.get_cell_index(p + np.array([-1, 0, 0], np.float32)), 1.0)
    return g


RT_INITIAL_POSES = [
    ([-1.0, 0.0, 0.0], [1, 0, 0, 0]),                                   # PerfectEstimate :83
    ([-0.8, 0.0, 0.0], [1, 0, 0, 0]),                                   # AlongX :88
    ([-1.0, 0.0, -0.2], [1, 0, 0, 0]),                                  # AlongZ :93
    ([-0.9, -0.2, 0.2], [1, 0, 0, 0]),                                  # AlongXYZ :98
    ([-1.0, 0.0, 0.0], quat_from_angle_axis(0.8 / 180 * math.pi, [1, 0, 0])),   # RotationAroundX
    ([-1.0, 0.0, 0.0], quat_from_angle_axis(0.8 / 180 * math.pi, [0, 1, 0])),   # RotationAroundY
    ([-1.0, 0.0, 0.0], quat_from_angle_axis(0.8 / 180 * math.pi, [0, 1, 1], False)),   # RotationAroundYZ
]


def is_nearly_3d(pose_a, pose_b, eps):
    """transform::IsNearly on the 4x4 homogeneous matrices (isApprox)."""
    def mat(p):
        t, (w, x, y, z) = p[:3], p[3:]
        r = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                      [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        m = np.eye(4)
        m[:3, :3] = r
        m[:3, 3] = t
        return m
    a, b = mat(np.asarray(pose_a, float)), mat(np.asarray(pose_b, float))
    return np.linalg.norm(a - b) <= eps * min(np.linalg.norm(a), np.linalg.norm(b))


@pytest.mark.parametrize("t,q", RT_INITIAL_POSES)
def test_rt3d_reference_fixture(oracle, synth, t, q):
    # all seven initial poses must return (-1, 0, 0) within 1e-3 (:74-117)
    g = rt3d_fixture(synth)
    r = oracle.rt3d_match(0.1, g.voxels(), list(t) + list(q), RT_CLOUD, 0.3, math.radians(1.0),
                          1e-1, 1.0)
    assert is_nearly_3d(r["pose"], [-1, 0, 0, 1, 0, 0, 0], 1e-3)
    assert r["score"] > 0.85


def test_precomputation_grid_3d_against_naive(oracle, synth):
    # precomputation_grid_3d_test.cc:31-77: 1000 random voxels, depths 0..3 with
    # shift 2^(d-1), no half resolution -> stack with full_resolution_depth > depth.
    rng = np.random.default_rng(23847)
    g = synth.HybridGrid(2.0)
    coords = rng.integers(-50, 50, (1000, 3))
    probs = rng.uniform(0.1, 0.9, 1000).astype(np.float32)
    dense = {}
    for c, p in zip(coords, probs):
        g.set_probability(c, float(p))
        dense[tuple(int(v) for v in c)] = g.get_probability(c)
    vox = g.voxels()
    m = oracle.FastCorrelativeScanMatcher3D(2.0, vox, 2.0, vox, np.zeros(4, np.float32), 4, 8,
                                            0.0, 0.0, 1.0, 1.0, 0.1)
    for depth in range(4):
        level = m.level(depth)
        table = {(int(r[0]), int(r[1]), int(r[2])): int(r[3]) for r in level}
        width = 1 << depth
        for _ in range(100):
            x, y, z = (int(v) for v in rng.integers(-50, 50, 3))
            expect = 0.0
            for dx in range(width):
                for dy in range(width):
                    for dz in range(width):
                        expect = max(expect, dense.get((x + dx, y + dy, z + dz), 0.1))
            got = 0.1 + table.get((x, y, z), 0) * (0.8 / 255.0)
            assert got == pytest.approx(expect, abs=1e-2)


def test_rotational_scan_matcher_interpolates(oracle):
    # rotational_scan_matcher_test.cc:38-67, closed form t / hypot(t, 1-t) to 1e-6
    nb = 10
    per_bucket = np.float32(math.pi / nb)
    submap = np.zeros(nb, np.float32); submap[3] = 1
    unit2 = np.zeros(nb, np.float32); unit2[2] = 1
    unit4 = np.zeros(nb, np.float32); unit4[4] = 1
    t = np.float32(0.0)
    while t < 1.0:
        expected = float(t) / math.hypot(float(t), 1 - float(t)) if t > 0 else 0.0
        s = oracle.rotational_match(submap, unit2, 0.0, [t * per_bucket])
        assert s[0] == pytest.approx(expected, abs=1e-6)
        s = oracle.rotational_match(submap, unit2, 0.0, [(2 - t) * per_bucket])
        assert s[0] == pytest.approx(expected, abs=1e-6)
        s = oracle.rotational_match(submap, unit4, 0.0, [-t * per_bucket, (t - 2) * per_bucket])
        assert s[0] == pytest.approx(expected, abs=1e-6)
        assert s[1] == pytest.approx(expected, abs=1e-6)
        t = np.float32(t + np.float32(0.1))


def test_rotational_scan_matcher_own_histogram(oracle):
    # :28-36  a histogram matches itself with score 1 and a rotated copy with less
    h = np.array([1, 43, 0.5, 0.3123, 23, 42, 0], np.float32)
    s = oracle.rotational_match(h, h, 0.0, [0.0, 1.0])
    assert s[0] == pytest.approx(1.0, abs=1e-6) and s[1] < 1.0


FAST3D_CLOUD = np.array([[4, 0, 0], [4.5, 0, 0], [5, 0, 0], [5.5, 0, 0], [0, 4, 0], [0, 4.5, 0],
                         [0, 5, 0], [0, 5.5, 0], [0, 0, 4], [0, 0, 4.5], [0, 0, 5], [0, 0, 5.5]],
                        np.float32)


def fast3d_fixture(synth, pose_t, theta):
    """GetFastCorrelativeScanMatcher (fast_correlative_scan_matcher_3d_test.cc:108-124):
    grid 0.05, one range-data insertion of the cloud transformed by the expected pose."""
    c, s = math.cos(theta), math.sin(theta)
    rot = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], np.float32)
    in_map = (FAST3D_CLOUD @ rot.T + np.asarray(pose_t, np.float32)).astype(np.float32)
    g = synth.HybridGrid(0.05)
    g.insert(np.asarray(pose_t, np.float32), in_map, 0.7, 0.4, 5)
    return g


def test_fast3d_correct_pose_for_match(oracle, synth):
    # :146-178  depth 6 = full_resolution_depth 6, window 0.8/0.8/0.3, 20 random poses,
    # tolerance 0.05; a far-away low-resolution cloud must yield no match.
    rng = np.random.default_rng(42)
    for _ in range(8):
        t = 0.7 * rng.uniform(-1, 1, 3)
        theta = 0.2 * rng.uniform(-1, 1)
        g = fast3d_fixture(synth, t, theta)
        vox = g.voxels()
        hist = np.zeros(10, np.float32)
        m = oracle.FastCorrelativeScanMatcher3D(0.05, vox, 0.05, vox, hist, 6, 6, 0.1, 0.15,
                                                0.8, 0.8, 0.3)
        ident = [0, 0, 0, 1, 0, 0, 0]
        r = m.match(ident, ident, [1, 0, 0, 0], FAST3D_CLOUD, FAST3D_CLOUD, hist, 0.1)
        assert r["found"] and r["score"] > 0.1
        assert r["rotational_score"] > 0.09 and r["low_resolution_score"] > 0.14
        expected = list(t) + quat_from_angle_axis(theta, [0, 0, 1])
        assert is_nearly_3d(expected, r["pose"], 0.05)
        far = np.array([[42, 42, 42]], np.float32)
        r2 = m.match(ident, ident, [1, 0, 0, 0], FAST3D_CLOUD, far, hist, 0.1)
        assert not r2["found"]


def test_fast3d_correct_pose_for_match_full_submap(oracle, synth):
    # :180-204
    rng = np.random.default_rng(7)
    t = 0.7 * rng.uniform(-1, 1, 3)
    theta = 0.2 * rng.uniform(-1, 1)
    g = fast3d_fixture(synth, t, theta)
    vox = g.voxels()
    hist = np.zeros(10, np.float32)
    m = oracle.FastCorrelativeScanMatcher3D(0.05, vox, 0.05, vox, hist, 6, 6, 0.1, 0.15, 0.8,
                                            0.8, 0.3)
    r = m.match_full_submap([1, 0, 0, 0], [1, 0, 0, 0], [1, 0, 0, 0], FAST3D_CLOUD, FAST3D_CLOUD,
                            hist, 0.1)
    assert r["found"] and r["score"] > 0.1
    expected = list(t) + quat_from_angle_axis(theta, [0, 0, 1])
    assert is_nearly_3d(expected, r["pose"], 0.05)
    far = np.array([[42, 42, 42]], np.float32)
    r2 = m.match_full_submap([1, 0, 0, 0], [1, 0, 0, 0], [1, 0, 0, 0], FAST3D_CLOUD, far, hist,
                             0.1)
    assert not r2["found"]


# ---- mapping/3d/hybrid_grid_test.cc ------------------------------------------
def test_hybrid_grid_get_cell_index(synth):
    # :106-127  resolution 2: lround(p / 2), halves round away from zero
    g = synth.HybridGrid(2.0)
    cases = [((0, 0, 0), (0, 0, 0)), ((0, 26, 10), (0, 13, 5)), ((14, 0, 10), (7, 0, 5)),
             ((14, 26, 0), (7, 13, 0)), ((8.5, 11.5, 0.5), (4, 6, 0)),
             ((7.5, 12.5, 1.5), (4, 6, 1)), ((6.5, 14.5, 2.5), (3, 7, 1)),
             ((5.5, 13.5, 3.5), (3, 7, 2))]
    for p, want in cases:
        assert tuple(g.get_cell_index(np.array(p, np.float32))) == want


# ---- mapping/3d/range_data_inserter_3d_test.cc ----------------------------------
INSERTER_3D_RETURNS = np.array([[-3, -1, 4], [-2, 0, 4], [-1, 1, 4], [0, 2, 4]], np.float32)
INSERTER_3D_ORIGIN = np.array([0, 0, -4], np.float32)


def _known(grid, x, y, z):
    v = grid.voxels()
    return bool(((v["x"] == x) & (v["y"] == y) & (v["z"] == z)).any())


def test_range_data_inserter_3d_insert_point_cloud(synth):
    """:97-114: 1 m grid, hit 0.7 / miss 0.4, 1000 free-space voxels."""
    g = synth.HybridGrid(1.0)
    g.insert(INSERTER_3D_ORIGIN, INSERTER_3D_RETURNS, 0.7, 0.4, 1000)
    for z in (-4, -3, -2):
        assert abs(g.get_probability((0, 0, z)) - 0.4) < 1e-4
    for x in range(-4, 5):
        for y in range(-4, 5):
            if x < -3 or x > 0 or y != x + 2:
                assert not _known(g, x, y, 4)
            else:
                assert abs(g.get_probability((x, y, 4)) - 0.7) < 1e-4


def test_range_data_inserter_3d_probability_progression(synth):
    """:138-157: 1001 insertions saturate hit and miss cells at the probability bounds."""
    g = synth.HybridGrid(1.0)
    g.insert(INSERTER_3D_ORIGIN, INSERTER_3D_RETURNS, 0.7, 0.4, 1000)
    assert abs(g.get_probability((-2, 0, 4)) - 0.7) < 1e-4
    assert abs(g.get_probability((-2, 0, 3)) - 0.4) < 1e-4
    for _ in range(1000):
        g.insert(INSERTER_3D_ORIGIN, INSERTER_3D_RETURNS, 0.7, 0.4, 1000)
    assert abs(g.get_probability((-2, 0, 4)) - 0.9) < 1e-3
    assert abs(g.get_probability((-2, 0, 3)) - 0.1) < 1e-3
