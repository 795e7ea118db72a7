"""CeresScanMatcher3D: the oracle's restatement against the reference's own
test fixture and expectations (ceres_scan_matcher_3d_test.cc:36-128, without the intensity term:
IntensityHybridGrid is out of scope) and against finite differences of its own residuals.  Ceres
itself is absent from /root/reference (third party); see oracle/oracle_ceres_3d.h for what is
restated and what stays unpinned."""
import math

import numpy as np
import pytest

from test_oracle_reference_pins_3d import quat_from_angle_axis

POINTS = np.array([[-3, 2, 0], [-4, 2, 0], [-5, 2, 0], [-6, 2, 0], [-6, 3, 1], [-6, 4, 2],
                   [-7, 3, 1]], np.float32)


def quat_mul(a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.array([aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw])


def rotate(q, v):
    qv = np.array([0.0, *v])
    conj = np.array([q[0], -q[1], -q[2], -q[3]])
    return quat_mul(quat_mul(q, qv), conj)[1:]


def fixture(synth, points, expected_t=(-1.0, 0.0, 0.0), expected_q=(1.0, 0.0, 0.0, 0.0)):
    """hybrid_grid_(1.f) with probability 1 at the cells of expected_pose * point (:43-55)."""
    grid = synth.HybridGrid(1.0)
    q = np.asarray(expected_q, np.float64)
    for p in points:
        w = rotate(q, p.astype(np.float64)) + np.asarray(expected_t)
        grid.set_probability(tuple(int(round(float(np.float32(c)))) for c in w), 1.0)
    return grid.voxels()


def to_matrix(pose7):
    t, q = np.asarray(pose7[:3], np.float64), np.asarray(pose7[3:], np.float64)
    m = np.eye(4)
    for k in range(3):
        e = np.zeros(3)
        e[k] = 1.0
        m[:3, k] = rotate(q, e)          # Eigen's toRotationMatrix of the stored coefficients
    m[:3, 3] = t
    return m


def is_nearly(pose7, expected7, epsilon):
    """transform::IsNearly (rigid_transform_test_helpers.h:42-46): Eigen's isApprox on the 4 x 4
    transforms, |a - b|_F <= epsilon min(|a|_F, |b|_F)."""
    a, b = to_matrix(pose7), to_matrix(expected7)
    return np.linalg.norm(a - b) <= epsilon * min(np.linalg.norm(a), np.linalg.norm(b))


CASES = [
    ("PerfectEstimate", (-1.0, 0.0, 0.0), (1, 0, 0, 0)),
    ("AlongX", (-0.8, 0.0, 0.0), (1, 0, 0, 0)),
    ("AlongZ", (-1.0, 0.0, -0.2), (1, 0, 0, 0)),
    ("AlongXYZ", (-0.9, -0.2, 0.2), (1, 0, 0, 0)),
]


@pytest.mark.parametrize("name,t,q", CASES)
def test_reference_ceres_scan_matcher_3d_test(oracle, synth, name, t, q):
    vox = fixture(synth, POINTS)
    init = list(t) + list(q)
    out = oracle.ceres3d_match([(POINTS, 1.0, vox)], t, init, [1.0], translation_weight=0.01,
                               rotation_weight=0.1, use_nonmonotonic_steps=True,
                               max_num_iterations=10)
    assert abs(out["final_cost"]) < 1e-2, out
    assert is_nearly(out["pose"], [-1.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0], 3e-2), out


def test_reference_full_pose_correction(oracle, synth):
    """FullPoseCorrection (:113-125): the cloud turned by 0.05 rad about z, the search started
    with a rotation about x."""
    add = np.array(quat_from_angle_axis(0.05, [0, 0, 1]))
    cloud = np.array([rotate(add, p.astype(np.float64)) for p in POINTS], np.float32)
    inv = np.array([add[0], -add[1], -add[2], -add[3]])
    vox = fixture(synth, POINTS)                       # expected_pose * original points
    init_t = (-0.95, -0.05, 0.05)
    init = list(init_t) + quat_from_angle_axis(0.05, [1, 0, 0])
    out = oracle.ceres3d_match([(cloud, 1.0, vox)], init_t, init, [1.0], translation_weight=0.01,
                               rotation_weight=0.1, use_nonmonotonic_steps=True,
                               max_num_iterations=10)
    assert abs(out["final_cost"]) < 1e-2, out
    # expected_pose = Translation(-1, 0, 0) * additional_transform^-1
    assert is_nearly(out["pose"], [-1.0, 0.0, 0.0] + list(inv), 3e-2), out


def test_jacobian_matches_finite_differences(oracle, synth):
    grid, world = synth.make_submap_3d(5, 0.1, (8.0, 8.0, 4.0), 4, 8, 96)
    low, _ = synth.make_submap_3d(5, 0.3, (8.0, 8.0, 4.0), 4, 8, 96)
    pos = world.free_position(6, 0.5)
    full = world.scan(pos, 0.3, 8, 64, seed=9)
    hi, lo = full[::3].copy(), full[::11].copy()
    pairs = [(hi, 0.1, grid.voxels()), (lo, 0.3, low.voxels())]
    q = np.array(quat_from_angle_axis(0.3, [0.2, -0.1, 0.9]))
    pose = np.array(list(pos + np.array([0.013, -0.021, 0.017])) + list(q))
    target_q = quat_from_angle_axis(0.27, [0.1, 0.0, 1.0])
    weights = [1.0, 6.0]
    r, J = oracle.ceres3d_residuals(pairs, pos, target_q, pose, weights)
    assert r.shape[0] == len(hi) + len(lo) + 6 and np.isfinite(J).all()
    # Central differences; the interpolant is C1 with kinks in its second derivative at voxel
    # centres, so a few rows next to a centre are less accurate.
    for k in range(7):
        d = np.zeros(7)
        d[k] = 1e-6
        rp, _ = oracle.ceres3d_residuals(pairs, pos, target_q, pose + d, weights)
        rm, _ = oracle.ceres3d_residuals(pairs, pos, target_q, pose - d, weights)
        fd = (rp - rm) / 2e-6
        err = np.abs(fd - J[:, k])
        assert np.median(err) < 1e-6 and (err > 1e-3).sum() <= 3, (k, err.max())


@pytest.mark.parametrize("yaw_only", [False, True])
def test_match_reduces_the_cost_and_keeps_the_parameterization(oracle, synth, yaw_only):
    grid, world = synth.make_submap_3d(7, 0.1, (8.0, 8.0, 4.0), 4, 8, 96)
    low, _ = synth.make_submap_3d(7, 0.3, (8.0, 8.0, 4.0), 4, 8, 96)
    pos = world.free_position(8, 0.5)
    yaw = 0.4
    full = world.scan(pos, yaw, 16, 128, seed=3)
    hi, lo = full[::2].copy(), full[::9].copy()
    pairs = [(hi, 0.1, grid.voxels()), (lo, 0.3, low.voxels())]
    truth_q = quat_from_angle_axis(yaw, [0, 0, 1])
    init_t = pos + np.array([0.04, -0.03, 0.0 if yaw_only else 0.02])
    init = list(init_t) + quat_from_angle_axis(yaw + 0.02, [0, 0, 1])
    out = oracle.ceres3d_match(pairs, init_t, init, [1.0, 6.0], translation_weight=5.0,
                               rotation_weight=4e2, only_optimize_yaw=yaw_only,
                               max_num_iterations=12)
    assert out["final_cost"] < out["initial_cost"] and out["num_successful_steps"] >= 1
    assert np.linalg.norm(out["pose"][:3] - pos) < 0.25      # stays in the basin it started in
    if yaw_only:     # x and y of the rotation stay (numerically) untouched
        assert np.abs(out["pose"][4:6]).max() < 1e-12
    assert abs(np.linalg.norm(out["pose"][3:]) - 1.0) < 1e-6
    del truth_q


def test_refine_batch_checks_its_arguments_before_touching_a_device():
    """cmx_fast3d_refine_batch: the reference's CHECKs (ceres_scan_matcher_3d.cc:110,138,144) and
    the two-pair shape of the constraint refinement are argument errors, whatever the machine."""
    import ctypes as C
    from cartographer_amd import _lib
    L = _lib.lib()
    o = _lib.Ceres3DOptions()
    o.occupied_space_weight[0] = 1.0
    o.occupied_space_weight[1] = 1.0
    o.translation_weight, o.rotation_weight = 1.0, 1.0
    o.num_pairs, o.max_num_iterations = 1, 5
    data = _lib.NodeData3D()
    poses = (_lib.Pose3d * 1)()
    out = (_lib.Pose3d * 1)()
    handles = (C.c_void_p * 1)()
    call = lambda opt, n: L.cmx_fast3d_refine_batch(      # noqa: E731
        C.byref(opt), handles, n, None, C.cast(poses, C.c_void_p), C.byref(data),
        C.cast(out, C.c_void_p), None)
    assert call(o, 0) == _lib.INVALID_ARGUMENT          # num_pairs must be 2
    o.num_pairs = 2
    assert call(o, 0) == _lib.INVALID_ARGUMENT          # no point clouds
    o.translation_weight = 0.0
    assert call(o, 0) == _lib.INVALID_ARGUMENT
    assert b"translation_weight" in L.cmx_last_error()


# ---- InterpolatedGrid and the delta functors, on the reference's own unit tests ------------
def _interpolated(oracle, vox, resolution, xyz):
    """InterpolatedGrid::GetInterpolatedValue at every row of xyz, read off the occupied-space
    residuals of an identity pose: residual_i = w / sqrt(N) * (1 - value_i)."""
    xyz = np.ascontiguousarray(xyz, np.float32)
    r, _ = oracle.ceres3d_residuals([(xyz, resolution, vox)], (0, 0, 0), (1, 0, 0, 0),
                                    [0, 0, 0, 1, 0, 0, 0], [1.0], translation_weight=1.0,
                                    rotation_weight=1.0)
    return 1.0 - r[:len(xyz)] * math.sqrt(len(xyz))


def _interpolated_grid_fixture(synth):
    """InterpolatedGridTest (interpolated_grid_test.cc:28-48): hybrid_grid_(0.1f) with
    probability 1 at seven points."""
    grid = synth.HybridGrid(0.1)
    for p in POINTS:
        grid.set_probability(grid.get_cell_index(p), 1.0)
    return grid


def _lattice():
    """The test's loops: z, y, x advanced by resolution() (the FLOAT 0.1f) in double."""
    step = float(np.float32(0.1))

    def axis(lo, hi):
        out, v = [], lo
        while v < hi:
            out.append(v)
            v += step
        return out
    return axis(-8.0, -2.0), axis(1.0, 5.0), axis(-1.0, 3.0)


def test_reference_interpolated_grid_interpolates_grid_points(oracle, synth):
    """InterpolatedGridTest.InterpolatesGridPoints (interpolated_grid_test.cc:50-60): at every
    lattice point the interpolated value is the grid's probability, to 1e-6.  (The query points
    pass through float32, as a RangefinderPoint does; lattice coordinates of magnitude < 8 move
    by < 5e-7, within the reference's own tolerance on a piecewise cubic with zero slope at the
    cell centres.)"""
    grid = _interpolated_grid_fixture(synth)
    xs, ys, zs = _lattice()
    pts = np.array([(x, y, z) for z in zs[::4] for y in ys[::3] for x in xs], np.float64)
    # the seven occupied cells and their neighbours, whatever the stride above skipped
    near = np.array([p + d for p in POINTS.astype(np.float64)
                     for d in ((0, 0, 0), (0.1, 0, 0), (-0.1, 0, 0), (0, 0.1, 0), (0, 0, -0.1))])
    pts = np.vstack([pts, near])
    want = np.array([grid.get_probability(grid.get_cell_index(np.float32(p))) for p in pts])
    got = _interpolated(oracle, grid.voxels(), 0.1, pts)
    assert want.max() > 0.85 and want.min() < 0.15
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-6)


def test_reference_interpolated_grid_is_monotonic_between_grid_points(oracle, synth):
    """InterpolatedGridTest.MonotonicBehaviorBetweenGridPointsInX (:62-87): between two lattice
    points of different probability the value moves monotonically towards the second."""
    grid = _interpolated_grid_fixture(synth)
    res = float(np.float32(0.1))
    step = res / 10.0
    checked = 0
    for p in POINTS.astype(np.float64):
        for x0 in (p[0] - res, p[0]):                  # the rise into and the fall out of the cell
            start = grid.get_probability(grid.get_cell_index(np.float32([x0, p[1], p[2]])))
            nxt = grid.get_probability(grid.get_cell_index(np.float32([x0 + res, p[1], p[2]])))
            if abs(nxt - start) < 1e-6:
                continue
            samples, s = [], step
            while s < res - 2 * step:
                samples.append(s)
                s += step
            a = _interpolated(oracle, grid.voxels(), 0.1,
                              [(x0 + s, p[1], p[2]) for s in samples])
            b = _interpolated(oracle, grid.voxels(), 0.1,
                              [(x0 + s + step, p[1], p[2]) for s in samples])
            assert np.all((nxt - start) * (b - a) > 0.0), (p, x0)
            checked += 1
    assert checked >= 8


def _rotation_delta_squared_cost(oracle, rotation, scaling_factor, target):
    """RotationDeltaCostFunctor3D's squared cost (rotation_delta_cost_functor_3d_test.cc:30-47):
    the last three residuals of the problem."""
    pts = np.zeros((1, 3), np.float32)
    vox = np.zeros(0, dtype=[("x", np.int32), ("y", np.int32), ("z", np.int32),
                             ("value", np.uint16), ("pad", np.uint16)])
    r, _ = oracle.ceres3d_residuals([(pts, 1.0, vox)], (0, 0, 0), target,
                                    [0, 0, 0, *rotation], [1.0], translation_weight=1.0,
                                    rotation_weight=scaling_factor)
    return float(np.sum(r[-3:] ** 2))


def test_reference_rotation_delta_cost_functor(oracle):
    """RotationDeltaCostFunctor3DTest.SameRotationGivesZeroCost / .ComputesCorrectCost
    (rotation_delta_cost_functor_3d_test.cc:49-85), precision 1e-8."""
    ident = np.array([1.0, 0, 0, 0])
    unit = lambda v: np.asarray(v, np.float64) / np.linalg.norm(v)      # noqa: E731
    assert abs(_rotation_delta_squared_cost(oracle, ident, 1.0, ident)) < 1e-8
    rot = np.array(quat_from_angle_axis(0.9, unit([0.2, 0.1, 0.3])))
    assert abs(_rotation_delta_squared_cost(oracle, rot, 1.0, rot)) < 1e-8
    scaling, angle = 1.2, 0.8
    rotation = np.array(quat_from_angle_axis(angle, unit([0.2, 0.1, 0.8])))
    target = np.array(quat_from_angle_axis(0.2, unit([-0.5, 0.3, 0.4])))
    expected = (scaling * math.sin(angle / 2.0)) ** 2
    assert abs(_rotation_delta_squared_cost(oracle, rotation, scaling, ident) - expected) < 1e-8
    assert abs(_rotation_delta_squared_cost(oracle, quat_mul(target, rotation), scaling, target)
               - expected) < 1e-8
    assert abs(_rotation_delta_squared_cost(oracle, quat_mul(rotation, target), scaling, target)
               - expected) < 1e-8
