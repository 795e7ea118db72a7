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
