"""oracle/_ref/libref_ceres.so: the REFERENCE'S OWN Ceres-side sources -- occupied_space_cost_function_2d.cc,
ceres_scan_matcher_2d.cc, translation_/rotation_delta_cost_functor_{2d,3d}.h, interpolated_grid.h,
occupied_space_cost_function_3d.h, intensity_cost_function_3d.{h,cc}, rotation_parameterization.h,
ceres_pose.cc, ceres_scan_matcher_3d.cc -- compiled unmodified from /root/reference
(oracle/Makefile `ref_ceres`) over the stand-in ceres/ headers of oracle/ref_shims (Jets,
AutoDiffCostFunction, parameterizations, HuberLoss, Problem, a dense trust-region solver written
from Ceres' published algorithm; ceres-solver itself is absent from /root/reference), against the
oracle's closed-form restatement (oracle/oracle_ceres_{2d,3d}.cc).

What this pins: every line of the reference's cost functors (coordinate mapping with kPadding, the
scaling factors, the tricubic InterpolatedGrid, quaternion algebra through Jets), the problem
set-up of CeresScanMatcher2D / 3D::Match (weights / sqrt(N), targets, parameterizations, which
pose comes back), and -- two independently written solvers (Householder QR on Jets' Jacobians here,
normal equations on closed-form derivatives in the oracle) walking the same iterates.
What it does NOT pin: Ceres' own iterates (the solver stays a restatement on both sides).

Skipped where neither /root/reference nor a prebuilt oracle/_ref/libref_ceres.so exists.
"""
import math

import numpy as np
import pytest

from test_ceres_3d import CASES, POINTS, fixture, is_nearly
from test_oracle_reference_pins_3d import quat_from_angle_axis


@pytest.fixture(scope="module")
def refc(oracle):
    lib = oracle.ref_ceres_lib()
    if lib is None:
        pytest.skip("reference tree not available and oracle/_ref/libref_ceres.so not prebuilt")
    return lib


# ---------------------------------------------------------------------------- 2D ---
def _workload_2d(synth, seed, n_points):
    cells, lim, world = synth.make_submap(seed, 120, 100, 0.05, 12, 400, 8.0, 0.01)
    pose = world.free_pose(seed + 3, 0.5)
    scan = world.scan(pose, n_points, 8.0, 0.01, 2)
    return cells, lim, np.asarray(pose), scan


@pytest.mark.parametrize("seed", [5, 6, 7])
def test_2d_residuals_and_jacobians_equal_the_reference_cost_functions(refc, oracle, synth, seed):
    """CreateOccupiedSpaceCostFunction2D (occupied_space_cost_function_2d.cc:39-108) +
    TranslationDeltaCostFunctor2D + RotationDeltaCostFunctor2D evaluated through Jets, against the
    oracle's closed-form derivatives, on random poses: 1e-12."""
    cells, lim, pose, scan = _workload_2d(synth, seed, 150)
    rng = np.random.default_rng(seed)
    for _ in range(6):
        x = pose + rng.uniform(-1, 1, 3) * np.array([0.3, 0.3, 0.2])
        args = (cells, 0.05, lim["max_x"], lim["max_y"], x[:2] + 0.01, x[2] - 0.02)
        kw = dict(occupied_space_weight=20.0, translation_weight=10.0, rotation_weight=1.0)
        r, J = oracle.ceres2d_residuals(*args, x, scan, **kw)
        rr, JJ = oracle.ceres2d_residuals(*args, x, scan, reference=True, **kw)
        np.testing.assert_allclose(r, rr, rtol=0, atol=1e-12)
        np.testing.assert_allclose(J, JJ, rtol=1e-12, atol=1e-12)
        assert np.abs(J[:-3]).max() > 1.0                      # a non-trivial field


def test_2d_residuals_outside_the_grid_equal_the_reference(refc, oracle, synth):
    """Points that land in the padding (GridArrayAdapter returns kMaxCorrespondenceCost there)."""
    cells, lim, pose, scan = _workload_2d(synth, 5, 60)
    x = np.array([lim["max_x"] + 1.0, lim["max_y"] - 2.0, 0.4])
    args = (cells, 0.05, lim["max_x"], lim["max_y"], x[:2], x[2])
    r, J = oracle.ceres2d_residuals(*args, x, scan)
    rr, JJ = oracle.ceres2d_residuals(*args, x, scan, reference=True)
    np.testing.assert_allclose(r, rr, rtol=0, atol=1e-12)
    np.testing.assert_allclose(J, JJ, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("nonmonotonic", [False, True])
def test_2d_match_equals_the_reference_match(refc, oracle, synth, nonmonotonic):
    """CeresScanMatcher2D::Match itself (ceres_scan_matcher_2d.cc:63-107) over the stand-in
    solver against the restatement: same pose, costs and step counts."""
    cells, lim, world = synth.make_submap(9, 200, 200, 0.05, 25, 800, 10.0, 0.01)
    truth = world.free_pose(17, 0.5)
    scan = world.scan(truth, 300, 10.0, 0.01, 4)
    init = (truth[0] + 0.04, truth[1] - 0.03, truth[2] + 0.015)
    kw = dict(occupied_space_weight=20.0, translation_weight=10.0, rotation_weight=1.0,
              use_nonmonotonic_steps=nonmonotonic, max_num_iterations=10)
    a = oracle.ceres2d_match(cells, 0.05, lim["max_x"], lim["max_y"], init[:2], init, scan, **kw)
    b = oracle.ceres2d_match(cells, 0.05, lim["max_x"], lim["max_y"], init[:2], init, scan,
                             reference=True, **kw)
    np.testing.assert_allclose(a["pose"], b["pose"], rtol=0, atol=1e-9)
    assert a["initial_cost"] == pytest.approx(b["initial_cost"], rel=1e-12)
    assert a["final_cost"] == pytest.approx(b["final_cost"], rel=1e-9)
    assert (a["num_successful_steps"], a["num_unsuccessful_steps"], a["termination"]) == \
        (b["num_successful_steps"], b["num_unsuccessful_steps"], b["termination"])


def _is_nearly_2d(pose, expected, epsilon):
    """transform::IsNearly for Rigid2d (rigid_transform_test_helpers.h:32-46): Eigen's isApprox on
    the homogeneous 3 x 3 transforms, |a - b|_F <= epsilon * min(|a|_F, |b|_F)."""
    def m(p):
        c, s = math.cos(p[2]), math.sin(p[2])
        return np.array([[c, -s, p[0]], [s, c, p[1]], [0, 0, 1.0]])
    a, b = m(pose), m(expected)
    return np.linalg.norm(a - b) <= epsilon * min(np.linalg.norm(a), np.linalg.norm(b))


@pytest.mark.parametrize("init", [(-0.5, 0.5), (-0.3, 0.5), (-0.45, 0.3), (-0.3, 0.3)])
def test_reference_ceres_scan_matcher_test_through_the_reference_source(refc, oracle, synth, init):
    """CeresScanMatcherTest (ceres_scan_matcher_2d_test.cc:34-111) run on the reference's own
    ceres_scan_matcher_2d.cc: final cost within 1e-2 of 0 and IsNearly(expected, 1e-2)."""
    from test_ceres_2d import _reference_test_grid
    g = _reference_test_grid(synth)
    lim = g.limits
    cloud = np.array([[-3.0, 2.0, 0.0]], np.float32)
    out = oracle.ceres2d_match(g.cells, 1.0, lim["max_x"], lim["max_y"], init,
                               (init[0], init[1], 0.0), cloud, occupied_space_weight=1.0,
                               translation_weight=0.1, rotation_weight=1.5,
                               use_nonmonotonic_steps=True, max_num_iterations=50, reference=True)
    assert abs(out["final_cost"]) < 1e-2
    assert _is_nearly_2d(out["pose"], (-0.5, 0.5, 0.0), 1e-2), out


# ---------------------------------------------------------------------------- 3D ---
def _workload_3d(synth, seed):
    grid, world = synth.make_submap_3d(seed, 0.1, (8.0, 8.0, 4.0), 4, 8, 96)
    low, _ = synth.make_submap_3d(seed, 0.3, (8.0, 8.0, 4.0), 4, 8, 96)
    pos = world.free_position(seed + 1, 0.5)
    full = world.scan(pos, 0.3, 8, 64, seed=9)
    hi, lo = full[::3].copy(), full[::11].copy()
    return [(hi, 0.1, grid.voxels()), (lo, 0.3, low.voxels())], pos


@pytest.mark.parametrize("seed", [5, 8])
def test_3d_residuals_and_jacobians_equal_the_reference_cost_functions(refc, oracle, synth, seed):
    """OccupiedSpaceCostFunction3D over InterpolatedGrid (occupied_space_cost_function_3d.h:66-97,
    interpolated_grid.h:36-151), TranslationDeltaCostFunctor3D, RotationDeltaCostFunctor3D
    through Jets, against the oracle's closed forms: residuals and the 7-column ambient
    Jacobians on random poses, 1e-12 (relative to the Jacobian's scale)."""
    pairs, pos = _workload_3d(synth, seed)
    rng = np.random.default_rng(seed)
    for _ in range(4):
        q = np.array(quat_from_angle_axis(rng.uniform(0.05, 0.6), rng.uniform(-1, 1, 3)))
        pose = np.array(list(pos + rng.uniform(-0.05, 0.05, 3)) + list(q))
        target_q = quat_from_angle_axis(0.27, [0.1, 0.0, 1.0])
        weights = [1.0, 6.0]
        r, J = oracle.ceres3d_residuals(pairs, pos, target_q, pose, weights)
        rr, JJ = oracle.ceres3d_residuals(pairs, pos, target_q, pose, weights, reference=True)
        np.testing.assert_allclose(r, rr, rtol=0, atol=1e-12)
        np.testing.assert_allclose(J, JJ, rtol=0, atol=1e-12 * max(1.0, np.abs(J).max()))
        assert np.abs(J[:-6]).max() > 0.1


@pytest.mark.parametrize("yaw_only,nonmonotonic", [(False, False), (True, False), (False, True)])
def test_3d_match_equals_the_reference_match(refc, oracle, synth, yaw_only, nonmonotonic):
    """CeresScanMatcher3D::Match itself (ceres_scan_matcher_3d.cc:90-156: CeresPose,
    QuaternionParameterization or AutoDiffLocalParameterization<YawOnlyQuaternionPlus>, the
    residual blocks, Solve, ToRigid) over the stand-in solver against the restatement."""
    grid, world = synth.make_submap_3d(7, 0.1, (8.0, 8.0, 4.0), 4, 8, 96)
    low, _ = synth.make_submap_3d(7, 0.3, (8.0, 8.0, 4.0), 4, 8, 96)
    pos = world.free_position(8, 0.5)
    yaw = 0.4
    full = world.scan(pos, yaw, 16, 128, seed=3)
    hi, lo = full[::2].copy(), full[::9].copy()
    pairs = [(hi, 0.1, grid.voxels()), (lo, 0.3, low.voxels())]
    init_t = pos + np.array([0.04, -0.03, 0.0 if yaw_only else 0.02])
    init = list(init_t) + quat_from_angle_axis(yaw + 0.02, [0.05, -0.02, 1.0])
    kw = dict(translation_weight=5.0, rotation_weight=4e2, only_optimize_yaw=yaw_only,
              use_nonmonotonic_steps=nonmonotonic, max_num_iterations=12)
    a = oracle.ceres3d_match(pairs, init_t, init, [1.0, 6.0], **kw)
    b = oracle.ceres3d_match(pairs, init_t, init, [1.0, 6.0], reference=True, **kw)
    np.testing.assert_allclose(a["pose"], b["pose"], rtol=0, atol=1e-9)
    assert a["initial_cost"] == pytest.approx(b["initial_cost"], rel=1e-12)
    assert a["final_cost"] == pytest.approx(b["final_cost"], rel=1e-9)
    assert (a["num_successful_steps"], a["num_unsuccessful_steps"], a["termination"]) == \
        (b["num_successful_steps"], b["num_unsuccessful_steps"], b["termination"])


@pytest.mark.parametrize("name,t,q", CASES)
def test_reference_ceres_scan_matcher_3d_test_through_the_reference_source(refc, oracle, synth,
                                                                           name, t, q):
    """CeresScanMatcher3DTest (ceres_scan_matcher_3d_test.cc:36-111, its probability-grid part) run
    on the reference's own ceres_scan_matcher_3d.cc."""
    vox = fixture(synth, POINTS)
    init = list(t) + list(q)
    out = oracle.ceres3d_match([(POINTS, 1.0, vox)], t, init, [1.0], translation_weight=0.01,
                               rotation_weight=0.1, use_nonmonotonic_steps=True,
                               max_num_iterations=10, reference=True)
    assert abs(out["final_cost"]) < 1e-2, out
    assert is_nearly(out["pose"], [-1.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0], 3e-2), out
    mine = oracle.ceres3d_match([(POINTS, 1.0, vox)], t, init, [1.0], translation_weight=0.01,
                                rotation_weight=0.1, use_nonmonotonic_steps=True,
                                max_num_iterations=10)
    np.testing.assert_allclose(mine["pose"], out["pose"], rtol=0, atol=1e-8)



# ------------------------------------------------------------- intensity (f1, 3D) ---
def _intensity_world(synth, oracle, seed):
    """A probability grid from a synthetic room plus an intensity grid over the same voxels:
    AddIntensity-style (count, sum) records with a smooth intensity field and per-point
    intensities that partly exceed the threshold."""
    grid, world = synth.make_submap_3d(seed, 0.1, (8.0, 8.0, 4.0), 4, 8, 96)
    vox = grid.voxels()
    rng = np.random.default_rng(seed)
    iv = np.zeros(len(vox), oracle.INTENSITY_VOXEL_DTYPE)
    iv["x"], iv["y"], iv["z"] = vox["x"], vox["y"], vox["z"]
    iv["count"] = rng.integers(1, 5, len(vox))
    field = 60.0 + 25.0 * np.sin(0.21 * vox["x"]) + 15.0 * np.cos(0.17 * vox["y"] + 0.3 * vox["z"])
    iv["sum"] = (field * iv["count"]).astype(np.float32)
    iv["count"][::17] = 0                                  # cells that were never hit: intensity 0
    pos = world.free_position(seed + 1, 0.5)
    cloud = world.scan(pos, 0.3, 8, 64, seed=9)[::2].copy()
    intensities = rng.uniform(20.0, 140.0, len(cloud)).astype(np.float32)
    return vox, iv, pos, cloud, intensities


def test_reference_intensity_cost_function_smoke_test(refc, oracle):
    """IntensityCostFunction3DTest.SmokeTest (intensity_cost_function_3d_test.cc:36-64) on the
    reference's own intensity_cost_function_3d.{h,cc} and on the restatement: residuals
    (0, -100, 0)."""
    cloud = np.array([[0, 0, 0], [1, 1, 1], [2, 2, 2]], np.float32)
    intensities = np.array([50.0, 100.0, 150.0], np.float32)
    iv = np.zeros(1, oracle.INTENSITY_VOXEL_DTYPE)
    iv["count"], iv["sum"] = 1, 50.0                        # AddIntensity(cell of (0,0,0), 50)
    pose = [0, 0, 0, 1, 0, 0, 0]
    for reference in (True, False):
        r, _ = oracle.intensity3d_residuals(1.0, 100.0, cloud, intensities, 0.3, iv, pose,
                                            reference=reference)
        np.testing.assert_allclose(r, [0.0, -100.0, 0.0], rtol=0, atol=1e-9)


@pytest.mark.parametrize("seed", [5, 8])
def test_intensity_residuals_and_jacobians_equal_the_reference(refc, oracle, synth, seed):
    """IntensityCostFunction3D over InterpolatedGrid<IntensityHybridGrid> through Jets against
    the restatement: residuals and 7-column Jacobians on random poses, points above the
    threshold included (residual and derivative zero)."""
    _, iv, pos, cloud, intensities = _intensity_world(synth, oracle, seed)
    rng = np.random.default_rng(seed)
    for _ in range(4):
        q = np.array(quat_from_angle_axis(rng.uniform(0.05, 0.6), rng.uniform(-1, 1, 3)))
        pose = np.array(list(pos * 0 + rng.uniform(-0.05, 0.05, 3)) + list(q))
        r, J = oracle.intensity3d_residuals(0.7, 100.0, cloud, intensities, 0.1, iv, pose)
        rr, JJ = oracle.intensity3d_residuals(0.7, 100.0, cloud, intensities, 0.1, iv, pose,
                                              reference=True)
        np.testing.assert_allclose(r, rr, rtol=0, atol=1e-10)
        np.testing.assert_allclose(J, JJ, rtol=0, atol=1e-10 * max(1.0, np.abs(J).max()))
        above = intensities > 100.0
        assert above.any() and (~above).any()
        assert np.all(r[above] == 0.0) and np.all(J[above] == 0.0) and np.abs(r[~above]).max() > 1.0


@pytest.mark.parametrize("huber_scale", [0.3, 1e6])
def test_3d_match_with_intensity_equals_the_reference_match(refc, oracle, synth, huber_scale):
    """CeresScanMatcher3D::Match with an IntensityCostFunction3D block under ceres::HuberLoss
    (ceres_scan_matcher_3d.cc:118-137): huber_scale 0.3 puts the block in the outlier region
    (rho' < 1), 1e6 leaves it quadratic.  The reference's own source against the restatement."""
    vox, iv, pos, cloud, intensities = _intensity_world(synth, oracle, 7)
    low, _ = synth.make_submap_3d(7, 0.3, (8.0, 8.0, 4.0), 4, 8, 96)
    pairs = [(cloud, 0.1, vox, intensities, iv, (0.5, huber_scale, 100.0)),
             (cloud[::4].copy(), 0.3, low.voxels())]
    init_t = np.array([0.04, -0.03, 0.02])
    init = list(init_t) + quat_from_angle_axis(0.02, [0.05, -0.02, 1.0])
    kw = dict(translation_weight=5.0, rotation_weight=4e2, max_num_iterations=12)
    a = oracle.ceres3d_match_intensity(pairs, init_t, init, [1.0, 6.0], **kw)
    b = oracle.ceres3d_match_intensity(pairs, init_t, init, [1.0, 6.0], reference=True, **kw)
    np.testing.assert_allclose(a["pose"], b["pose"], rtol=0, atol=1e-9)
    assert a["initial_cost"] == pytest.approx(b["initial_cost"], rel=1e-12)
    assert a["final_cost"] == pytest.approx(b["final_cost"], rel=1e-9)
    assert (a["num_successful_steps"], a["num_unsuccessful_steps"], a["termination"]) == \
        (b["num_successful_steps"], b["num_unsuccessful_steps"], b["termination"])
    # the intensity block matters: without it the match ends elsewhere
    plain = oracle.ceres3d_match([p[:3] for p in pairs], init_t, init, [1.0, 6.0], **kw)
    assert np.abs(plain["pose"] - a["pose"]).max() > 1e-6
