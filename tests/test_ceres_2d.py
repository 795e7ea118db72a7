"""CeresScanMatcher2D: the oracle's restatement against the reference's own
known-answer test, its analytic Jacobian against finite differences, and the optimiser's basic
contracts.  Ceres itself is absent from /root/reference (third-party); see
oracle/oracle_ceres_2d.h for what is restated and what stays unpinned."""
import math

import numpy as np
import pytest


def _reference_test_grid(synth):
    """CeresScanMatcherTest fixture (ceres_scan_matcher_2d_test.cc:36-46): 20 x 20 cells of 1 m,
    max (10, 10), the cell containing (-3.5, 2.5) set to kMaxProbability; one point (-3, 2)."""
    g = synth.ProbabilityGrid(1.0, (10.0, 10.0), 20, 20)
    # MapLimits::GetCellIndex((-3.5, 2.5)): ix = lround((10 - 2.5) / 1 - 0.5) = 7,
    # iy = lround((10 + 3.5) / 1 - 0.5) = 13  (map_limits.h:69-76)
    g.set_probability(7, 13, 0.9)
    return g


@pytest.mark.parametrize("init", [(-0.5, 0.5), (-0.3, 0.5), (-0.45, 0.3), (-0.3, 0.3)])
def test_reference_ceres_scan_matcher_test(oracle, synth, init):
    """testPerfectEstimate / testOptimizeAlongX / AlongY / AlongXY
    (ceres_scan_matcher_2d_test.cc:97-111): IsNearly((-0.5, 0.5, 0), 1e-2), final cost within
    1e-2 of 0; options of the fixture (:49-59).  One tolerance for all four cases."""
    g = _reference_test_grid(synth)
    lim = g.limits
    cloud = np.array([[-3.0, 2.0, 0.0]], np.float32)
    out = oracle.ceres2d_match(g.cells, 1.0, lim["max_x"], lim["max_y"], init, (init[0], init[1], 0.0),
                               cloud, occupied_space_weight=1.0, translation_weight=0.1,
                               rotation_weight=1.5, use_nonmonotonic_steps=True,
                               max_num_iterations=50)
    assert abs(out["final_cost"]) < 1e-2
    # transform::IsNearly(expected_pose, 1e-2) (rigid_transform_test_helpers.h:32-46) is Eigen's
    # isApprox on the homogeneous 3 x 3 transforms: |a - b|_F <= 1e-2 * min(|a|_F, |b|_F) -- a
    # relative Frobenius bound (0.0187 here), not 1e-2 per component.  (Round 2 read it as the
    # latter and loosened one case; the minimum of this objective itself lies 1.2e-2 from the
    # expected pose in y for testOptimizeAlongY -- the rotation residual is cheaper than the
    # translation residual at the margin -- so no solver could meet a per-component 1e-2 there.)
    assert _is_nearly(out["pose"], (-0.5, 0.5, 0.0), 1e-2), out


def _is_nearly(pose, expected, epsilon):
    def m(p):
        c, s = math.cos(p[2]), math.sin(p[2])
        return np.array([[c, -s, p[0]], [s, c, p[1]], [0, 0, 1.0]])
    a, b = m(pose), m(expected)
    return np.linalg.norm(a - b) <= epsilon * min(np.linalg.norm(a), np.linalg.norm(b))


def test_jacobian_matches_finite_differences(oracle, synth):
    cells, lim, world = synth.make_submap(5, 120, 100, 0.05, 12, 400, 8.0, 0.01)
    pose = world.free_pose(3, 0.5)
    scan = world.scan(pose, 150, 8.0, 0.01, 2)
    x = np.array([pose[0] + 0.03, pose[1] - 0.02, pose[2] + 0.01])
    args = (cells, 0.05, lim["max_x"], lim["max_y"], x[:2] + 0.01, x[2] - 0.02)
    r, J = oracle.ceres2d_residuals(*args, x, scan)
    assert r.shape == (len(scan) + 3,) and J.shape == (len(scan) + 3, 3)
    # (the interpolation argument carries kPadding = INT_MAX / 4 cells, so its fractional part
    # has ~6e-8 cells of f64 resolution: central differences need a step well above that)
    for k, h in enumerate((2e-4, 2e-4, 2e-5)):
        d = np.zeros(3); d[k] = h
        rp, _ = oracle.ceres2d_residuals(*args, x + d, scan)
        rm, _ = oracle.ceres2d_residuals(*args, x - d, scan)
        # (Catmull-Rom is C1: next to a cell boundary the second derivative jumps and a central
        # difference is only first-order accurate there)
        np.testing.assert_allclose((rp - rm) / (2 * h), J[:, k], atol=3e-3, rtol=2e-3)


@pytest.mark.parametrize("nonmonotonic", [False, True])
def test_match_decreases_the_cost_and_recovers_the_pose(oracle, synth, nonmonotonic):
    cells, lim, world = synth.make_submap(9, 200, 200, 0.05, 25, 800, 10.0, 0.01)
    truth = world.free_pose(17, 0.5)
    scan = world.scan(truth, 300, 10.0, 0.01, 4)
    init = (truth[0] + 0.04, truth[1] - 0.03, truth[2] + 0.015)
    out = oracle.ceres2d_match(cells, 0.05, lim["max_x"], lim["max_y"], init[:2], init, scan,
                               occupied_space_weight=20.0, translation_weight=10.0,
                               rotation_weight=1.0, use_nonmonotonic_steps=nonmonotonic,
                               max_num_iterations=10)          # pose_graph.lua:30-39
    assert out["final_cost"] < out["initial_cost"]
    assert out["num_successful_steps"] >= 1
    err0 = math.hypot(init[0] - truth[0], init[1] - truth[1])
    err1 = math.hypot(out["pose"][0] - truth[0], out["pose"][1] - truth[1])
    assert err1 < err0 and err1 < 0.03


def test_reference_occupied_space_cost_function_smoke_test(oracle):
    """OccupiedSpaceCostFunction2DTest.SmokeTest (occupied_space_cost_function_2d_test.cc:32-53):
    an all-unknown 2 x 2 grid at resolution 1 with max (1, 1), one point at the origin, pose 0,
    scaling factor 1: the residual is exactly kMaxProbability (1 - kMinCorrespondenceCost)."""
    cells = np.zeros((2, 2), np.uint16)
    r, J = oracle.ceres2d_residuals(cells, 1.0, 1.0, 1.0, np.zeros(2), 0.0, np.zeros(3),
                                    np.zeros((1, 3), np.float32), occupied_space_weight=1.0,
                                    translation_weight=1.0, rotation_weight=1.0)
    k_max_probability = float(np.float32(1.0) - np.float32(0.1))   # 1.f - kMinProbability, in float
    assert r[0] == k_max_probability
    assert np.all(J[0] == 0.0)                               # a constant field
