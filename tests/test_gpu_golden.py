"""The device against tests/golden/reference_results.json: what the REFERENCE'S OWN scan-matcher
sources returned (oracle/_ref, generated in the build container by
tests/golden/make_reference_results.py) on the seeded workloads of tests/golden/workloads.py --
the bench workload, BASELINE config C1, the reference test's TSDF fixture, a 3D real-time match
and a 3D loop-closure match.  Scores bit-equal; 2D poses to 1e-12 (composed in f64 on the host),
3D poses exact.
"""
import json
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
if GOLDEN not in sys.path:
    sys.path.insert(0, GOLDEN)


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(GOLDEN, "reference_results.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def sm():
    from cartographer_amd import _lib, scan_matching
    assert _lib.lib().cmx_device_count() >= 1, "no HIP device: these tests need the GPU"
    return scan_matching


def _pose2(p):
    return [p.x, p.y, p.theta]


def test_fast2d_bench_workload_equals_the_reference(sm, synth, golden):
    import workloads as w
    b = w.fast2d_bench(synth)
    lim = b["lim"]
    grid = sm.Grid2D(b["cells"], lim["resolution"], lim["max_x"], lim["max_y"])
    m = sm.FastCorrelativeScanMatcher2D(grid, b["depth"])
    found, score, pose = m.match_full_submap(b["scan"], 0.6)
    g = golden["fast2d_full_submap"]
    assert found and np.float32(score) == np.float32(g["score"])
    np.testing.assert_allclose(_pose2(pose), g["pose"], rtol=0, atol=1e-12)
    found, score, pose = m.match(sm.Rigid2d(*b["init"]), b["scan"], 0.55)
    g = golden["fast2d_windowed"]
    assert found and np.float32(score) == np.float32(g["score"])
    np.testing.assert_allclose(_pose2(pose), g["pose"], rtol=0, atol=1e-12)
    assert not m.match(sm.Rigid2d(*b["init"]), b["scan"], 0.99)[0]


def test_rt2d_c1_equals_the_reference(sm, synth, golden):
    import workloads as w
    c = w.rt2d_c1(synth)
    lim = c["lim"]
    grid = sm.Grid2D(c["cells"], lim["resolution"], lim["max_x"], lim["max_y"])
    m = sm.RealTimeCorrelativeScanMatcher2D(c["lin"], c["ang"], c["tw"], c["rw"])
    score, pose = m.match(sm.Rigid2d(*c["init"]), c["scan"], grid)
    assert score == golden["rt2d_c1"]["score"]
    np.testing.assert_allclose(_pose2(pose), golden["rt2d_c1"]["pose"], rtol=0, atol=1e-12)


def test_rt2d_tsdf_fixture_equals_the_reference(sm, golden):
    """The TSDF the reference's own TSDFRangeDataInserter2D built for its real-time matcher test
    (tests/golden/rt2d_tsdf_fixture.npz), matched with that test's options."""
    import workloads as w
    t = w.rt2d_tsdf()
    grid = sm.TSDF2D(t["tsd"], t["weight"], t["res"], t["max_x"], t["max_y"], t["truncation"],
                     t["max_weight"])
    m = sm.RealTimeCorrelativeScanMatcher2D(t["lin"], t["ang"], t["tw"], t["rw"])
    score, pose = m.match(sm.Rigid2d(*t["init"]), t["cloud"], grid)
    assert score == golden["rt2d_tsdf"]["score"]
    np.testing.assert_allclose(_pose2(pose), golden["rt2d_tsdf"]["pose"], rtol=0, atol=1e-12)
    # ScorePerfectHighResolutionCandidateTSDF (..._2d_test.cc:143-160): a zero window scores the
    # one candidate (0, 0, 0)
    m0 = sm.RealTimeCorrelativeScanMatcher2D(0.0, 0.0, 0.0, 0.0)
    score0, _ = m0.match(sm.Rigid2d(0.0, 0.0, 0.0), t["cloud"], grid)
    assert 0.95 < score0 and abs(score0 - 1.0) < 1e-1


def test_rt3d_equals_the_reference(synth, golden):
    import workloads as w
    from cartographer_amd import scan_matching_3d as sm3
    d = w.rt3d(synth)
    m = sm3.RealTimeCorrelativeScanMatcher3D(d["lin"], d["ang"], d["tw"], d["rw"])
    score, pose = m.match(sm3.Rigid3d(tuple(d["init"][:3]), tuple(d["init"][3:])), d["cloud"],
                          d["res"], d["vox"])
    assert np.float32(score) == np.float32(golden["rt3d"]["score"])
    np.testing.assert_array_equal(list(pose.translation) + list(pose.rotation),
                                  golden["rt3d"]["pose"])


def test_fast3d_equals_the_reference(synth, golden):
    import workloads as w
    from cartographer_amd import scan_matching_3d as sm3
    f = w.fast3d(synth)
    o = f["options"]
    m = sm3.FastCorrelativeScanMatcher3D(
        f["res"], f["vox"], f["grid_size"], f["low_res"], f["low_vox"], f["hist"],
        branch_and_bound_depth=o["depth"], full_resolution_depth=o["frd"],
        min_rotational_score=o["min_rot"], min_low_resolution_score=o["min_low"],
        linear_xy_search_window=o["lin_xy"], linear_z_search_window=o["lin_z"],
        angular_search_window=o["ang"])
    data = sm3.TrajectoryNodeData(f["hi"], f["lo"], f["scan_hist"], tuple(f["gravity"]))
    got = m.match(sm3.Rigid3d(tuple(f["node_pose"][:3]), tuple(f["node_pose"][3:])),
                  sm3.Rigid3d(tuple(f["submap_pose"][:3]), tuple(f["submap_pose"][3:])), data,
                  f["min_score"])
    g = golden["fast3d"]
    assert got is not None and g["found"]
    for key in ("score", "rotational_score", "low_resolution_score"):
        assert np.float32(got[key]) == np.float32(g[key]), key
    p = got["pose_estimate"]
    np.testing.assert_array_equal(list(p.translation) + list(p.rotation), g["pose"])
