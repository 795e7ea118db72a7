"""tests/golden: results of the REFERENCE'S OWN sources (oracle/_ref, generated here by
tests/golden/make_reference_results.py and make_tsdf_fixture.py, committed) on seeded workloads.
CPU part: the oracle reproduces them exactly, and -- where oracle/_ref can be built -- regenerating
them gives the committed bytes (so the files cannot drift from the reference).  The GPU part is
tests/test_gpu_zz_new.py.
"""
import json
import os
import sys

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
if GOLDEN not in sys.path:
    sys.path.insert(0, GOLDEN)


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(GOLDEN, "reference_results.json")) as f:
        return json.load(f)


def test_golden_files_regenerate_identically(oracle, golden):
    if oracle.ref_lib() is None:
        pytest.skip("reference tree not available and oracle/_ref not prebuilt")
    import make_reference_results
    import make_tsdf_fixture
    assert make_reference_results.build() == golden
    fresh = make_tsdf_fixture.build()
    stored = np.load(os.path.join(GOLDEN, "rt2d_tsdf_fixture.npz"))
    for key, value in fresh.items():
        np.testing.assert_array_equal(np.asarray(value), stored[key], err_msg=key)


def test_tsdf_fixture_meets_the_reference_tests_expectations(oracle):
    """ScorePerfectHighResolutionCandidateTSDF / ScorePartiallyCorrect...TSDF
    (real_time_correlative_scan_matcher_2d_test.cc:143-160, :181-199) on the fixture the
    reference's own TSDFRangeDataInserter2D produced."""
    import workloads
    t = workloads.rt2d_tsdf()
    assert t["tsd"].shape == (40, 40)                     # GrowLimits doubled the 20x20 grid once
    args = (t["tsd"], t["weight"], t["res"], t["max_x"], t["max_y"], t["truncation"],
            t["max_weight"], [0, 0, 0], t["cloud"])
    r = oracle.rt2d_match_tsdf(*args, 0.0, 0.0, 0.0, 0.0, want_scores=True)
    assert r["num_candidates"] == 1
    assert 0.95 < r["scores"][0] and abs(r["scores"][0] - 1.0) < 1e-1
    r = oracle.rt2d_match_tsdf(*args, 0.05, 0.0, 0.0, 0.0, want_scores=True)
    partial = r["scores"].reshape(3, 3)[1, 2]             # x_offset 0, y_offset +1
    assert 1.0 - 4.0 / (7.0 * 6.0) < partial < 1.0


def test_oracle_reproduces_the_golden_results(oracle, synth, golden):
    import workloads as w
    b = w.fast2d_bench(synth)
    lim = b["lim"]
    m = oracle.FastCorrelativeScanMatcher2D(b["cells"], lim["resolution"], lim["max_x"],
                                            lim["max_y"], b["depth"])
    r = m.match_full_submap(b["scan"], 0.6)
    g = golden["fast2d_full_submap"]
    assert r["found"] and np.float32(r["score"]) == np.float32(g["score"])
    np.testing.assert_array_equal(r["pose"], g["pose"])
    assert abs(g["pose"][0] - b["truth"][0]) < 0.1 and abs(g["pose"][1] - b["truth"][1]) < 0.1
    r = m.match(b["init"], b["scan"], 0.55)
    g = golden["fast2d_windowed"]
    assert r["found"] and np.float32(r["score"]) == np.float32(g["score"])
    np.testing.assert_array_equal(r["pose"], g["pose"])
    assert not m.match(b["init"], b["scan"], 0.99)["found"]
    assert not golden["fast2d_windowed_unreachable"]["found"]

    c = w.rt2d_c1(synth)
    lim = c["lim"]
    r = oracle.rt2d_match(c["cells"], lim["resolution"], lim["max_x"], lim["max_y"], c["init"],
                          c["scan"], c["lin"], c["ang"], c["tw"], c["rw"])
    assert r["score"] == golden["rt2d_c1"]["score"]
    np.testing.assert_array_equal(r["pose"], golden["rt2d_c1"]["pose"])

    t = w.rt2d_tsdf()
    r = oracle.rt2d_match_tsdf(t["tsd"], t["weight"], t["res"], t["max_x"], t["max_y"],
                               t["truncation"], t["max_weight"], t["init"], t["cloud"], t["lin"],
                               t["ang"], t["tw"], t["rw"])
    assert r["score"] == golden["rt2d_tsdf"]["score"]
    np.testing.assert_array_equal(r["pose"], golden["rt2d_tsdf"]["pose"])

    d = w.rt3d(synth)
    r = oracle.rt3d_match(d["res"], d["vox"], d["init"], d["cloud"], d["lin"], d["ang"], d["tw"],
                          d["rw"])
    assert np.float32(r["score"]) == np.float32(golden["rt3d"]["score"])
    np.testing.assert_array_equal(r["pose"], golden["rt3d"]["pose"])

    f = w.fast3d(synth)
    o = f["options"]
    m3 = oracle.FastCorrelativeScanMatcher3D(f["res"], f["vox"], f["low_res"], f["low_vox"],
                                             f["hist"], o["depth"], o["frd"], o["min_rot"],
                                             o["min_low"], o["lin_xy"], o["lin_z"], o["ang"])
    r = m3.match(f["node_pose"], f["submap_pose"], f["gravity"], f["hi"], f["lo"],
                 f["scan_hist"], f["min_score"])
    g = golden["fast3d"]
    assert r["found"] and g["found"]
    for key in ("score", "rotational_score", "low_resolution_score"):
        assert np.float32(r[key]) == np.float32(g[key]), key
    np.testing.assert_array_equal(r["pose"], g["pose"])


def test_oracle_reproduces_the_c4_shaped_reference_result(oracle, synth):
    """tests/golden/rt3d_c4_reference.json (make_rt3d_c4_golden.py: the reference's own
    real_time_correlative_scan_matcher_3d.cc): the C4-SHAPED workload -- L = 5 (216 groups of
    2x2x2 translations per rotation), A = 3, tilted initial orientation, 4096 points, 456 533
    candidates -- reproduced bit for bit by the oracle (z slices on the host's threads).  The
    full C4 entry (1 771 561 candidates x 65 536 points, ten minutes of the reference on 8 cores)
    is only checked for presence here; the device is compared with both (-m gpu)."""
    import workloads as w
    with open(os.path.join(GOLDEN, "rt3d_c4_reference.json")) as f:
        g = json.load(f)
    assert g["rt3d_c4"]["num_candidates"] == 1331 * 1331 and g["rt3d_c4"]["num_points"] == 65536
    d = w.rt3d_c4_shaped(synth)
    r = oracle.rt3d_match(d["res"], d["vox"], d["init"], d["cloud"], d["lin"], d["ang"], d["tw"],
                          d["rw"], num_threads=max(2, os.cpu_count() or 2))
    s = g["rt3d_c4_shaped"]
    assert r["num_candidates"] == s["num_candidates"] == d["num_candidates"]
    assert np.float32(r["score"]) == np.float32(s["score"])
    np.testing.assert_array_equal(r["pose"], s["pose"])


def test_c4_golden_workload_is_the_one_bench_times(synth):
    """The C4 golden is only worth something if bench.py's Rt3DWorkload is that workload."""
    import bench
    import workloads as w
    wl = bench.Rt3DWorkload(None, 0)
    d = w.rt3d_c4(synth)
    assert np.array_equal(wl.cloud, d["cloud"]) and np.array_equal(wl.vox, d["vox"])
    assert list(wl.init.translation) + list(wl.init.rotation) == list(d["init"])
    assert (wl.m.options.linear_search_window, wl.m.options.angular_search_window) == \
        (d["lin"], d["ang"])
