"""GPU tests added after the last run on hardware; the file sorts last so that `pytest -x` reaches
them only after every test that has already passed on an MI355X.

Part 1 -- the device against tests/golden/reference_results.json: what the REFERENCE'S OWN scan-matcher
sources returned (oracle/_ref, generated in the build container by
tests/golden/make_reference_results.py) on the seeded workloads of tests/golden/workloads.py --
the bench workload, BASELINE config C1, the reference test's TSDF fixture, a 3D real-time match
and a 3D loop-closure match.  Scores bit-equal; 2D poses to 1e-12 (composed in f64 on the host),
3D poses exact.

Part 2 -- known answers of the reference's own tests on the device: RangeDataInserterTest2D on the
device inserter, and the found case of the synthetic 3D loop-closure world.
"""
import json
import math
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


# ---------------------------------------------------------------------------- part 2
@pytest.fixture(scope="module")
def g2():
    from cartographer_amd import _lib, grid_2d
    assert _lib.lib().cmx_device_count() >= 1, "no HIP device: these tests need the GPU"
    return grid_2d


def _assert_same(dev, host):
    assert dev.limits == host.limits
    np.testing.assert_array_equal(dev.cells, host.cells)


class _DeviceGridView:
    """Gives the reference-test checker of test_oracle_reference_pins the surface it reads."""

    def __init__(self, dev, oracle):
        self._dev = dev
        self._v2c = oracle.value_tables()[1]        # kValueToCorrespondenceCost

    limits = property(lambda self: self._dev.limits)
    cells = property(lambda self: self._dev.cells)

    def get_probability(self, ix, iy):              # probability_grid.cc:78-83
        return float(np.float32(1) - self._v2c[self._dev.cells[iy, ix]])


def test_reference_range_data_inserter_2d_test(g2, synth, oracle):
    """RangeDataInserterTest2D.InsertPointCloud / ProbabilityProgression
    (mapping/2d/range_data_inserter_2d_test.cc:65-134): the reference test's own known answers,
    on the device inserter -- a 5x5 grid of 1 m cells."""
    from test_oracle_reference_pins import (INSERTER_2D_ORIGIN, INSERTER_2D_RETURNS,
                                            check_inserter_2d_fixture)
    dev = g2.ProbabilityGridOnDevice(1.0, (1.0, 5.0), 5, 5)
    host = synth.ProbabilityGrid(1.0, (1.0, 5.0), 5, 5)
    dev.insert(INSERTER_2D_ORIGIN, INSERTER_2D_RETURNS, None, 0.7, 0.4, True)
    host.insert(INSERTER_2D_ORIGIN, INSERTER_2D_RETURNS, None, 0.7, 0.4, True)
    view = _DeviceGridView(dev, oracle)
    check_inserter_2d_fixture(view)
    _assert_same(dev, host)
    for _ in range(1000):
        dev.insert(INSERTER_2D_ORIGIN, INSERTER_2D_RETURNS, None, 0.7, 0.4, True)
        host.insert(INSERTER_2D_ORIGIN, INSERTER_2D_RETURNS, None, 0.7, 0.4, True)
    assert abs(view.get_probability(4, 4) - 0.9) < 1e-3       # the hit at (-3.5, 0.5)
    assert abs(view.get_probability(4, 3) - 0.1) < 1e-3       # the miss at (-2.5, 0.5)
    _assert_same(dev, host)


@pytest.mark.parametrize("seed,depth,frd", [(21, 6, 3), (22, 5, 2)])
def test_fast3d_synthetic_world_found(oracle, synth, seed, depth, frd):
    """tests/test_gpu_3d.py::test_fast3d_synthetic_world with a threshold below the best score
    (0.25-0.29): the match is FOUND through the selective yaw filter, with non-identity node /
    submap poses and a tilted gravity alignment, so score, rotational / low-resolution scores and
    the composed pose are all compared."""
    from cartographer_amd import scan_matching_3d as sm3
    from test_oracle_reference_pins_3d import quat_from_angle_axis
    grid, world = synth.make_submap_3d(seed, 0.1, (9.0, 8.0, 4.0), 5, 10, 128)
    low, _ = synth.make_submap_3d(seed, 0.45, (9.0, 8.0, 4.0), 5, 10, 128)
    vox, low_vox = grid.voxels(), low.voxels()
    rng = np.random.default_rng(seed)
    hist = rng.uniform(0.0, 1.0, 120).astype(np.float32)
    hist[10:14] += 6.0
    scan_hist = np.roll(hist, -19).copy()
    pos = world.free_position(seed + 3, 0.6)
    yaw = 0.4
    hi = world.scan(pos, yaw, 8, 96, seed=1)
    lo = hi[::7].copy()
    opt = dict(branch_and_bound_depth=depth, full_resolution_depth=frd, min_rotational_score=0.9,
               min_low_resolution_score=0.3, linear_xy_search_window=1.5,
               linear_z_search_window=0.5, angular_search_window=math.radians(20.0))
    om = oracle.FastCorrelativeScanMatcher3D(0.1, vox, 0.45, low_vox, hist, depth, frd, 0.9, 0.3,
                                             1.5, 0.5, math.radians(20.0))
    gm = sm3.FastCorrelativeScanMatcher3D(0.1, vox, grid.grid_size, 0.45, low_vox, hist, **opt)
    submap_pose = [0.3, -0.2, 0.1] + quat_from_angle_axis(0.2, [0, 0, 1])
    c, s = math.cos(0.2), math.sin(0.2)
    local = np.array([pos[0] + 0.35, pos[1] - 0.25, pos[2] + 0.1])
    node_t = [submap_pose[0] + c * local[0] - s * local[1],
              submap_pose[1] + s * local[0] + c * local[1], submap_pose[2] + local[2]]
    node_pose = node_t + quat_from_angle_axis(0.2 + yaw + 0.1, [0, 0, 1])
    gravity = quat_from_angle_axis(0.01, [1, 0, 0])
    ref = om.match(node_pose, submap_pose, gravity, hi, lo, scan_hist, 0.15)
    got = gm.match(sm3.Rigid3d(tuple(node_pose[:3]), tuple(node_pose[3:])),
                   sm3.Rigid3d(tuple(submap_pose[:3]), tuple(submap_pose[3:])),
                   sm3.TrajectoryNodeData(hi, lo, scan_hist, tuple(gravity)), 0.15)
    assert ref["found"] and got is not None
    for key in ("score", "rotational_score", "low_resolution_score"):
        assert np.float32(got[key]) == np.float32(ref[key]), key
    p = got["pose_estimate"]
    np.testing.assert_array_equal(list(p.translation) + list(p.rotation), ref["pose"])
    assert gm.last_stats["num_scans"] == ref["num_scans"]


def test_constraint_lists_3d_match_restatement(oracle, synth):
    """ConstraintBuilder3D mirror (cartographer_amd/constraint_builder.py) against the restatement
    of constraint_builder_3d.cc over the oracle: two submaps, three nodes, windowed and
    full-submap pairs, a pair beyond max_constraint_distance, a sampler that passes every other
    call, a trimmed submap -- identical constraint lists (ids, order, scores, transforms)."""
    from cartographer_amd import constraint_builder as cb, scan_matching_3d as sm3
    from oracle import constraint_builder_ref as ref
    from test_oracle_reference_pins_3d import quat_from_angle_axis
    opt = dict(depth=5, frd=2, min_rot=0.0, min_low=0.2, lin_xy=1.0, lin_z=0.4,
               ang=math.radians(10.0))
    builder = cb.ConstraintBuilder3D(cb.ConstraintBuilderOptions3D(
        sampling_ratio=0.5, max_constraint_distance=6.0, min_score=0.12,
        global_localization_min_score=0.12, branch_and_bound_depth=opt["depth"],
        full_resolution_depth=opt["frd"], min_rotational_score=opt["min_rot"],
        min_low_resolution_score=opt["min_low"], linear_xy_search_window=opt["lin_xy"],
        linear_z_search_window=opt["lin_z"], angular_search_window=opt["ang"]))
    restated = ref.ConstraintBuilder3DRef(0.5, 6.0, 0.12, 0.12, opt["depth"], opt["frd"],
                                          opt["min_rot"], opt["min_low"], opt["lin_xy"],
                                          opt["lin_z"], opt["ang"])
    hist = np.zeros(16, np.float32)                 # zero histograms: every yaw passes
    submaps, worlds = {}, {}
    for k in range(2):
        grid, world = synth.make_submap_3d(70 + k, 0.2, (8.0, 8.0, 3.0), 4, 8, 96)
        vox = grid.voxels()
        submaps[(0, k)] = (cb.Submap3D(0.2, vox, grid.grid_size, 0.2, vox, hist),
                           (0.2, vox, 0.2, vox, hist))
        worlds[(0, k)] = world
    for node in range(3):
        world = worlds[(0, node % 2)]
        pos = world.free_position(200 + node, 0.6)
        hi = world.scan(pos, 0.1 * node, 6, 64, seed=node)
        lo = hi[::5].copy()
        gravity = quat_from_angle_axis(0.01, [1, 0, 0])
        data = sm3.TrajectoryNodeData(hi, lo, hist, tuple(gravity))
        data_ref = (gravity, hi, lo, hist)
        node_pose = list(pos + np.array([0.3, -0.2, 0.1])) + \
            quat_from_angle_axis(0.1 * node + 0.05, [0, 0, 1])
        for sid, (submap, submap_ref) in submaps.items():
            submap_pose = [0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0]
            if sid[1] != node % 2 and node == 2:
                submap_pose = [40.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0]      # too far: filtered out
            for _ in range(2):                                           # sampler: every other call
                builder.maybe_add_constraint(
                    sid, submap, (0, node), data,
                    sm3.Rigid3d(tuple(node_pose[:3]), tuple(node_pose[3:])),
                    sm3.Rigid3d(tuple(submap_pose[:3]), tuple(submap_pose[3:])))
                restated.maybe_add_constraint(sid, submap_ref, (0, node), data_ref, node_pose,
                                              submap_pose)
            if sid[1] == node % 2:
                builder.maybe_add_global_constraint(sid, submap, (0, node), data, node_pose[3:],
                                                    [1, 0, 0, 0])
                restated.maybe_add_global_constraint(sid, submap_ref, (0, node), data_ref,
                                                     node_pose[3:], [1, 0, 0, 0])
        builder.notify_end_of_node()
        restated.notify_end_of_node()
        if node == 1:                                                     # trimmed submap
            builder.delete_scan_matcher((0, 1))
            restated.delete_scan_matcher((0, 1))
            assert builder.num_scan_matchers() == 1
    got = []
    builder.when_done(got.extend)
    want = restated.when_done()
    assert builder.get_num_finished_nodes() == restated.finished == 3
    assert len(want) >= 3                                                 # not vacuous
    assert [(c.submap_id, c.node_id) for c in got] == [(c["submap_id"], c["node_id"]) for c in want]
    for c, w in zip(got, want):
        for key in ("score", "rotational_score", "low_resolution_score"):
            assert np.float32(getattr(c, key)) == np.float32(w[key]), key
        np.testing.assert_array_equal(list(c.zbar_ij.translation) + list(c.zbar_ij.rotation),
                                      w["zbar_ij"])
        assert c.tag == "INTER_SUBMAP"


def test_device_crop_equals_compute_cropped_grid(g2, synth, oracle):
    """cmx_grid2d_crop = ProbabilityGrid::ComputeCroppedGrid (probability_grid.cc:90-106; the host
    restatement `cropped()` is pinned against the reference's own in
    tests/test_reference_ref_grid.py): limits and cells after scans that made the grid grow, an
    all-unknown grid (-> 1 x 1), a second crop (idempotent), and the loop-closure matcher built
    from the cropped device grid against the oracle on the same cells."""
    from cartographer_amd import scan_matching as sm
    _, lim, world = synth.make_submap(17, 200, 200, 0.05, 2, 100, 30.0, 0.01)
    start = (lim["max_x"] - 4.0, lim["max_y"] - 4.0)
    host = synth.ProbabilityGrid(0.05, start, 16, 16)
    dev = g2.ProbabilityGridOnDevice(0.05, start, 16, 16)
    scan = None
    for k in range(6):
        pose = world.free_pose(1700 + k, 0.4)
        sensor = world.scan(pose, 257, 30.0, 0.01, k)
        c, s = math.cos(pose[2]), math.sin(pose[2])
        cloud = np.zeros_like(sensor)
        cloud[:, 0] = (pose[0] + c * sensor[:, 0].astype(np.float64)
                       - s * sensor[:, 1].astype(np.float64)).astype(np.float32)
        cloud[:, 1] = (pose[1] + s * sensor[:, 0].astype(np.float64)
                       + c * sensor[:, 1].astype(np.float64)).astype(np.float32)
        host.insert(pose[:2], cloud)
        dev.insert(pose[:2], cloud)
        scan = sensor
    before = dev.limits
    cropped = host.cropped()
    dev.crop()
    _assert_same(dev, cropped)
    assert dev.limits["num_x_cells"] < before["num_x_cells"]          # it did shrink
    dev.crop()                                                         # idempotent
    _assert_same(dev, cropped)
    # the finished submap's matcher, built from the cropped grid in HBM
    cells, l2 = dev.cells, dev.limits
    gm = dev.fast_matcher(5)
    om = oracle.FastCorrelativeScanMatcher2D(cells, l2["resolution"], l2["max_x"], l2["max_y"], 5)
    ref = om.match_full_submap(scan, 0.3)
    found, score, pose = gm.match_full_submap(scan, 0.3)
    assert bool(found) == ref["found"]
    if found:
        assert np.float32(score) == np.float32(ref["score"])
        np.testing.assert_allclose([pose.x, pose.y, pose.theta], ref["pose"], rtol=0, atol=1e-12)
    # all unknown -> CellLimits(1, 1), max unchanged (grid_2d.cc:106-110)
    empty = g2.ProbabilityGridOnDevice(0.1, (3.0, -2.0), 30, 20)
    empty.crop()
    assert empty.limits == dict(resolution=0.1, max_x=3.0, max_y=-2.0, num_x_cells=1,
                                num_y_cells=1)
    assert empty.cells.tolist() == [[0]]


# ---------------------------------------------------------------------------- part 3: 3D grid
def _same_voxel_grid(dev, host):
    assert dev.grid_size == host.grid_size
    np.testing.assert_array_equal(dev.voxels(), host.voxels())


def test_device_range_data_inserter_3d_reference_fixture(synth):
    """RangeDataInserter3DTest.InsertPointCloud / ProbabilityProgression
    (mapping/3d/range_data_inserter_3d_test.cc:28-53, :97-114, :138-157) on the device inserter:
    the reference test's known answers, and the host builder (pinned on the reference's own
    inserter) voxel for voxel."""
    from cartographer_amd import grid_3d
    from test_oracle_reference_pins_3d import INSERTER_3D_ORIGIN, INSERTER_3D_RETURNS
    dev, host = grid_3d.HybridGridOnDevice(1.0), synth.HybridGrid(1.0)
    dev.insert(INSERTER_3D_ORIGIN, INSERTER_3D_RETURNS, 0.7, 0.4, 1000)
    host.insert(INSERTER_3D_ORIGIN, INSERTER_3D_RETURNS, 0.7, 0.4, 1000)
    _same_voxel_grid(dev, host)
    v = dev.voxels()
    at = {(int(r["x"]), int(r["y"]), int(r["z"])): int(r["value"]) for r in v}
    assert (0, 0, -4) in at and (0, 0, -3) in at and (0, 0, -2) in at      # misses along a ray
    for x in range(-4, 5):
        for y in range(-4, 5):
            known = (x, y, 4) in at
            assert known == (not (x < -3 or x > 0 or y != x + 2))           # the four hits
    assert at[(-2, 0, 4)] > at[(-2, 0, 3)]                                  # hit 0.7 > miss 0.4
    for _ in range(1000):
        dev.insert(INSERTER_3D_ORIGIN, INSERTER_3D_RETURNS, 0.7, 0.4, 1000)
        host.insert(INSERTER_3D_ORIGIN, INSERTER_3D_RETURNS, 0.7, 0.4, 1000)
    _same_voxel_grid(dev, host)
    assert abs(host.get_probability((-2, 0, 4)) - 0.9) < 1e-3               # saturated


@pytest.mark.parametrize("seed,free", [(2, 2), (7, 0), (9, 40)])
def test_device_range_data_inserter_3d_scans(synth, seed, free):
    """Eight scans of a synthetic room, as synth.make_submap_3d inserts them: re-updates through
    the odds tables, hits before misses, the last `free` voxels of every ray, DynamicGrid growth
    from 128 to 256 voxels, the brick re-allocated as the scene widens."""
    from cartographer_amd import grid_3d
    world = synth.World3D(seed, (15.0, 15.0, 7.5))
    dev, host = grid_3d.HybridGridOnDevice(0.1), synth.HybridGrid(0.1)
    for p in range(8):
        pos = world.free_position(seed * 1009 + p, 0.5)
        yaw = 0.37 * p
        sensor = world.scan(pos, yaw, 8, 96, seed=seed * 31 + p).astype(np.float64)
        c, s = math.cos(yaw), math.sin(yaw)
        in_map = np.stack([pos[0] + c * sensor[:, 0] - s * sensor[:, 1],
                           pos[1] + s * sensor[:, 0] + c * sensor[:, 1],
                           pos[2] + sensor[:, 2]], 1).astype(np.float32)
        hit, miss = (0.55, 0.49) if p % 3 == 2 else (0.7, 0.4)
        dev.insert(pos.astype(np.float32), in_map, hit, miss, free)
        host.insert(pos.astype(np.float32), in_map, hit, miss, free)
        _same_voxel_grid(dev, host)
    assert dev.grid_size == 256
    # nothing to insert: a no-op on both
    dev.insert(pos.astype(np.float32), np.zeros((0, 3), np.float32), 0.7, 0.4, free)
    _same_voxel_grid(dev, host)


def test_rt3d_on_the_device_built_grid(synth, oracle):
    """Insert -> Match: the real-time 3D matcher on the voxels of the grid built in HBM equals the
    oracle on the host-built grid (LocalTrajectoryBuilder3D's per-scan loop)."""
    from cartographer_amd import grid_3d, scan_matching_3d as sm3
    from test_oracle_reference_pins_3d import quat_from_angle_axis
    world = synth.World3D(5, (8.0, 8.0, 4.0))
    dev, host = grid_3d.HybridGridOnDevice(0.1), synth.HybridGrid(0.1)
    for p in range(4):
        pos = world.free_position(50 + p, 0.5)
        sensor = world.scan(pos, 0.2 * p, 6, 64, seed=p).astype(np.float64)
        c, s = math.cos(0.2 * p), math.sin(0.2 * p)
        in_map = np.stack([pos[0] + c * sensor[:, 0] - s * sensor[:, 1],
                           pos[1] + s * sensor[:, 0] + c * sensor[:, 1],
                           pos[2] + sensor[:, 2]], 1).astype(np.float32)
        dev.insert(pos.astype(np.float32), in_map, 0.7, 0.4, 2)
        host.insert(pos.astype(np.float32), in_map, 0.7, 0.4, 2)
    cloud = world.scan(pos, 0.6, 6, 64, seed=9)
    init = list(pos + np.array([0.07, -0.04, 0.02])) + quat_from_angle_axis(0.61, [0, 0, 1])
    ref = oracle.rt3d_match(0.1, host.voxels(), init, cloud, 0.2, math.radians(1.0), 0.1, 0.1)
    m = sm3.RealTimeCorrelativeScanMatcher3D(0.2, math.radians(1.0), 0.1, 0.1)
    score, pose = m.match(sm3.Rigid3d(tuple(init[:3]), tuple(init[3:])), cloud, 0.1, dev.voxels())
    assert np.float32(score) == np.float32(ref["score"])
    np.testing.assert_array_equal(list(pose.translation) + list(pose.rotation), ref["pose"])
    # ... and on the brick itself, where it lies in HBM (cmx_rt3d_match_grid: no voxel list)
    score, pose = m.match_grid(sm3.Rigid3d(tuple(init[:3]), tuple(init[3:])), cloud, dev)
    assert np.float32(score) == np.float32(ref["score"])
    np.testing.assert_array_equal(list(pose.translation) + list(pose.rotation), ref["pose"])
    assert 1 <= m.last_stats["nodes_expanded"] <= 4096          # the bounds path ran
    # a grid nothing was inserted into: every candidate scores kMinProbability, the first wins
    empty = grid_3d.HybridGridOnDevice(0.1)
    none = oracle.rt3d_match(0.1, host.voxels()[:0], init, cloud[:200], 0.1, math.radians(1.0),
                             0.0, 0.0)
    small = sm3.RealTimeCorrelativeScanMatcher3D(0.1, math.radians(1.0), 0.0, 0.0)
    score, pose = small.match_grid(sm3.Rigid3d(tuple(init[:3]), tuple(init[3:])), cloud[:200], empty)
    assert np.float32(score) == np.float32(none["score"])
    np.testing.assert_array_equal(list(pose.translation) + list(pose.rotation), none["pose"])


def test_ceres3d_on_the_resident_grids(synth, oracle):
    """Insert -> refine: CeresScanMatcher3D against the HybridGrids cmx_grid3d keeps in HBM
    (cmx_ceres3d_match_grids, LocalTrajectoryBuilder3D::ScanMatch's call) equals the voxel-list
    entry point bit for bit and the oracle on the host-built grids to 1e-6; a grid nothing was
    inserted into reads kMinProbability everywhere (zero gradient: the pose stays)."""
    from cartographer_amd import grid_3d, scan_matching_3d as sm3
    from test_oracle_reference_pins_3d import quat_from_angle_axis
    world = synth.World3D(5, (8.0, 8.0, 4.0))
    dev = [grid_3d.HybridGridOnDevice(0.1), grid_3d.HybridGridOnDevice(0.3)]
    host = [synth.HybridGrid(0.1), synth.HybridGrid(0.3)]
    for p in range(4):
        pos = world.free_position(50 + p, 0.5)
        sensor = world.scan(pos, 0.2 * p, 6, 64, seed=p).astype(np.float64)
        c, s = math.cos(0.2 * p), math.sin(0.2 * p)
        in_map = np.stack([pos[0] + c * sensor[:, 0] - s * sensor[:, 1],
                           pos[1] + s * sensor[:, 0] + c * sensor[:, 1],
                           pos[2] + sensor[:, 2]], 1).astype(np.float32)
        for g in dev + host:
            g.insert(pos.astype(np.float32), in_map, 0.7, 0.4, 2)
    cloud = world.scan(pos, 0.6, 6, 64, seed=9)
    hi, lo = cloud[::2].copy(), cloud[::5].copy()
    init_t = pos + np.array([0.03, -0.02, 0.01])
    init = list(init_t) + quat_from_angle_axis(0.61, [0.02, -0.01, 1.0])
    first = sm3.Rigid3d(tuple(init[:3]), tuple(init[3:]))
    m = sm3.CeresScanMatcher3D([1.0, 6.0], 5.0, 4e2, max_num_iterations=12)
    pose, summary = m.match_grids(init_t, first, [(hi, dev[0]), (lo, dev[1])])
    pairs = [(hi, 0.1, host[0].voxels()), (lo, 0.3, host[1].voxels())]
    listed, listed_summary = m.match(init_t, first, pairs)
    assert pose == listed and summary == listed_summary
    ref = oracle.ceres3d_match(pairs, init_t, init, [1.0, 6.0], translation_weight=5.0,
                               rotation_weight=4e2, max_num_iterations=12)
    np.testing.assert_allclose(list(pose.translation) + list(pose.rotation), ref["pose"],
                               rtol=0, atol=1e-6)
    assert summary["num_successful_steps"] == ref["num_successful_steps"] >= 1
    empty = grid_3d.HybridGridOnDevice(0.1)
    one = sm3.CeresScanMatcher3D([1.0], 5.0, 4e2, max_num_iterations=12)
    stay, stay_summary = one.match_grids(init_t, first, [(hi, empty)])
    np.testing.assert_allclose(list(stay.translation) + list(stay.rotation), init, rtol=0,
                               atol=1e-12)
    assert stay_summary["initial_cost"] == pytest.approx(0.5 * 0.9 ** 2, rel=1e-7)   # (0.1f)


def test_fast3d_batch_equals_individual(synth):
    """cmx_fast3d_match_batch (pairs searched concurrently from host threads on separate streams)
    returns, pair by pair, exactly what the single calls return -- windowed and full-submap pairs
    mixed, per-pair thresholds, one pair that finds nothing; also twice in a row (the leased
    workspaces are reused)."""
    from cartographer_amd import scan_matching_3d as sm3
    from test_oracle_reference_pins_3d import quat_from_angle_axis
    hist = np.zeros(16, np.float32)
    opt = dict(branch_and_bound_depth=5, full_resolution_depth=2, min_rotational_score=0.0,
               min_low_resolution_score=0.2, linear_xy_search_window=1.0,
               linear_z_search_window=0.4, angular_search_window=math.radians(10.0))
    matchers, worlds = [], []
    for k in range(3):
        grid, world = synth.make_submap_3d(70 + k, 0.2, (8.0, 8.0, 3.0), 4, 8, 96)
        vox = grid.voxels()
        matchers.append(sm3.FastCorrelativeScanMatcher3D(0.2, vox, grid.grid_size, 0.2, vox, hist,
                                                         **opt))
        worlds.append(world)
    pos = worlds[0].free_position(200, 0.6)
    hi = worlds[0].scan(pos, 0.0, 6, 64, seed=0)
    data = sm3.TrajectoryNodeData(hi, hi[::5].copy(), hist,
                                  tuple(quat_from_angle_axis(0.01, [1, 0, 0])))
    node = sm3.Rigid3d(tuple(pos + np.array([0.3, -0.2, 0.1])),
                       tuple(quat_from_angle_axis(0.05, [0, 0, 1])))
    ident = sm3.Rigid3d()
    pairs = [(0, False, 0.12), (1, False, 0.12), (2, False, 0.99), (0, True, 0.12),
             (1, True, 0.3), (2, False, 0.12), (0, False, 0.4)]
    expected = []
    for k, full, threshold in pairs:
        if full:
            expected.append(matchers[k].match_full_submap(node.rotation, ident.rotation, data,
                                                          threshold))
        else:
            expected.append(matchers[k].match(node, ident, data, threshold))
    assert any(e is not None for e in expected) and any(e is None for e in expected)
    for _ in range(2):
        got, stats = sm3.fast3d_match_batch([matchers[k] for k, _, _ in pairs],
                                            [node] * len(pairs), [ident] * len(pairs),
                                            [full for _, full, _ in pairs],
                                            [t for _, _, t in pairs], data)
        assert stats["candidates_scored"] > 0
        for e, g in zip(expected, got):
            assert (e is None) == (g is None)
            if e is None:
                continue
            for key in ("score", "rotational_score", "low_resolution_score"):
                assert np.float32(e[key]) == np.float32(g[key]), key
            assert e["pose_estimate"] == g["pose_estimate"]


# ---- BASELINE config C4 at its own window, against the reference's own source -----------------
@pytest.fixture(scope="module")
def golden_c4():
    with open(os.path.join(GOLDEN, "rt3d_c4_reference.json")) as f:
        return json.load(f)


def _rt3d_against_golden(synth, d, g, debug, bulk):
    from cartographer_amd import scan_matching_3d as sm3
    # group bounds checked against member bounds; staged second round: every candidate still
    # evaluated in full, every intermediate bound checked against its final sum, the drops
    # applied afterwards
    debug(rt3d_verify=0 if bulk in ("shipped", "unstaged") else 1)
    # "tiles": LDS-tiled bulk passes (cross-checked against the gather kernels element by
    # element); "1": gather kernels; "0": exhaustive
    # "fixed": the tiled passes as shipped (fixed-point group centres), "fixed-all": the same
    # with every group expanded so that VERIFY compares every group bound with all its members
    # "shipped": exactly what a caller gets (no verification mode: the staged second round
    # really drops candidates from its work lists); "unstaged": the same without the stages
    # Round 4: "fixed" / "fixed-all" / "shipped" / "unstaged" run the rotation-block level above
    # the group pass (verification: every computed (rotation, group) bound against its block's;
    # "fixed-all" computes and checks every pair); "dense": the group pass over all pairs, verified;
    # "dense-shipped": that without verification
    debug(rt3d_verify=0 if bulk in ("shipped", "unstaged", "dense-shipped") else 1)
    debug(rt3d_legacy=1 if bulk == "0" else 0,
          rt3d_no_tiles=0 if bulk in ("tiles", "fixed", "fixed-all", "shipped", "unstaged", "dense",
                                      "dense-shipped") else 1,
          rt3d_unstaged=1 if bulk == "unstaged" else 0,
          rt3d_crosscheck=1 if bulk == "tiles" else 0,
          rt3d_expand_all=1 if bulk == "fixed-all" else 0,
          rt3d_no_rotblocks=1 if bulk in ("dense", "dense-shipped") else 0)
    m = sm3.RealTimeCorrelativeScanMatcher3D(d["lin"], d["ang"], d["tw"], d["rw"])
    score, pose = m.match(sm3.Rigid3d(tuple(d["init"][:3]), tuple(d["init"][3:])), d["cloud"],
                          d["res"], d["vox"])
    assert m.last_stats["candidates_scored"] == g["num_candidates"]
    assert np.float32(score) == np.float32(g["score"]), (score, g["score"])
    np.testing.assert_array_equal(list(pose.translation) + list(pose.rotation), g["pose"])
    return m.last_stats


@pytest.mark.parametrize("bulk", ["tiles", "fixed", "fixed-all", "shipped", "unstaged", "dense",
                                  "dense-shipped", "1", "0"])
def test_rt3d_c4_shaped_equals_the_reference(synth, golden_c4, debug, bulk):
    """C4's shape at 4096 points: L = 5 -> 6^3 = 216 groups of 2x2x2 translations per rotation
    (the flat 192-lane group mapping of rt_3d.hip spans rotations), A = 3 -> 343 rotations, a
    tilted initial orientation.  Bounds path and exhaustive path, f32 score bit-equal and pose
    equal to what RealTimeCorrelativeScanMatcher3D::Match itself returned
    (tests/golden/rt3d_c4_reference.json)."""
    import workloads as w
    st = _rt3d_against_golden(synth, w.rt3d_c4_shaped(synth), golden_c4["rt3d_c4_shaped"],
                              debug, bulk)
    if bulk not in ("0", "fixed-all"):
        assert st["coarse_candidates"] < st["candidates_scored"]      # bounds did exclude


@pytest.mark.parametrize("bulk", ["tiles", "fixed", "shipped", "unstaged", "dense", "1", "0"])
def test_rt3d_c4_at_its_baseline_window_equals_the_reference(synth, golden_c4, debug, bulk):
    """BASELINE config[3] exactly as bench.py times it (65 536 points, 150^3 grid, +-0.5 m /
    +-2 deg: 1 771 561 candidates) against the reference's own
    real_time_correlative_scan_matcher_3d.cc run over that search space (ten minutes on 8 host
    cores, tests/golden/make_rt3d_c4_golden.py): score bit-equal, pose equal, with the bounds
    (debug switch rt3d_verify) and with every candidate scored exhaustively."""
    import workloads as w
    _rt3d_against_golden(synth, w.rt3d_c4(synth), golden_c4["rt3d_c4"], debug, bulk)


def test_intensity_grid_on_the_device_equals_the_reference_and_feeds_the_refinement(synth, oracle):
    """f3, intensities: cmx_grid3d_insert_with_intensities fills a resident IntensityHybridGrid
    cell for cell like InsertIntensitiesIntoGrid (range_data_inserter_3d.cc:54-70) -- counts and
    f32 sums bit-identical to the restatement (which tests/test_reference_ref_3d.py pins on the
    reference's own inserter), across brick growth, returns above the threshold, a NaN, shared
    voxels, a scan without intensities -- and cmx_ceres3d_match_grids_intensity on the two resident
    grids returns what the voxel-list entry point returns, bit for bit."""
    from cartographer_amd import grid_3d, scan_matching_3d as sm3
    from test_oracle_reference_pins_3d import quat_from_angle_axis
    rng = np.random.default_rng(11)
    world = synth.World3D(5, (8.0, 8.0, 4.0))
    res = 0.1
    dev, idev = grid_3d.HybridGridOnDevice(res), grid_3d.IntensityHybridGridOnDevice(res)
    host = synth.HybridGrid(res)
    vox = np.zeros(0, oracle.INTENSITY_VOXEL_DTYPE)
    assert len(idev.voxels()) == 0
    for p in range(5):
        pos = world.free_position(50 + p, 0.5)
        sensor = world.scan(pos, 0.2 * p, 8, 96, seed=p).astype(np.float64)
        c, s = math.cos(0.2 * p), math.sin(0.2 * p)
        in_map = np.stack([pos[0] + c * sensor[:, 0] - s * sensor[:, 1],
                           pos[1] + s * sensor[:, 0] + c * sensor[:, 1],
                           pos[2] + sensor[:, 2]], 1).astype(np.float32)
        if p == 1:
            in_map[:40] = in_map[40:80] + 0.003          # several returns per voxel, in order
        ints = rng.uniform(0.0, 60.0, len(in_map)).astype(np.float32)
        use = None if p == 3 else ints
        dev.insert_with_intensities(idev, pos.astype(np.float32), in_map, use, 0.7, 0.4, 2, 40.0)
        host.insert(pos.astype(np.float32), in_map, 0.7, 0.4, 2)
        vox = oracle.insert_intensities(res, vox, in_map, use, 40.0)
        got = idev.voxels()
        assert got.tobytes() == vox.tobytes(), p
    np.testing.assert_array_equal(dev.voxels(), host.voxels())
    assert vox["count"].max() >= 2
    # refinement against the two resident grids = against their voxel lists
    cloud = world.scan(pos, 0.6, 6, 64, seed=9)
    cints = rng.uniform(0.0, 60.0, len(cloud)).astype(np.float32)
    init_t = pos + np.array([0.03, -0.02, 0.01])
    init = list(init_t) + quat_from_angle_axis(0.61, [0.02, -0.01, 1.0])
    first = sm3.Rigid3d(tuple(init[:3]), tuple(init[3:]))
    m = sm3.CeresScanMatcher3D([1.0], 5.0, 4e2, max_num_iterations=12)
    opts = (0.5, 0.3, 45.0)
    pose, summary = m.match_grids(init_t, first, [(cloud, dev, cints, idev, opts)])
    listed, listed_summary = m.match(init_t, first, [(cloud, res, host.voxels(), cints, vox, opts)])
    assert pose == listed and summary == listed_summary
    plain, _ = m.match_grids(init_t, first, [(cloud, dev)])
    assert plain != pose                                   # the intensity block is there
    ref = oracle.ceres3d_match_intensity([(cloud, res, host.voxels(), cints, vox, opts)], init_t,
                                         init, [1.0], translation_weight=5.0, rotation_weight=4e2,
                                         max_num_iterations=12)
    np.testing.assert_allclose(list(pose.translation) + list(pose.rotation), ref["pose"],
                               rtol=0, atol=1e-6)
    # a NaN intensity is not above the threshold (`>`): inserted, as in the reference
    nan_dev, nan_idev = grid_3d.HybridGridOnDevice(res), grid_3d.IntensityHybridGridOnDevice(res)
    ints = rng.uniform(0.0, 30.0, len(in_map)).astype(np.float32)
    ints[5] = np.nan
    nan_dev.insert_with_intensities(nan_idev, pos.astype(np.float32), in_map, ints, 0.7, 0.4, 2, 40.0)
    got = nan_idev.voxels()
    want = oracle.insert_intensities(res, np.zeros(0, oracle.INTENSITY_VOXEL_DTYPE), in_map, ints, 40.0)
    for field in ("x", "y", "z", "count"):
        np.testing.assert_array_equal(got[field], want[field])
    np.testing.assert_array_equal(got["sum"], want["sum"])        # (NaN == NaN here)
    assert np.isnan(got["sum"]).sum() == 1
    # an intensity grid nothing was inserted into interpolates 0 everywhere
    empty = grid_3d.IntensityHybridGridOnDevice(res)
    a, _ = m.match_grids(init_t, first, [(cloud, dev, cints, empty, opts)])
    b, _ = m.match(init_t, first, [(cloud, res, host.voxels(), cints,
                                    np.zeros(0, oracle.INTENSITY_VOXEL_DTYPE), opts)])
    assert a == b


def test_intensity_runs_of_every_length_keep_the_point_order(oracle):
    """Many returns in ONE voxel (dense near-range returns, coarse grids): the run's head adds them
    in point order, fetching eight ahead.  Runs of 1..20, 63, 64, 65 and 500 returns, interleaved
    with one another and with skipped returns, against the restatement's sequential loop: counts and
    f32 sums bit-identical -- sums that depend on the order (values over eight decades)."""
    from cartographer_amd import grid_3d
    rng = np.random.default_rng(5)
    res = 0.4
    lengths = list(range(1, 21)) + [63, 64, 65, 500]
    points, ints = [], []
    for k, length in enumerate(lengths):
        centre = res * np.array([k % 6, k // 6, 1.0])     # (cells are ROUNDED coordinates)
        points.append(centre + rng.uniform(-0.15, 0.15, (length, 3)))
        ints.append(10.0 ** rng.uniform(-4.0, 1.5, length))
    points = np.concatenate(points).astype(np.float32)
    ints = np.concatenate(ints).astype(np.float32)
    order = rng.permutation(len(points))                  # runs interleaved in the scan
    points, ints = points[order], ints[order]
    ints[::37] = 50.0                                     # above the threshold: skipped
    origin = np.array([0.0, 0.0, 5.0], np.float32)
    dev, idev = grid_3d.HybridGridOnDevice(res), grid_3d.IntensityHybridGridOnDevice(res)
    vox = np.zeros(0, oracle.INTENSITY_VOXEL_DTYPE)
    for _ in range(2):                                    # the second scan adds to existing sums
        dev.insert_with_intensities(idev, origin, points, ints, 0.7, 0.4, 2, 40.0)
        vox = oracle.insert_intensities(res, vox, points, ints, 40.0)
        assert idev.voxels().tobytes() == vox.tobytes()
    assert len(vox) == len(lengths) and vox["count"].max() >= 900


def test_bench_force_dist_on_one_gpu():
    """bench.py's distributed branch on the one GPU of the box: `--force-dist --gpus 1` initialises
    torch.distributed over RCCL (backend "nccl") with one rank and runs the collectives of a
    sharded step -- the all-gather of the per-submap results and the all-reduce(max) of the best
    key -- on the device; rank 0 prints the ONE compact line, parity gate included."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT="29561")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--force-dist", "--gpus",
                          "1", "--steps", "2", "--warmup", "1", "--submaps", "2", "--no-other",
                          "--no-cpu-baseline", "--passes-per-step", "2", "--details", ""],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 4096
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["config"]["name"] == "c3"
    assert line["config"]["constraints_found_node_wide"] >= 1
    # (every one of the 8 scans the passes cycle through against both submaps)
    assert line["parity"]["bit_exact"] and line["parity"]["checked"] == 16
