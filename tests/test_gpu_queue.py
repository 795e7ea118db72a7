"""The work-queue branch and bound of round 6 (fast_2d.hip, TreeQueueKernel) against the oracle and
against the level-synchronous launches it replaces (debug switch fast2d_queue = 2), on searches
that expand from 1 600 to 60 000 nodes, in every shape the queue can take: one workgroup, a grid
that does not divide the list, sub-queues so small that they overflow (the fall-back paths), the
queue for a batch, eight host threads at once.  Bars as in test_gpu_2d.py: found flag and f32 score
bit-equal, pose to 1e-12 (f64 from integer offsets).
(SM2 = /root/reference/cartographer/mapping/internal/2d/scan_matching; the search restated is
SM2/fast_correlative_scan_matcher_2d.cc:264-378.)
"""
import math
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sm():
    from cartographer_amd import _lib, scan_matching
    assert _lib.lib().cmx_device_count() >= 1, "no HIP device: these tests need the GPU"
    return scan_matching


@pytest.fixture(scope="module")
def world8(synth, oracle):
    """bench.py's C2 world (seed 42, 400 x 400, depth 7) with the eight scans its headline cycles
    through, and the oracle's full-submap result for each (0.2 - 0.7 s of one core each)."""
    cells, lim, world = synth.make_submap(42, 400, 400, 0.05, 30, 1000, 30.0, 0.01)
    scans = [world.scan(world.free_pose(1234 + k, 0.5), 1000, 30.0, 0.01, 7 + k) for k in range(8)]
    om = oracle.FastCorrelativeScanMatcher2D(cells, 0.05, lim["max_x"], lim["max_y"], 7)
    with ThreadPoolExecutor(8) as pool:
        refs = list(pool.map(lambda sc: om.match_full_submap(sc, 0.6), scans))
    return cells, lim, world, scans, refs


def _matcher(sm, cells, lim, depth=7, **kw):
    return sm.FastCorrelativeScanMatcher2D(
        sm.Grid2D(cells, lim["resolution"], lim["max_x"], lim["max_y"]), depth, **kw)


def _xyt(pose):
    return (pose.x, pose.y, pose.theta) if hasattr(pose, "theta") else tuple(np.asarray(pose)[:3])


def _same(got, ref):
    found, score, pose = got
    assert bool(found) == bool(ref["found"])
    if ref["found"]:
        assert np.float32(score) == np.float32(ref["score"])
        assert np.max(np.abs(np.asarray(_xyt(pose)) - np.asarray(ref["pose"][:3]))) < 1e-12


@pytest.mark.parametrize("switches", [
    {},                                                   # as shipped
    {"fast2d_queue": 2},                                  # the level-synchronous launches
    {"fast2d_queue_blocks": 1},                           # four wavefronts do everything
    {"fast2d_queue_blocks": 37},                          # a grid that divides nothing
    {"fast2d_queue_blocks": 2048},                        # more wavefronts than sub-queues
    {"fast2d_queue_lost": 1},
    {"fast2d_queue_lost": 1000},
    {"fast2d_queue_capacity": 2},                         # sub-queues overflow: level path takes over
    {"fast2d_queue_capacity": 2, "frontier_capacity": 4096},   # ... and overflows too: strict, chunked
], ids=lambda s: ",".join(f"{k}={v}" for k, v in s.items()) or "default")
def test_queue_search_equals_the_oracle_on_eight_scans(sm, world8, debug, switches):
    cells, lim, _, scans, refs = world8
    debug(**switches)
    gm = _matcher(sm, cells, lim)
    for scan, ref in zip(scans, refs):
        assert ref["found"]
        _same(gm.match_full_submap(scan, 0.6), ref)


def test_queue_search_counts_its_work(sm, world8, debug):
    """The work counters of the queue path: every scored candidate at every depth.  The count
    depends on when the bound rises (it is not the same from run to run); on the easy scan (#0,
    whose dive finds the optimum when the lowest-resolution scores are exact: fast2d_group = 1)
    both paths expand exactly the 3 076 nodes that reach the final bound, and on the hardest of the
    eight the queue's chains, taken best first, need a third of what the level-synchronous launches
    expand.  Under group bounds (the default at this depth) the dives start from bounds: a few
    thousand nodes on the easy scan, the same order on the hard one."""
    cells, lim, _, scans, _ = world8
    gm = _matcher(sm, cells, lim)
    debug(fast2d_group=1)
    gm.match_full_submap(scans[0], 0.6)
    queue = dict(gm.last_stats)
    debug(fast2d_group=1, fast2d_queue=2)
    gm.match_full_submap(scans[0], 0.6)
    level = dict(gm.last_stats)
    assert queue["coarse_candidates"] == level["coarse_candidates"]
    assert queue["nodes_expanded"] == level["nodes_expanded"] == 3076
    assert queue["candidates_scored"] == level["candidates_scored"]
    debug(fast2d_group=0, fast2d_queue=0)
    gm.match_full_submap(scans[0], 0.6)
    grouped = dict(gm.last_stats)
    assert grouped["coarse_candidates"] == queue["coarse_candidates"]
    assert 3076 <= grouped["nodes_expanded"] < 12000
    gm.match_full_submap(scans[7], 0.6)       # the hardest of the eight
    hard = dict(gm.last_stats)
    assert 20000 < hard["nodes_expanded"] < 45000      # level path: 66 000 - 93 000; perfect seed: 25 000
    assert hard["expansion_lookups"] > 0


# ----------------------------------------------------------------------------
# Group bounds of the fused front end (round 6): three rotations, one sum over the dilated level
# ----------------------------------------------------------------------------
@pytest.mark.parametrize("switches", [
    {"fast2d_group_verify": 1},                           # as shipped + every bound checked on the device
    {"fast2d_group": 1},                                  # every rotation summed on the level itself
    {"fast2d_group_verify": 3},                           # every unit as if its premise had failed
    {"fast2d_group_verify": 1, "fast2d_queue": 2},        # the level-synchronous launches behind it
], ids=lambda s: ",".join(f"{k}={v}" for k, v in s.items()))
def test_group_bounds_equal_the_oracle_on_eight_scans(sm, world8, debug, switches):
    """The lowest-resolution scores of the default path are upper bounds shared by three
    neighbouring rotations (fast_2d.hip, PrepScoreFusedKernel): results bit-equal to the oracle's on
    all eight scans, with the device checking every bound against the exact sums of its rotations
    (a violation fails the call), and with every unit's outer rotations unbounded."""
    cells, lim, _, scans, refs = world8
    debug(**switches)
    gm = _matcher(sm, cells, lim)
    for scan, ref in zip(scans, refs):
        _same(gm.match_full_submap(scan, 0.6), ref)
    assert gm.last_stats["coarse_candidates"] > 300000     # (the search space, bounded or summed)


@pytest.mark.parametrize("depth,group", [(2, 2), (3, 2), (4, 2), (5, 0), (6, 0)])
def test_group_bounds_at_every_depth(sm, oracle, synth, debug, depth, group):
    """Group bounds forced on below their default depth (two cells of dilation against windows of
    2 - 8 cells: valid, if useless) and as shipped above it; windowed and full-submap searches,
    bounds verified on the device."""
    cells, lim, world = synth.make_submap(7, 160, 160, 0.05, 12, 500, 30.0, 0.01)
    scan = world.scan(world.free_pose(5, 0.5), 700, 30.0, 0.01, 2)
    om = oracle.FastCorrelativeScanMatcher2D(cells, 0.05, lim["max_x"], lim["max_y"], depth, 2.0,
                                             math.radians(25.0))
    gm = _matcher(sm, cells, lim, depth, linear_search_window=2.0,
                  angular_search_window=math.radians(25.0))
    debug(fast2d_group=group, fast2d_group_verify=1)
    _same(gm.match_full_submap(scan, 0.5), om.match_full_submap(scan, 0.5))
    truth = world.free_pose(5, 0.5)
    init = [truth[0] + 0.3, truth[1] - 0.2, truth[2] + 0.1]
    _same(gm.match(sm.Rigid2d(*init), scan, 0.4), om.match(init, scan, 0.4))


def test_group_bounds_and_ties(sm, oracle, debug):
    """Leaves that tie for the best score: the replay of the reference's order needs the exact
    lowest-resolution scores, which the front end computes again for that problem (RescoreExact).
    A grid with two identical islands far apart: the two best leaves tie, their lowest-resolution
    ancestors differ."""
    cells = np.zeros((128, 128), np.uint16)
    for (cy, cx) in ((30, 30), (95, 95)):
        cells[cy:cy + 3, cx:cx + 3] = 2000            # (a low correspondence cost: occupied)
    cloud = np.array([[0.0, 0.0, 0.0], [0.05, 0.0, 0.0], [0.0, 0.05, 0.0]], np.float32)
    om = oracle.FastCorrelativeScanMatcher2D(cells, 0.05, 6.4, 6.4, 5, 7.0, math.radians(30.0))
    gm = sm.FastCorrelativeScanMatcher2D(sm.Grid2D(cells, 0.05, 6.4, 6.4), 5,
                                         linear_search_window=7.0,
                                         angular_search_window=math.radians(30.0))
    ref = om.match_full_submap(cloud, 0.1)
    for group in (2, 1):
        debug(fast2d_group=group, fast2d_group_verify=1 if group == 2 else 0)
        _same(gm.match_full_submap(cloud, 0.1), ref)


def test_group_bounds_for_a_batch(sm, world8, synth, debug):
    """A batch (the level-synchronous launches, store_scans) under group bounds, verified, against
    the same batch with exact lowest-resolution scores."""
    cells, lim, _, scans, refs = world8
    grids = [(cells, lim)]
    for seed in (43, 44):
        c, l, _ = synth.make_submap(seed, 400, 400, 0.05, 30, 1000, 30.0, 0.01)
        grids.append((c, l))
    matchers = [_matcher(sm, *grids[g]) for g in (0, 1, 0, 2, 0, 1)]
    debug(fast2d_group_verify=1)
    found, scores, poses, _ = sm.match_full_submap_batch(matchers, scans[3], 0.6)
    debug(fast2d_group=1, fast2d_group_verify=0)
    found2, scores2, poses2, _ = sm.match_full_submap_batch(matchers, scans[3], 0.6)
    np.testing.assert_array_equal(found, found2)
    np.testing.assert_array_equal(scores[found != 0], scores2[found2 != 0])
    np.testing.assert_array_equal(np.asarray(poses)[found != 0], np.asarray(poses2)[found2 != 0])
    assert found[0] and np.float32(scores[0]) == np.float32(refs[3]["score"])


def test_queue_search_windowed_and_not_found(sm, oracle, world8, debug):
    """Match() with a search window (children beyond the bounds are masked), and a threshold
    nothing reaches: found = 0 on both sides."""
    cells, lim, world, scans, _ = world8
    om = oracle.FastCorrelativeScanMatcher2D(cells, 0.05, lim["max_x"], lim["max_y"], 7, 3.0,
                                             math.radians(20.0))
    gm = _matcher(sm, cells, lim, linear_search_window=3.0,
                  angular_search_window=math.radians(20.0))
    for k in (1, 4, 7):
        truth = world.free_pose(1234 + k, 0.5)
        init = [truth[0] + 0.4, truth[1] - 0.3, truth[2] + 0.1]
        ref = om.match(init, scans[k], 0.5)
        _same(gm.match(sm.Rigid2d(*init), scans[k], 0.5), ref)
    ref = om.match_full_submap(scans[2], 0.97)
    assert not ref["found"]
    _same(gm.match_full_submap(scans[2], 0.97), ref)


@pytest.mark.parametrize("depth", [2, 3, 5])
def test_queue_search_shallow_trees(sm, oracle, synth, depth):
    """Depth 2 (the chain's first expansion already scores leaves) to 5, on a small grid."""
    cells, lim, world = synth.make_submap(7, 160, 160, 0.05, 12, 500, 30.0, 0.01)
    scan = world.scan(world.free_pose(5, 0.5), 700, 30.0, 0.01, 2)
    om = oracle.FastCorrelativeScanMatcher2D(cells, 0.05, lim["max_x"], lim["max_y"], depth)
    gm = _matcher(sm, cells, lim, depth)
    _same(gm.match_full_submap(scan, 0.5), om.match_full_submap(scan, 0.5))


def test_queue_search_all_ties(sm, oracle, debug):
    """A grid of unknown cells: every candidate of the search space ties.  The queue's sub-queues
    fill up, the level-synchronous path takes over from the bounds found so far, and the leaf
    the reference's depth-first order meets first comes back (ConstraintBuilder2DTest's grid,
    mapping/internal/constraints/constraint_builder_2d_test.cc:60-75)."""
    cells = np.zeros((110, 100), np.uint16)
    cloud = np.array([[0.1, 0.2, 0.3]], np.float32)
    om = oracle.FastCorrelativeScanMatcher2D(cells, 1.0, 2.0, 3.0, 7, 7.0, math.radians(30.0))
    gm = sm.FastCorrelativeScanMatcher2D(sm.Grid2D(cells, 1.0, 2.0, 3.0), 7,
                                         linear_search_window=7.0,
                                         angular_search_window=math.radians(30.0))
    _same(gm.match_full_submap(cloud, 0.0), om.match_full_submap(cloud, 0.0))
    _same(gm.match(sm.Rigid2d(4.0, 5.0, 0.0), cloud, 0.0), om.match([4.0, 5.0, 0.0], cloud, 0.0))


def test_queue_search_for_a_batch(sm, world8, synth, oracle, debug):
    """fast2d_queue = 1: the queue for a batch of five problems too (several problems' nodes in
    one set of sub-queues, work counters per problem)."""
    cells, lim, _, scans, refs = world8
    debug(fast2d_queue=1)
    grids = [(cells, lim)]
    for seed in (43, 44):
        c, l, _ = synth.make_submap(seed, 400, 400, 0.05, 30, 1000, 30.0, 0.01)
        grids.append((c, l))
    order = [0, 1, 0, 2, 0]
    matchers = [_matcher(sm, *grids[g]) for g in order]
    found, scores, poses, stats = sm.match_full_submap_batch(matchers, scans[1], 0.6)
    debug(fast2d_queue=2)
    found2, scores2, poses2, _ = sm.match_full_submap_batch(matchers, scans[1], 0.6)
    np.testing.assert_array_equal(found, found2)
    np.testing.assert_array_equal(scores[found != 0], scores2[found2 != 0])
    for a, b, ok in zip(poses, poses2, found):
        if ok:
            assert _xyt(a) == _xyt(b)
    assert found[0] and np.float32(scores[0]) == np.float32(refs[1]["score"])
    assert scores[0] == scores[2] == scores[4]


def test_queue_search_from_eight_threads(sm, world8):
    """Eight host threads, each cycling through the eight scans (the bench headline's shape):
    every call leases its own workspace, queue buffer and epoch; 160 searches, every one equal to
    the oracle's."""
    cells, lim, _, scans, refs = world8
    gm = _matcher(sm, cells, lim)
    clouds = [sm.PointCloudOnDevice(s) for s in scans]

    def worker(t):
        out = []
        for j in range(20):
            k = (t + j) % 8
            f, s, p, _ = sm.match_full_submap_batch([gm], clouds[k], 0.6)
            out.append((k, f[0], s[0], p[0]))
        return out
    with ThreadPoolExecutor(8) as pool:
        for results in pool.map(worker, range(8)):
            for k, f, s, p in results:
                _same((f, s, p), refs[k])


def test_large_batches_fan_out_with_the_same_results(sm, synth, debug):
    """cmx_fast2d_match_batch with 40 full-submap searches (per-problem thresholds): as shipped the
    batch runs as independent single searches over the host pool (fast2d_fanout, from 32 problems
    on); with the switch at 1 as level-synchronous launches over the whole batch.  Same found
    flags, scores, poses and candidate counts; a batch with ONE windowed search among them stays
    on the batch launches either way."""
    grids = [synth.make_submap(900 + k, 160, 160, 0.05, 12, 500, 30.0, 0.01) for k in range(5)]
    scan = grids[2][2].scan(grids[2][2].free_pose(5, 0.5), 600, 30.0, 0.01, 3)
    matchers = [_matcher(sm, grids[k % 5][0], grids[k % 5][1], 6) for k in range(40)]
    initial = [sm.Rigid2d(0.0, 0.0, 0.0)] * 40
    thresholds = [0.3 + 0.01 * (k % 7) for k in range(40)]
    out = sm.match_batch(matchers, initial, [1] * 40, thresholds, scan)
    debug(fast2d_fanout=1)
    ref = sm.match_batch(matchers, initial, [1] * 40, thresholds, scan)
    debug(fast2d_fanout=0)
    np.testing.assert_array_equal(out[0], ref[0])
    np.testing.assert_array_equal(out[1][ref[0] != 0], ref[1][ref[0] != 0])
    for a, b, ok in zip(out[2], ref[2], ref[0]):
        if ok:
            assert _xyt(a) == _xyt(b)
    assert out[0][2] == 1 and out[3]["coarse_candidates"] == ref[3]["coarse_candidates"]
    # one windowed search in the batch: no fan-out (the batch's own launches), same results
    truth = grids[2][2].free_pose(5, 0.5)
    flags = [1] * 40
    flags[2] = 0
    initial = list(initial)
    initial[2] = sm.Rigid2d(truth[0] + 0.2, truth[1] - 0.1, truth[2] + 0.05)
    mixed = sm.match_batch(matchers, initial, flags, thresholds, scan)
    assert mixed[0][2] == 1
    for k in (0, 7, 39):
        assert mixed[0][k] == ref[0][k] and (not ref[0][k] or mixed[1][k] == ref[1][k])

