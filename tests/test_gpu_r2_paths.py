"""Round-2 device paths against the oracle, old and new side by side.

  * fast 2D: the fused front end (PrepScoreFusedKernel: prep + bucketing + lowest-resolution
    scoring in one block per rotation) vs the separate launches (debug switch fast2d_unfused);
  * real-time 2D: the LDS-staged integer bulk pass + exact finalists (Rt2DBulkKernel /
    Rt2DFinishKernel) vs one thread per candidate (debug switch rt2d_legacy).
Both toggles are read per call, so one process runs both paths on identical inputs.  Bars as in
test_gpu_2d.py: integer work bit-exact, f32 scores bit-equal, poses to 1e-12.
"""
import math
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sm():
    from cartographer_amd import _lib, scan_matching
    assert _lib.lib().cmx_device_count() >= 1, "no HIP device: these tests need the GPU"
    return scan_matching


def _grid(sm, cells, lim):
    return sm.Grid2D(cells, lim["resolution"], lim["max_x"], lim["max_y"])


@pytest.fixture(scope="module")
def c2(synth):
    cells, lim, world = synth.make_submap(42, 400, 400, 0.05, 30, 1000, 30.0, 0.01)
    pose = world.free_pose(1234, 0.5)
    scan = world.scan(pose, 1000, 30.0, 0.01, 7)
    return cells, lim, world, pose, scan


@pytest.mark.parametrize("fused", ["1", "0"])
def test_fast2d_prepare_both_front_ends(sm, oracle, c2, debug, fused):
    debug(fast2d_unfused=1 if fused == "0" else 0)
    cells, lim, _, _, scan = c2
    om = oracle.FastCorrelativeScanMatcher2D(cells, 0.05, lim["max_x"], lim["max_y"], 7)
    gm = sm.FastCorrelativeScanMatcher2D(_grid(sm, cells, lim), 7)
    ref = om.prepare([0, 0, 0], scan, True)
    got = gm.debug_prepare(None, scan, True)
    assert got["num_scans"] == ref["num_scans"] and got["step"] == ref["step"]
    np.testing.assert_array_equal(got["scans"], ref["scans"])
    np.testing.assert_array_equal(got["bounds"], ref["bounds"])
    np.testing.assert_array_equal(got["sums"], ref["sums"])


def test_concurrent_native_callers_get_the_single_caller_results(sm, c2):
    """Four std::threads (csrc/host/thread_driver.cc: a C++ caller's thread pool in miniature)
    issue three full-submap searches each through the C ABI at once: every one finds the match
    and scores about as many candidates as a search on its own (workspaces, streams and scratch are per
    call)."""
    from cartographer_amd import synth
    cells, lim, _, _, scan = c2
    gm = sm.FastCorrelativeScanMatcher2D(_grid(sm, cells, lim), 7)
    cloud = sm.PointCloudOnDevice(scan)
    found, scores, poses, stats = sm.match_full_submap_batch([gm], cloud, 0.55)
    assert found[0] == 1
    seconds, candidates, hits = synth.threaded_full_submap_searches([gm], [cloud], 0.55, 4, 3)
    assert seconds > 0 and hits == 12
    # (how many nodes a search expands depends a little on when its blocks see the rising bound)
    assert abs(candidates - 12 * stats["candidates_scored"]) <= 0.02 * 12 * stats["candidates_scored"]


def test_timing_brackets_only_on_request(sm, c2, debug):
    """cmx_match_stats.*_ms: 0 by default (no HIP-event packets in the chain of launches), filled
    after cmx_debug_set("timing", 1); the result and the counters are the same either way."""
    cells, lim, _, _, scan = c2
    gm = sm.FastCorrelativeScanMatcher2D(_grid(sm, cells, lim), 7)
    plain = gm.match_full_submap(scan, 0.55)
    quiet = dict(gm.last_stats)
    debug(timing=1)
    timed = gm.match_full_submap(scan, 0.55)
    loud = dict(gm.last_stats)
    assert plain[0] and timed[0] and plain[1] == timed[1]
    assert quiet["device_ms"] == 0.0 and quiet["dominant_kernel_ms"] == 0.0
    assert loud["device_ms"] > 0.0 and 0.0 < loud["dominant_kernel_ms"] < loud["device_ms"]
    # (candidates_scored: the tree search's share depends on when the bound rises)
    for key in ("coarse_candidates", "num_scans"):
        assert quiet[key] == loud[key] > 0
    assert quiet["candidates_scored"] > 0 and loud["candidates_scored"] > 0


def test_timing_switch_flipped_while_calls_are_in_flight(sm, c2, debug):
    """The switch is process-wide and unsynchronised: flipped by one thread while another is inside
    a match, that call finds events that were never recorded.  It reports no timing for itself and
    returns its result -- it does not fail."""
    import threading
    cells, lim, _, _, scan = c2
    gm = sm.FastCorrelativeScanMatcher2D(_grid(sm, cells, lim), 7)
    want = gm.match_full_submap(scan, 0.55)
    stop = threading.Event()

    def flip():
        k = 0
        while not stop.is_set():
            k += 1
            debug(timing=k & 1)

    flipper = threading.Thread(target=flip)
    flipper.start()
    try:
        for _ in range(300):
            assert gm.match_full_submap(scan, 0.55) == want
    finally:
        stop.set()
        flipper.join()
        debug(timing=0)


@pytest.mark.parametrize("fused", ["1", "0"])
@pytest.mark.parametrize("depth,n", [(7, 1000), (5, 333), (3, 64), (6, 1)])
def test_fast2d_match_both_front_ends(sm, oracle, c2, debug, fused, depth, n):
    debug(fast2d_unfused=1 if fused == "0" else 0)
    cells, lim, _, truth, scan = c2
    cloud = scan[:n]
    om = oracle.FastCorrelativeScanMatcher2D(cells, 0.05, lim["max_x"], lim["max_y"], depth, 2.0,
                                             math.radians(25.0))
    gm = sm.FastCorrelativeScanMatcher2D(_grid(sm, cells, lim), depth, 2.0, math.radians(25.0))
    for full, init, min_score in [(True, None, 0.55), (False, (truth[0] + 0.4, truth[1] - 0.3,
                                                               truth[2] + 0.2), 0.3)]:
        if full:
            ref = om.match_full_submap(cloud, min_score)
            found, score, pose = gm.match_full_submap(cloud, min_score)
        else:
            ref = om.match(list(init), cloud, min_score)
            found, score, pose = gm.match(sm.Rigid2d(*init), cloud, min_score)
        assert found == ref["found"]
        if found:
            assert np.float32(score) == np.float32(ref["score"])
            np.testing.assert_allclose([pose.x, pose.y, pose.theta], ref["pose"], rtol=0,
                                       atol=1e-12)
        assert gm.last_stats["coarse_candidates"] == ref["coarse_candidates"]


def test_fast2d_fused_and_unfused_batches_agree(sm, synth, c2, debug):
    """A ConstraintBuilder batch (windowed and full-submap pairs mixed) gives the same
    constraint list through either front end, and the min_score gate on the discretised
    scans (only scans that can still matter are kept for the tree search) changes nothing."""
    _, _, _, truth, scan = c2
    matchers = []
    for seed in (42, 43, 44, 45, 46):
        cells, lim, _ = synth.make_submap(seed, 400, 400, 0.05, 30, 1000, 30.0, 0.01)
        matchers.append(sm.FastCorrelativeScanMatcher2D(_grid(sm, cells, lim), 7, 3.0,
                                                        math.radians(20.0)))
    initial = [sm.Rigid2d(truth[0] + 0.3 * k, truth[1] - 0.2 * k, truth[2] + 0.05 * k)
               for k in range(5)]
    full = [1, 0, 1, 0, 1]
    thresholds = [0.6, 0.4, 0.9, 0.55, 0.3]
    out = {}
    for fused in ("1", "0"):
        debug(fast2d_unfused=1 if fused == "0" else 0)
        out[fused] = sm.match_batch(matchers, initial, full, thresholds, scan)
    a, b = out["1"], out["0"]
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1][a[0] != 0], b[1][b[0] != 0])
    for pa, pb, f in zip(a[2], b[2], a[0]):
        if f:
            assert (pa.x, pa.y, pa.theta) == (pb.x, pb.y, pb.theta)
    assert a[3]["coarse_candidates"] == b[3]["coarse_candidates"]


# ----------------------------------------------------------------------------
# Real-time 2D: bulk pass vs per-candidate kernels vs oracle
# ----------------------------------------------------------------------------
RT2D_PATHS = ["bounds", "bounds_img3", "bounds2", "bounds1", "tiles", "tiles1", "tiles64", "tiles56g", "0"]


def _rt2d_path(debug, path):
    """'bounds' (round 5): upper bounds of 2 x 2 blocks of translations from a max-pooled
    image first, sums only for the blocks that reach the best lower bound (rt_2d_bounds.h; windows
    beyond 16 x 16 cells take the next path) -- since round 6 as two kernels, the bounds of every
    match (blocks of 4 x 4 translations) and then the tails of every match (the sixteen
    candidates of every surviving block summed); 'bounds2': two kernels, 2 x 2 blocks as the
    first level; 'bounds1': 2 x 2 bounds and tail of a match by one workgroup in one kernel
    (round 5's shape); 'tiles': the exhaustive integer bulk pass out of LDS
    tiles + exact finalists (one tile per match
    where it fits, discretised inside the tile kernel); 'tiles1': that shape through the prep
    kernel and its planner; 'tiles64' / 'tiles56g': the same with small tiles (up to 4 x 4 per match: sums meet by atomics) and
    with one work item per (tile, rotation); '0': one thread per candidate."""
    if path == "bounds":     # block bounds first where the window is at most 16 x 16 (the
        debug(rt2d_bounds=1)  # default from 192 matches per call on; here: calls of any size)
        return
    if path == "bounds_img3":   # (the grid's derived images by the three kernels Rt2DImageKernel replaced)
        debug(rt2d_bounds=1, rt2d_image_kernels=1, rt2d_no_image_cache=1)
        return
    if path == "bounds2":
        debug(rt2d_bounds=1, rt2d_bounds_level=2)
        return
    if path == "bounds1":
        debug(rt2d_bounds=1, rt2d_bounds_fused=1)
        return
    if path == "0":
        debug(rt2d_legacy=1)
    elif path == "tiles":
        debug(rt2d_no_bounds=1)
    elif path == "tiles1":
        debug(rt2d_unfused=1)
    elif path == "tiles64":
        debug(rt2d_tile=64, rt2d_no_image_cache=1)
    elif path == "tiles56g":
        debug(rt2d_tile=56, rt2d_groups=1000, rt2d_target=1)


@pytest.mark.parametrize("bulk", RT2D_PATHS)
@pytest.mark.parametrize("seed,size,beams,lin,ang,weights", [
    (42, 200, 1000, 0.3, 7.0, (0.1, 0.1)),      # C1
    (7, 200, 400, 0.3, 7.0, (0.0, 0.0)),        # unweighted: ties resolved by generation order
    (11, 120, 300, 0.2, 4.0, (10.0, 1.0)),
    (3, 160, 250, 0.15, 10.0, (0.1, 5.0)),
    (5, 97, 61, 0.55, 1.0, (0.1, 0.1)),         # odd row length, 23 x 23 window: 3 lane slices
    (9, 64, 64, 0.0, 0.0, (0.1, 0.1)),          # the single-candidate window
    (13, 150, 700, 0.1, 20.0, (0.1, 0.1)),      # trajectory_builder_2d.lua's own window: 5 x 5, B = 2
    (17, 130, 333, 0.2, 6.0, (0.1, 0.1)),       # 9 x 9: B = 3, pitch a multiple of 8 only
    (19, 110, 500, 0.4, 3.0, (0.1, 0.1)),       # 17 x 17: B = 5, three rows per lane
])
def test_rt2d_both_paths(sm, oracle, synth, debug, bulk, seed, size, beams, lin, ang,
                         weights):
    _rt2d_path(debug, bulk)
    ny = size if size != 97 else 83
    cells, lim, world = synth.make_submap(seed, size, ny, 0.05, 20, 600, 5.0, 0.01)
    pose = world.free_pose(seed + 100, 0.5)
    scan = world.scan(pose, beams, 5.0, 0.01, 7)
    init = [pose[0] + 0.12, pose[1] - 0.08, pose[2] + math.radians(3.0)]
    ref = oracle.rt2d_match(cells, 0.05, lim["max_x"], lim["max_y"], init, scan, lin,
                            math.radians(ang), *weights)
    m = sm.RealTimeCorrelativeScanMatcher2D(lin, math.radians(ang), *weights)
    score, est = m.match(sm.Rigid2d(*init), scan, _grid(sm, cells, lim))
    assert m.last_stats["candidates_scored"] == ref["num_candidates"]
    assert score == ref["score"]
    np.testing.assert_allclose([est.x, est.y, est.theta], ref["pose"], rtol=0, atol=1e-12)
    # the path asked for is the path that ran (no silent fall-back to the per-candidate kernels):
    # only the tile path re-sums candidates with exact integers, and it scores a handful with
    # the f32 chain where the per-candidate kernels score everything
    st = m.last_stats
    if bulk == "0":
        assert st["refined_candidates"] == 0 and st["finalists"] == ref["num_candidates"]
    else:
        assert 1 <= st["finalists"] <= st["refined_candidates"] < ref["num_candidates"] or \
            ref["num_candidates"] <= 64
    # block bounds: the device summed a fraction of the search space (a bound per 2 x 2 block +
    # four candidates per surviving block), the exhaustive paths all of it
    side = 2 * math.ceil(lin / 0.05 - 1e-9) + 1
    if bulk in ("bounds", "bounds_img3") and 1 < side <= 16:      # (4 x 4 blocks: a bound per block + sixteen candidates per survivor)
        assert st["coarse_candidates"] < 0.75 * ref["num_candidates"], st
    elif bulk in ("bounds2", "bounds1") and 1 < side <= 16:
        assert st["coarse_candidates"] < 0.6 * ref["num_candidates"], st
    elif bulk not in ("0", "bounds", "bounds_img3", "bounds2", "bounds1") or side > 16:
        assert st["coarse_candidates"] == ref["num_candidates"]


@pytest.mark.parametrize("level", [4, 2])
@pytest.mark.parametrize("seed,size,beams,lin,ang,weights", [
    (42, 200, 1000, 0.3, 7.0, (0.1, 0.1)),      # C1: 7 x 7 blocks per rotation
    (7, 200, 400, 0.3, 7.0, (0.0, 0.0)),
    (13, 150, 700, 0.1, 20.0, (0.1, 0.1)),      # 5 x 5 window: 3 x 3 blocks, the last one a single row / column
    (3, 160, 250, 0.15, 10.0, (0.1, 5.0)),      # 7 x 7: 4 x 4 blocks
    (21, 90, 300, 0.35, 2.0, (0.1, 0.1)),       # 15 x 15: 8 x 8 blocks (the widest row read), a small grid
    (23, 64, 200, 0.05, 30.0, (0.1, 0.1)),      # 3 x 3: 2 x 2 blocks, many rotations
])
def test_rt2d_block_bounds_dominate_their_candidates(sm, oracle, synth, debug, seed, size, beams,
                                                      lin, ang, weights, level):
    """The invariant the pruning of rt_2d_bounds.h rests on, checked on the device for EVERY block
    of the search space (debug switch rt2d_bounds_verify: all blocks are summed; the call fails
    if the weighted bound of a block lies below the weighted value of one of its own candidates):
    the max-pooled byte image, the parity planes and their addressing, the weights' maxima.
    level 4 (round 6): the 4 x 4 blocks the default path bounds first; level 2: 2 x 2 blocks."""
    debug(rt2d_bounds_verify=1, rt2d_bounds=1, rt2d_bounds_level=level)
    cells, lim, world = synth.make_submap(seed, size, size, 0.05, 20, 600, 5.0, 0.01)
    pose = world.free_pose(seed + 100, 0.5)
    scan = world.scan(pose, beams, 5.0, 0.01, 7)
    init = [pose[0] + 0.12, pose[1] - 0.08, pose[2] + math.radians(3.0)]
    ref = oracle.rt2d_match(cells, 0.05, lim["max_x"], lim["max_y"], init, scan, lin,
                            math.radians(ang), *weights)
    m = sm.RealTimeCorrelativeScanMatcher2D(lin, math.radians(ang), *weights)
    score, est = m.match(sm.Rigid2d(*init), scan, _grid(sm, cells, lim))
    assert score == ref["score"]
    np.testing.assert_allclose([est.x, est.y, est.theta], ref["pose"], rtol=0, atol=1e-12)
    side = 2 * math.ceil(lin / 0.05 - 1e-9) + 1
    if level == 2:
        blocks = ref["num_candidates"] // (side * side) * ((side + 1) // 2) ** 2
        assert m.last_stats["coarse_candidates"] == 5 * blocks    # every bound + every block summed
    else:
        blocks = ref["num_candidates"] // (side * side) * ((side + 3) // 4) ** 2
        assert m.last_stats["coarse_candidates"] == blocks + ref["num_candidates"]   # every bound + every candidate


@pytest.mark.parametrize("bulk", RT2D_PATHS)
def test_rt2d_points_outside_and_unknown_grid(sm, oracle, debug, bulk):
    """A cloud that mostly falls outside a small grid, on an all-unknown grid and on a
    random one: the flat landscape makes every candidate a finalist (more than the list
    holds: the bulk path hands the batch to the per-candidate kernels)."""
    _rt2d_path(debug, bulk)
    rng = np.random.default_rng(5)
    scan = np.zeros((130, 3), np.float32)
    scan[:, :2] = rng.uniform(-4.0, 4.0, (130, 2))
    for cells in (np.zeros((50, 70), np.uint16),
                  rng.integers(0, 32768, (50, 70)).astype(np.uint16)):
        init = [1.0, 1.2, 0.3]
        ref = oracle.rt2d_match(cells, 0.05, 3.5, 2.5, init, scan, 0.5, 0.2, 0.0, 0.0)
        m = sm.RealTimeCorrelativeScanMatcher2D(0.5, 0.2, 0.0, 0.0)
        score, est = m.match(sm.Rigid2d(*init), scan, sm.Grid2D(cells, 0.05, 3.5, 2.5))
        assert score == ref["score"]
        np.testing.assert_allclose([est.x, est.y, est.theta], ref["pose"], rtol=0, atol=1e-12)


def test_rt2d_bound_kernel_with_two_window_classes_in_one_batch(sm, synth, debug):
    """One batch, two grid resolutions: 0.3 m is 13 x 13 candidates on the 5 cm grids (7 x 7 blocks)
    and 7 x 7 on the 10 cm grids (4 x 4 blocks).  The bound kernel is instantiated for the larger
    count and the smaller matches leave its outer blocks without a candidate; grids of three sizes,
    clouds of 60 ... 1100 points.  Every match equals the exhaustive path's result for it."""
    from cartographer_amd import grid_2d
    m = sm.RealTimeCorrelativeScanMatcher2D(0.3, math.radians(5.0), 0.1, 0.1)
    worlds = []
    for k, (res, size) in enumerate([(0.05, 200), (0.1, 120), (0.05, 140), (0.1, 90)]):
        cells, lim, world = synth.make_submap(80 + k, size, size, res, 20, 600, 5.0, 0.01)
        worlds.append((grid_2d.ProbabilityGridOnDevice(res, (lim["max_x"], lim["max_y"]), size, size,
                                                       cells=cells), world))
    grids, inits, scans = [], [], []
    for k in range(208):
        grid, world = worlds[k % 4]
        pose = world.free_pose(500 + k, 0.5)
        grids.append(grid)
        scans.append(world.scan(pose, 60 + 5 * k, 5.0, 0.01, k))
        inits.append(sm.Rigid2d(pose[0] + 0.08, pose[1] - 0.06, pose[2] - 0.03))
    debug(rt2d_no_bounds=1)
    singles = [m.match(inits[k], scans[k], grids[k]) for k in range(208)]
    from cartographer_amd import _lib
    _lib.debug_reset()
    scores, poses, stats = sm.rt2d_match_batch(m, grids, inits, scans)
    for k, (score, pose) in enumerate(singles):
        assert scores[k] == score, k
        assert (poses[k].x, poses[k].y, poses[k].theta) == (pose.x, pose.y, pose.theta), k
    assert stats["coarse_candidates"] < 0.6 * stats["candidates_scored"], stats   # the bounds ran


def test_rt2d_block_bounds_on_flat_landscapes(sm, oracle, debug):
    """Score landscapes the bounds cannot prune, inside the bound kernel's window limit (13 x 13):
    an all-unknown grid -- every candidate ties, every block reaches the lower bound, more blocks
    than the kernel lists: it says so and the match is repeated on the per-candidate kernels -- and
    a random grid, where a few dozen blocks survive and are summed.  Results = the oracle's,
    first-maximum rule included."""
    debug(rt2d_bounds=1)
    rng = np.random.default_rng(11)
    scan = np.zeros((200, 3), np.float32)
    scan[:, :2] = rng.uniform(-1.5, 1.5, (200, 2))
    for cells, weights in ((np.zeros((80, 90), np.uint16), (0.0, 0.0)),
                           (np.zeros((80, 90), np.uint16), (0.1, 0.1)),
                           (rng.integers(0, 32768, (80, 90)).astype(np.uint16), (0.0, 0.0)),
                           (rng.integers(1, 32768, (80, 90)).astype(np.uint16), (0.1, 0.1))):
        init = [2.0, 2.3, 0.3]
        ref = oracle.rt2d_match(cells, 0.05, 4.5, 4.0, init, scan, 0.3, 0.1, *weights)
        m = sm.RealTimeCorrelativeScanMatcher2D(0.3, 0.1, *weights)
        score, est = m.match(sm.Rigid2d(*init), scan, sm.Grid2D(cells, 0.05, 4.5, 4.0))
        assert score == ref["score"]
        np.testing.assert_allclose([est.x, est.y, est.theta], ref["pose"], rtol=0, atol=1e-12)
        assert m.last_stats["candidates_scored"] == ref["num_candidates"]


def test_rt2d_batch_on_resident_grids_both_paths(sm, oracle, synth, debug):
    from cartographer_amd import grid_2d
    m = sm.RealTimeCorrelativeScanMatcher2D(0.3, math.radians(7.0), 0.1, 0.1)
    grids, inits, scans, refs = [], [], [], []
    for k in range(20):
        cells, lim, world = synth.make_submap(60 + k % 5, 200, 200, 0.05, 20, 600, 5.0, 0.01)
        pose = world.free_pose(300 + k, 0.5)
        scan = world.scan(pose, 500 + 17 * k, 5.0, 0.01, k)
        init = [pose[0] + 0.1, pose[1] - 0.05, pose[2] + 0.04]
        grids.append(grid_2d.ProbabilityGridOnDevice(0.05, (lim["max_x"], lim["max_y"]), 200, 200,
                                                     cells=cells))
        inits.append(sm.Rigid2d(*init))
        scans.append(scan)
        refs.append(oracle.rt2d_match(cells, 0.05, lim["max_x"], lim["max_y"], init, scan, 0.3,
                                      math.radians(7.0), 0.1, 0.1))
    from cartographer_amd import _lib
    for bulk in RT2D_PATHS:
        _lib.debug_reset()
        _rt2d_path(debug, bulk)
        scores, poses, stats = sm.rt2d_match_batch(m, grids, inits, scans)
        for k, ref in enumerate(refs):
            assert scores[k] == ref["score"], (bulk, k)
            np.testing.assert_allclose([poses[k].x, poses[k].y, poses[k].theta], ref["pose"],
                                       rtol=0, atol=1e-12)
    _lib.debug_reset()
    # the prepared form of the same call (argument arrays built once), twice in a row
    batch = sm.Rt2DBatch(m, grids, scans)
    init = np.array([[p.x, p.y, p.theta] for p in inits])
    for _ in range(2):
        scores, poses, stats = batch.match(init)
        assert stats["candidates_scored"] > 0
        for k, ref in enumerate(refs):
            assert scores[k] == ref["score"], k
            np.testing.assert_allclose(poses[k], ref["pose"], rtol=0, atol=1e-12)


def test_rt2d_a_batch_of_two_hundred_takes_the_bound_kernel(sm, synth, debug):
    """From 192 matches per call on the block bounds are the default (rt_2d_bounds.h: the bound
    kernel also finishes its matches, one launch per call): 200 matches over five grids with
    scans of different sizes return, match by match, what the exhaustive tile kernel returns for
    them one by one, and the statistics show that a fraction of the search space was summed."""
    from cartographer_amd import grid_2d
    m = sm.RealTimeCorrelativeScanMatcher2D(0.3, math.radians(7.0), 0.1, 0.1)
    worlds = []
    for k in range(5):
        cells, lim, world = synth.make_submap(70 + k, 200, 200, 0.05, 20, 600, 5.0, 0.01)
        worlds.append((grid_2d.ProbabilityGridOnDevice(0.05, (lim["max_x"], lim["max_y"]), 200, 200,
                                                       cells=cells), world))
    grids, inits, scans = [], [], []
    for k in range(200):
        grid, world = worlds[k % 5]
        pose = world.free_pose(400 + k, 0.5)
        grids.append(grid)
        scans.append(world.scan(pose, 250 + 3 * k, 5.0, 0.01, k))
        inits.append(sm.Rigid2d(pose[0] + 0.1, pose[1] - 0.05, pose[2] + 0.04))
    debug(rt2d_no_bounds=1)
    singles = [m.match(inits[k], scans[k], grids[k]) for k in range(200)]
    from cartographer_amd import _lib
    _lib.debug_reset()
    batch = sm.Rt2DBatch(m, grids, scans, resident=True)
    init = np.array([[p.x, p.y, p.theta] for p in inits])
    for _ in range(2):
        scores, poses, stats = batch.match(init)
        for k, (score, pose) in enumerate(singles):
            assert scores[k] == score, k
            np.testing.assert_array_equal(poses[k], [pose.x, pose.y, pose.theta])
        assert stats["coarse_candidates"] < 0.6 * stats["candidates_scored"], stats
        assert 200 <= stats["finalists"] <= stats["refined_candidates"] < 0.1 * stats["candidates_scored"]


@pytest.mark.parametrize("resident", [False, True])
@pytest.mark.parametrize("parts", [2, 3])
def test_rt2d_large_batch_goes_out_as_two_halves(sm, synth, debug, resident, parts):
    """A large batch is issued in parts (own workspace and stream each, planned and launched one
    after the other by the calling thread, collected in order; from 192 matches on by default,
    forced here): 71 matches over five grids with scans of different sizes return, match by
    match, what the single-match entry point returns; the statistics are those of all parts; an
    error in the last part comes back as the call's status with its message."""
    debug(rt2d_parts=parts)
    from cartographer_amd import grid_2d
    from cartographer_amd._lib import CmxError
    m = sm.RealTimeCorrelativeScanMatcher2D(0.3, math.radians(7.0), 0.1, 0.1)
    worlds = []
    for k in range(5):
        cells, lim, world = synth.make_submap(60 + k, 200, 200, 0.05, 20, 600, 5.0, 0.01)
        worlds.append((grid_2d.ProbabilityGridOnDevice(0.05, (lim["max_x"], lim["max_y"]), 200, 200,
                                                       cells=cells), world))
    grids, inits, scans = [], [], []
    for k in range(71):
        grid, world = worlds[k % 5]
        pose = world.free_pose(300 + k, 0.5)
        grids.append(grid)
        scans.append(world.scan(pose, 300 + 9 * k, 5.0, 0.01, k))
        inits.append(sm.Rigid2d(pose[0] + 0.1, pose[1] - 0.05, pose[2] + 0.04))
    singles, candidates = [], 0
    for k in range(71):
        singles.append(m.match(inits[k], scans[k], grids[k]))
        candidates += m.last_stats["candidates_scored"]
    batch = sm.Rt2DBatch(m, grids, scans, resident=resident)
    init = np.array([[p.x, p.y, p.theta] for p in inits])
    for _ in range(2):
        scores, poses, stats = batch.match(init)
        for k, (score, pose) in enumerate(singles):
            assert scores[k] == score, k
            np.testing.assert_array_equal(poses[k], [pose.x, pose.y, pose.theta])
        assert stats["candidates_scored"] == candidates          # every part counted
    if not resident:
        bad = list(scans)
        bad[70] = np.zeros((0, 3), np.float32)            # an empty scan in the LAST part
        with pytest.raises(CmxError):
            sm.rt2d_match_batch(m, grids, inits, bad)


@pytest.mark.parametrize("count", [260, 530])
def test_rt2d_default_schedule_of_a_large_batch(sm, synth, debug, count):
    """Round 6: a call of 256 matches and more goes out in parts of DIFFERENT sizes (from 512 on a
    small first part, then 4 : 3), the next part prepared on the helper lane -- search parameters,
    plan, rotation tables over the host pool -- while the calling thread issues the current one,
    every part through the two bound kernels.  Every match equals the exhaustive tile kernel's
    single-match result, three calls in a row (the lane and its workers are reused), and the
    same with the lane switched off."""
    from cartographer_amd import grid_2d
    from cartographer_amd import _lib
    m = sm.RealTimeCorrelativeScanMatcher2D(0.3, math.radians(7.0), 0.1, 0.1)
    worlds = []
    for k in range(6):
        cells, lim, world = synth.make_submap(90 + k, 160, 160, 0.05, 20, 600, 5.0, 0.01)
        worlds.append((grid_2d.ProbabilityGridOnDevice(0.05, (lim["max_x"], lim["max_y"]), 160, 160,
                                                       cells=cells), world))
    grids, inits, scans = [], [], []
    for k in range(count):
        grid, world = worlds[k % 6]
        pose = world.free_pose(700 + k % 40, 0.5)
        grids.append(grid)
        scans.append(world.scan(pose, 120 + 7 * (k % 50), 5.0, 0.01, k % 40))
        inits.append(sm.Rigid2d(pose[0] + 0.02 * (k % 7), pose[1] - 0.015 * (k % 5), pose[2] + 0.01 * (k % 9)))
    debug(rt2d_no_bounds=1)
    singles = [m.match(inits[k], scans[k], grids[k]) for k in range(count)]
    _lib.debug_reset()
    batch = sm.Rt2DBatch(m, grids, scans, resident=True)
    init = np.array([[p.x, p.y, p.theta] for p in inits])
    for lane_off in (0, 0, 0, 1):
        debug(rt2d_no_lane=lane_off)
        scores, poses, stats = batch.match(init)
        for k, (score, pose) in enumerate(singles):
            assert scores[k] == score, (lane_off, k)
            np.testing.assert_array_equal(poses[k], [pose.x, pose.y, pose.theta])
        assert stats["coarse_candidates"] < 0.75 * stats["candidates_scored"], stats   # the bounds ran


def test_rt2d_a_flat_match_is_repeated_on_its_own(sm, oracle, synth, debug):
    """A batch through the bound kernels in which three matches have landscapes too flat for the
    lists (all-unknown grids: every candidate ties): those three -- and only those -- are
    repeated on the per-candidate kernels (until round 6: their whole part); every result is the
    oracle's, first-maximum rule included, and the statistics show the bounds at work for the
    rest."""
    m = sm.RealTimeCorrelativeScanMatcher2D(0.3, math.radians(5.0), 0.1, 0.1)
    cells, lim, world = synth.make_submap(33, 160, 160, 0.05, 20, 600, 5.0, 0.01)
    unknown = np.zeros_like(cells)
    from cartographer_amd import grid_2d
    on_device = {id(c): grid_2d.ProbabilityGridOnDevice(0.05, (lim["max_x"], lim["max_y"]), 160, 160, cells=c)
                 for c in (cells, unknown)}
    grids, inits, scans, refs = [], [], [], []
    for k in range(100):
        pose = world.free_pose(800 + k, 0.5)
        scan = world.scan(pose, 150 + 3 * k, 5.0, 0.01, k)
        init = [pose[0] + 0.07, pose[1] - 0.04, pose[2] + 0.02]
        use = unknown if k in (5, 50, 99) else cells
        grids.append(on_device[id(use)])
        inits.append(sm.Rigid2d(*init))
        scans.append(scan)
        refs.append(oracle.rt2d_match(use, 0.05, lim["max_x"], lim["max_y"], init, scan, 0.3,
                                      math.radians(5.0), 0.1, 0.1))
    scores, poses, stats = sm.rt2d_match_batch(m, grids, inits, scans)
    for k, ref in enumerate(refs):
        assert scores[k] == ref["score"], k
        np.testing.assert_allclose([poses[k].x, poses[k].y, poses[k].theta], ref["pose"], rtol=0, atol=1e-12)
    per_match = refs[0]["num_candidates"]
    # (three matches scored every candidate with the f32 chain, the others a handful)
    assert 3 * per_match <= stats["finalists"] < 3 * per_match + 97 * 40, stats
    assert stats["candidates_scored"] == sum(r["num_candidates"] for r in refs)


# ----------------------------------------------------------------------------
# Ceres refinement on the device vs the oracle's restatement
# ----------------------------------------------------------------------------
def _assert_ceres_close(pose, summary, ref):
    np.testing.assert_allclose([pose.x, pose.y, pose.theta], ref["pose"], rtol=0, atol=1e-6)
    assert abs(summary["initial_cost"] - ref["initial_cost"]) <= 1e-9 * max(1.0, ref["initial_cost"])
    assert abs(summary["final_cost"] - ref["final_cost"]) <= 1e-7 * max(1.0, ref["final_cost"])
    assert summary["termination"] == ref["termination"]
    assert summary["num_successful_steps"] == ref["num_successful_steps"]
    assert summary["num_unsuccessful_steps"] == ref["num_unsuccessful_steps"]


@pytest.mark.parametrize("init", [(-0.5, 0.5), (-0.3, 0.5), (-0.3, 0.3)])
def test_ceres2d_reference_fixture(sm, oracle, synth, init):
    """CeresScanMatcherTest (ceres_scan_matcher_2d_test.cc:34-112) on the device: the
    reference's tolerances (pose 1e-2, final cost 1e-2 around 0) and the oracle's iterates."""
    g = synth.ProbabilityGrid(1.0, (10.0, 10.0), 20, 20)
    g.set_probability(7, 13, 0.9)
    lim = g.limits
    cloud = np.array([[-3.0, 2.0, 0.0]], np.float32)
    m = sm.CeresScanMatcher2D(1.0, 0.1, 1.5, True, 50)
    pose, summary = m.match(init, sm.Rigid2d(init[0], init[1], 0.0), cloud,
                            sm.Grid2D(g.cells, 1.0, lim["max_x"], lim["max_y"]))
    assert abs(summary["final_cost"]) < 1e-2
    assert math.hypot(pose.x + 0.5, pose.y - 0.5) < 1e-2 and abs(pose.theta) < 1e-2


@pytest.mark.parametrize("seed,weights,nonmono,iters", [
    (9, (20.0, 10.0, 1.0), True, 10),        # pose_graph.lua constraint_builder.ceres_scan_matcher
    (4, (1.0, 10.0, 40.0), False, 20),       # trajectory_builder_2d.lua ceres_scan_matcher
    (17, (5.0, 1.0, 1.0), False, 50)])
def test_ceres2d_equals_the_oracle(sm, oracle, synth, seed, weights, nonmono, iters):
    from cartographer_amd import grid_2d
    cells, lim, world = synth.make_submap(seed, 200, 200, 0.05, 25, 800, 10.0, 0.01)
    truth = world.free_pose(seed + 5, 0.5)
    scan = world.scan(truth, 300, 10.0, 0.01, 4)
    init = (truth[0] + 0.04, truth[1] - 0.03, truth[2] + 0.015)
    target = (init[0] + 0.01, init[1] - 0.01)
    ref = oracle.ceres2d_match(cells, 0.05, lim["max_x"], lim["max_y"], target, init, scan,
                               weights[0], weights[1], weights[2], nonmono, iters)
    m = sm.CeresScanMatcher2D(weights[0], weights[1], weights[2], nonmono, iters)
    pose, summary = m.match(target, sm.Rigid2d(*init), scan, _grid(sm, cells, lim))
    _assert_ceres_close(pose, summary, ref)
    dev = grid_2d.ProbabilityGridOnDevice(0.05, (lim["max_x"], lim["max_y"]), 200, 200, cells=cells)
    pose2, summary2 = m.match(target, sm.Rigid2d(*init), scan, dev)
    assert (pose2.x, pose2.y, pose2.theta) == (pose.x, pose.y, pose.theta)
    assert summary2 == summary


def test_fast2d_match_then_refine_batch(sm, oracle, synth, c2):
    """ComputeConstraint's pair (fast correlative match, then Ceres) for a batch of submaps:
    the refined poses equal the oracle's refinement of the oracle's search results; pairs
    without a match pass through."""
    _, _, _, truth, scan = c2
    matchers, grids = [], []
    for seed in (42, 43, 44):
        cells, lim, _ = synth.make_submap(seed, 400, 400, 0.05, 30, 1000, 30.0, 0.01)
        matchers.append(sm.FastCorrelativeScanMatcher2D(_grid(sm, cells, lim), 7))
        grids.append((cells, lim))
    found, scores, poses, _ = sm.match_full_submap_batch(matchers, scan, 0.6)
    assert found[0] == 1
    ceres = sm.CeresScanMatcher2D(20.0, 10.0, 1.0, True, 10)
    pose_list = [sm.Rigid2d(*p) for p in poses]
    refined, summaries = ceres.refine_batch(matchers, found, pose_list, scan)
    for k, (cells, lim) in enumerate(grids):
        if not found[k]:
            assert (refined[k].x, refined[k].y, refined[k].theta) == tuple(poses[k])
            continue
        ref = oracle.ceres2d_match(cells, 0.05, lim["max_x"], lim["max_y"], poses[k][:2],
                                   poses[k], scan, 20.0, 10.0, 1.0, True, 10)
        _assert_ceres_close(refined[k], summaries[k], ref)
        assert summaries[k]["final_cost"] <= summaries[k]["initial_cost"]


# ----------------------------------------------------------------------------
# cmx_comm: the sharded entry points on the devices this box has
# ----------------------------------------------------------------------------
@pytest.mark.parametrize("virtual_ranks", [0, 2, 3])
def test_sharded_match_equals_the_batch(sm, synth, c2, debug, virtual_ranks):
    """cmx_fast2d_match_sharded over a communicator of every visible device (one on the test
    box): same constraint list as cmx_fast2d_match_batch, and the all-reduce returns the best
    found pair (lowest index among equal scores).  virtual_ranks = N: the one device as N ranks
    (debug switch comm_virtual_ranks) -- N device workers, the five submaps dealt to them by
    index range, each rank's batch on its own stream, the packed keys of all ranks reduced (on
    the host: RCCL refuses one device twice): the fan-out, the result order and the key
    arithmetic run with world = N on a one-GPU box."""
    import torch
    from cartographer_amd import sharding
    _, _, _, truth, scan = c2
    ndev = torch.cuda.device_count()
    if virtual_ranks:
        debug(comm_virtual_ranks=virtual_ranks)
        comm = sharding.Communicator([0])
        assert comm.num_devices == virtual_ranks
        ndev = 1
    else:
        comm = sharding.Communicator(list(range(ndev)))
    matchers = []
    seeds = (42, 43, 42, 44, 42)           # submaps 0, 2, 4 are identical: equal scores
    for k, seed in enumerate(seeds):
        cells, lim, _ = synth.make_submap(seed, 400, 400, 0.05, 30, 1000, 30.0, 0.01)
        dev = comm.device_of(k, len(seeds))
        assert 0 <= dev < ndev
        matchers.append(sm.FastCorrelativeScanMatcher2D(_grid(sm, cells, lim), 7, 3.0,
                                                        math.radians(20.0), device=dev))
    initial = [sm.Rigid2d(truth[0] + 0.2, truth[1] - 0.1, truth[2] + 0.03)] * len(seeds)
    full = [1, 1, 0, 1, 1]
    thresholds = [0.6, 0.6, 0.5, 0.6, 0.6]
    f, s, p, best, stats = comm.match_batch(matchers, initial, full, thresholds, scan)
    if ndev == 1:
        f1, s1, p1, stats1 = sm.match_batch(matchers, initial, full, thresholds, scan)
        np.testing.assert_array_equal(f, f1)
        np.testing.assert_array_equal(s[f != 0], s1[f1 != 0])
        for a, b, ok in zip(p, p1, f):
            if ok:
                assert (a.x, a.y, a.theta) == (b.x, b.y, b.theta)
        if virtual_ranks:      # (smaller batches per rank: the bound rises at other moments, and
            # a rank's two or three problems go through the work-queue search, whose chains
            # expand fewer nodes than the level-synchronous launches the batch of five takes)
            assert abs(stats["candidates_scored"] - stats1["candidates_scored"]) < \
                0.15 * stats1["candidates_scored"]
        else:
            assert stats["candidates_scored"] == stats1["candidates_scored"]
    assert f[0] == 1 and f[4] == 1 and s[0] == s[4]
    masked = np.where(f != 0, s, -1.0)
    assert best[0] == int(np.argmax(masked)) and np.float32(best[1]) == masked.max()


def test_sharded_match_through_rccl_on_one_device(sm, synth, c2, debug):
    """The RCCL binding of sharded.hip (dlopen, hand-typed prototypes, enum values) EXECUTED: with
    the debug switch comm_force_rccl a communicator of the one device is built by
    ncclCommInitAll(ndev = 1) and the node-wide best key goes through a grouped
    ncclAllReduce(int64, max) on the communicator's stream -- the reduction of one rank's key is
    that key: the call returns what the communicator without RCCL returns, bit for bit."""
    from cartographer_amd import sharding
    _, _, _, truth, scan = c2
    plain = sharding.Communicator([0])
    assert not plain.uses_rccl
    debug(comm_force_rccl=1)
    comm = sharding.Communicator([0])
    assert comm.uses_rccl and comm.num_devices == 1
    matchers = []
    for seed in (42, 43, 44):
        cells, lim, _ = synth.make_submap(seed, 400, 400, 0.05, 30, 1000, 30.0, 0.01)
        matchers.append(sm.FastCorrelativeScanMatcher2D(_grid(sm, cells, lim), 7, 3.0,
                                                        math.radians(20.0), device=0))
    initial = [sm.Rigid2d(truth[0] + 0.2, truth[1] - 0.1, truth[2] + 0.03)] * 3
    args = (matchers, initial, [1, 1, 1], [0.6, 0.6, 0.6], scan)
    for _ in range(2):                       # (the communicator is reusable)
        f, s, p, best, stats = comm.match_batch(*args)
        f0, s0, p0, best0, stats0 = plain.match_batch(*args)
        np.testing.assert_array_equal(f, f0)
        np.testing.assert_array_equal(s, s0)
        assert best == best0 and best[0] == 0 and np.float32(best[1]) == s[0]
    # nothing found anywhere: the reduced key is "not found" (-1) on this path too
    none = comm.match_batch(matchers, initial, [1, 1, 1], [0.999, 0.999, 0.999], scan)
    assert int(np.sum(none[0])) == 0 and none[3][0] == -1
    del comm                                  # ncclCommDestroy


# ----------------------------------------------------------------------------
# Voxel filters and rotational histogram on the device
# ----------------------------------------------------------------------------
@pytest.mark.parametrize("generic", [0, 1])
@pytest.mark.parametrize("seed,n,res", [(0, 2000, 0.3), (1, 20000, 0.05), (2, 500, 1.0),
                                        (3, 5000, 0.011), (4, 1, 0.1), (5, 300000, 0.2),
                                        (6, 4096, 0.2), (7, 4097, 0.2), (8, 63, 50.0),
                                        (9, 3333, 1000.0), (10, 4000, 1e-4)])
def test_voxel_filter_keeps_the_reference_points(oracle, debug, seed, n, res, generic):
    """The randomised reservoir filter, bit for bit: same kept points in the same order -- through
    the one-workgroup path of clouds up to 4096 points (round 6) and through the multi-launch
    path (generic = 1: debug switch filters_generic), every voxel its own point (res 1e-4), all
    points in one voxel (res 1000)."""
    from cartographer_amd import filters
    if generic and n > 4096:
        pytest.skip("the generic path is the only one for this size")
    debug(filters_generic=generic)
    rng = np.random.default_rng(seed)
    cloud = rng.normal(0.0, 3.0, (n, 3)).astype(np.float32)
    k = len(cloud[1::7])
    cloud[::7][:k] = cloud[1::7]
    used = oracle.voxel_filter_flags(cloud, res)
    got = filters.voxel_filter(cloud, res)
    np.testing.assert_array_equal(got, cloud[used])
    # ... and WHICH points: what the overloads over timed points and range measurements select
    # their payload with (the cloud has duplicated positions: the values alone could not tell)
    np.testing.assert_array_equal(filters.voxel_filter_indices(cloud, res), np.nonzero(used)[0])


def test_voxel_filter_reference_tests_on_device():
    from cartographer_amd import filters
    cloud = np.array([[0, 0, 0], [0.1, -0.1, 0.1], [0.3, -0.1, 0], [0, 0, 0.1]], np.float32)
    got = filters.voxel_filter(cloud, 0.3)
    assert len(got) == 2 and any((g == cloud[2]).all() for g in got)
    big = np.array([[100000., 0, 0], [100000.001, -0.0001, 0.0001], [100000.003, -0.0001, 0],
                    [-200000., 0, 0]], np.float32)
    got = filters.voxel_filter(big, 0.01)
    assert len(got) == 2 and any((g == big[3]).all() for g in got)
    assert len(filters.voxel_filter(np.zeros((0, 3), np.float32), 0.1)) == 0
    assert len(filters.voxel_filter_indices(np.zeros((0, 3), np.float32), 0.1)) == 0


@pytest.mark.parametrize("generic", [0, 1])
@pytest.mark.parametrize("seed,n,max_length,min_points,max_range", [
    (0, 20000, 0.5, 200, 50.0), (1, 60000, 2.0, 150, 15.0), (2, 60000, 4.0, 200, 60.0),
    (3, 100, 0.5, 200, 50.0), (4, 3000, 0.9, 2900, 80.0), (5, 1500, 0.5, 200, 50.0),
    (6, 4096, 2.0, 150, 15.0), (7, 900, 0.5, 200, 4.0), (8, 700, 4.0, 650, 60.0),
    (9, 2500, 0.5, 200, 0.01)])
def test_adaptive_voxel_filter_equals_the_oracle(oracle, debug, seed, n, max_length, min_points,
                                                 max_range, generic):
    """AdaptiveVoxelFilter, same points in the same order: clouds of the local trajectory builders'
    sizes through the one-workgroup path (the whole search of voxel_filter.cc:38-75 in one launch,
    round 6) and through the multi-launch path (generic = 1); searches that end at max_length, in
    the halving loop, in the bisection, with the full cloud (nothing dense enough), and with an
    empty range cut."""
    from cartographer_amd import filters
    if generic and n > 4096:
        pytest.skip("the generic path is the only one for this size")
    debug(filters_generic=generic)
    rng = np.random.default_rng(seed)
    cloud = (rng.normal(0.0, 8.0, (n, 3)) * np.array([1.0, 1.0, 0.2])).astype(np.float32)
    ref = oracle.adaptive_voxel_filter(cloud, max_length, min_points, max_range)
    got = filters.adaptive_voxel_filter(cloud, max_length, min_points, max_range)
    np.testing.assert_array_equal(got, ref)
    # ... and WHICH points: ascending indices into the input that select exactly that cloud (a
    # sensor::PointCloud carries its intensities along by them, voxel_filter.cc:138-161)
    kept = filters.adaptive_voxel_filter_indices(cloud, max_length, min_points, max_range)
    assert len(kept) == len(ref) and (np.diff(kept) > 0).all()
    np.testing.assert_array_equal(cloud[kept], ref)


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5])
def test_compute_histogram_equals_the_oracle(oracle, synth, seed):
    """Bit for bit: same slices, same order inside a slice (libm's atan2f arithmetic on the device,
    cmx_atan2f.h, pinned against libm by tests/test_atan2f.py), the `last point` chain walked 64
    points at a time, and every bucket's votes added one by one in the reference's order."""
    from cartographer_amd import filters
    grid, world = synth.make_submap_3d(20 + seed, 0.1, (8.0, 6.0, 3.0), 4, 10, 64)
    pos = world.free_position(seed, 0.5)
    cloud = world.scan(pos, 0.2 * seed, 16, 360, seed=seed)
    ref = oracle.compute_histogram(cloud, 120)
    got = filters.compute_histogram(cloud, 120)
    assert ref.sum() > 0
    np.testing.assert_array_equal(got, ref)


@pytest.mark.parametrize("n,size,spread", [(1, 120, 1.0), (2, 7, 0.3), (63, 1, 2.0), (65, 16, 0.5),
                                           (5000, 120, 4.0), (70000, 120, 6.0), (20000, 8192, 3.0)])
def test_compute_histogram_odd_shapes(oracle, n, size, spread):
    """Slices of one point, slices longer than the kernel's LDS chunk (70 000 points in ~10
    slices), a walk that moves `last` at almost every point (sparse noise) and hardly ever
    (a dense blob), histogram sizes 1 ... 8192: exact."""
    from cartographer_amd import filters
    rng = np.random.default_rng(n + size)
    cloud = (rng.normal(0.0, spread, (n, 3)) * np.array([1.0, 1.0, 0.15])).astype(np.float32)
    ref = oracle.compute_histogram(cloud, size)
    got = filters.compute_histogram(cloud, size)
    np.testing.assert_array_equal(got, ref)


@pytest.mark.parametrize("env", [{"fast2d_fanout": 1},
                                 {"fast2d_fanout": 1, "fast2d_store_scans": 1, "fast2d_xcd_affinity": 1},
                                 {"fanout": True}],
                         ids=["batch", "batch, no stored scans, any XCD", "fan-out (as shipped)"])
def test_c3_share_of_64_submaps_equals_the_single_searches(sm, synth, debug, env):
    """One GPU's share of BASELINE config[2] at its full size: one 1000-point scan against 64
    distinct 400x400 submaps, depth 7, full-submap search.  The batch keeps the cells of the
    scans that can enter the search and places a problem's nodes on one XCD (both only in
    batches); every pair must come back exactly as its single search (which re-derives the cells
    and spreads its nodes) returns it -- found flag, f32 score, pose -- and the work counted by
    the device must be the sum of the singles' lowest-resolution candidates."""
    # (fast2d_fanout = 1: the level-synchronous launches over the whole batch -- since round 6 a
    # batch of 32 and more problems runs as independent single searches over the host pool, the
    # third case; 1 = "never keep the scans' cells" / "nodes on any XCD")
    env = dict(env)
    fanout = env.pop("fanout", False)
    debug(**env)
    matchers, worlds = [], []
    for seed in range(64):
        cells, lim, world = synth.make_submap(300 + seed, 400, 400, 0.05, 30, 1000, 30.0, 0.01)
        matchers.append(sm.FastCorrelativeScanMatcher2D(_grid(sm, cells, lim), 7))
        worlds.append(world)
    scan = worlds[17].scan(worlds[17].free_pose(1234, 0.5), 1000, 30.0, 0.01, 7)
    found, scores, poses, stats = sm.match_full_submap_batch(matchers, scan, 0.6)
    assert found[17] == 1
    if not fanout:
        assert stats["expansion_launches"] == 2 and stats["expansion_lookups"] > 0
    coarse = 0
    for i, m in enumerate(matchers):
        f1, s1, p1 = m.match_full_submap(scan, 0.6)
        coarse += m.last_stats["coarse_candidates"]
        assert bool(found[i]) == bool(f1), i
        if f1:
            assert np.float32(s1) == np.float32(scores[i]), i
            assert (p1.x, p1.y, p1.theta) == tuple(poses[i]), i
    assert stats["coarse_candidates"] == coarse
    if fanout:
        # ... and exactly as the CPU restatement of the reference returns it (the oracle port,
        # itself pinned on the reference's own sources: tests/test_reference_ref.py), every one of
        # the 64 pairs, one host thread each
        from concurrent.futures import ThreadPoolExecutor
        from oracle import pyoracle as orc
        cells_of = [synth.make_submap(300 + seed, 400, 400, 0.05, 30, 1000, 30.0, 0.01)[:2]
                    for seed in range(64)]

        def one(k):
            cells, lim = cells_of[k]
            return orc.FastCorrelativeScanMatcher2D(cells, 0.05, lim["max_x"], lim["max_y"],
                                                    7).match_full_submap(scan, 0.6)
        with ThreadPoolExecutor(min(32, os.cpu_count() or 8)) as pool:
            refs = list(pool.map(one, range(64)))
        for i, r in enumerate(refs):
            assert bool(found[i]) == bool(r["found"]), i
            if r["found"]:
                assert np.float32(r["score"]) == np.float32(scores[i]), i
                np.testing.assert_allclose(poses[i], r["pose"], rtol=0, atol=1e-12, err_msg=str(i))


# ----------------------------------------------------------------------------
# Round 3: scans resident in HBM, the grid's staged image cached with the grid
# ----------------------------------------------------------------------------
def test_rt2d_resident_batch_and_image_cache_follow_the_grid(sm, oracle, synth):
    """cmx_rt2d_match_grid_batch_resident (clouds uploaded once) returns what the host-cloud
    batch returns; the staged image a cmx_grid2d keeps for the row-pair kernel is rebuilt after
    an insertion changed the cells (match -> insert -> match, each equal to the oracle on the
    grid as it is then) and when the window changes."""
    from cartographer_amd import grid_2d
    m = sm.RealTimeCorrelativeScanMatcher2D(0.3, math.radians(7.0), 0.1, 0.1)
    grids, hosts, inits, scans, worlds = [], [], [], [], []

    def insert_scan(k, seed, beams):
        at = worlds[k].free_pose(seed, 0.5)
        pts = worlds[k].scan(at, beams, 5.0, 0.01, seed).astype(np.float64)
        c, s = np.cos(at[2]), np.sin(at[2])
        in_map = np.zeros((pts.shape[0], 3), np.float32)
        in_map[:, 0] = at[0] + c * pts[:, 0] - s * pts[:, 1]
        in_map[:, 1] = at[1] + s * pts[:, 0] + c * pts[:, 1]
        grids[k].insert(at[:2], in_map)
        hosts[k].insert(at[:2], in_map)

    for k in range(6):
        _, lim, world = synth.make_submap(80 + k, 200, 200, 0.05, 20, 600, 5.0, 0.01)
        pose = world.free_pose(500 + k, 0.5)
        scans.append(world.scan(pose, 700 + 13 * k, 5.0, 0.01, k))
        inits.append([pose[0] + 0.1, pose[1] - 0.05, pose[2] + 0.04])
        # the grid grows from the scans inserted into it, on the device and in the host mirror
        grids.append(grid_2d.ProbabilityGridOnDevice(0.05, (lim["max_x"], lim["max_y"]), 200, 200))
        hosts.append(synth.ProbabilityGrid(0.05, (lim["max_x"], lim["max_y"]), 200, 200))
        worlds.append(world)
        for j in range(5):
            insert_scan(k, 600 + 10 * k + j, 400)
        assert grids[k].limits == hosts[k].limits and np.array_equal(grids[k].cells, hosts[k].cells)
    batch = sm.Rt2DBatch(m, grids, scans, resident=True)
    plain = sm.Rt2DBatch(m, grids, scans)

    def check_all():
        s_res, p_res, _ = batch.match(np.array(inits))
        s_res, p_res = s_res.copy(), p_res.copy()
        s_host, p_host, _ = plain.match(np.array(inits))
        np.testing.assert_array_equal(s_res, s_host)
        np.testing.assert_array_equal(p_res, p_host)
        for k in range(len(grids)):
            hl = hosts[k].limits
            ref = oracle.rt2d_match(hosts[k].cells, hl["resolution"], hl["max_x"], hl["max_y"],
                                    inits[k], scans[k], 0.3, math.radians(7.0), 0.1, 0.1)
            assert s_res[k] == ref["score"], k
            np.testing.assert_allclose(p_res[k], ref["pose"], rtol=0, atol=1e-12)

    check_all()
    check_all()                                    # second call: every image comes from its cache
    for k in (1, 4):                               # the cells of two grids change
        insert_scan(k, 900 + k, 300)
        assert np.array_equal(grids[k].cells, hosts[k].cells)
    check_all()
    # another window on the same grids: the cached geometry no longer fits
    m2 = sm.RealTimeCorrelativeScanMatcher2D(0.1, math.radians(5.0), 0.1, 0.1)
    s2, p2, _ = sm.Rt2DBatch(m2, grids, scans, resident=True).match(np.array(inits))
    for k in range(len(grids)):
        hl = hosts[k].limits
        ref = oracle.rt2d_match(hosts[k].cells, hl["resolution"], hl["max_x"], hl["max_y"],
                                inits[k], scans[k], 0.1, math.radians(5.0), 0.1, 0.1)
        assert s2[k] == ref["score"], k
    check_all()
