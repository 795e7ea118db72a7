#!/usr/bin/env python3
"""A/B of the two fused front ends of the fast 2D matcher on one box: PrepScoreLdsKernel
(phase planes staged in LDS, CMX_LDS_FRONT=1) against PrepScoreFusedKernel (=0).

  parity : every introspection array (discretised scans, bounds, lowest-resolution sums) and the
           match result of a few shapes (w = 64 / 32 / 16, 48-cell planes, n = 1 .. 1024) equal;
  timing : C2 single-stream latency and front-end time for both, 8-thread throughput, the
           16-submap batch.
Usage: python tools/lds_front_probe.py [parity] [time] [threads] [batch]
"""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from cartographer_amd import scan_matching as sm, synth  # noqa: E402


def make(seed, nx, ny, res, depth, beams, max_range=30.0):
    cells, lim, world = synth.make_submap(seed, nx, ny, res, 30, 1000, max_range, 0.01)
    pose = world.free_pose(1234, 0.5)
    scan = world.scan(pose, beams, max_range, 0.01, 7)
    g = sm.Grid2D(cells, lim["resolution"], lim["max_x"], lim["max_y"])
    return sm.FastCorrelativeScanMatcher2D(g, depth), scan


def parity():
    os.environ["CMX_LDS_FRONT_MIN_SCANS"] = "64"
    bad = 0
    shapes = [(400, 400, 0.05, 7, 1000), (400, 300, 0.05, 7, 700), (200, 200, 0.05, 6, 300),
              (100, 120, 0.05, 5, 100), (400, 400, 0.05, 7, 1024), (200, 200, 0.05, 6, 1),
              (400, 400, 0.05, 7, 257)]
    for nx, ny, res, depth, beams in shapes:
        m, scan = make(42, nx, ny, res, depth, beams)
        out = {}
        for flag in ("1", "0"):
            os.environ["CMX_LDS_FRONT"] = flag
            d = m.debug_prepare(None, scan, True)
            r = m.match_full_submap(scan, 0.3)
            out[flag] = (d, r, dict(m.last_stats))
        a, b = out["1"], out["0"]
        same = (a[0]["num_scans"] == b[0]["num_scans"]
                and np.array_equal(a[0]["scans"], b[0]["scans"])
                and np.array_equal(a[0]["bounds"], b[0]["bounds"])
                and np.array_equal(a[0]["sums"], b[0]["sums"]))
        ra, rb = a[1], b[1]
        same_match = ra[0] == rb[0] and (not ra[0] or (
            np.float32(ra[1]) == np.float32(rb[1])
            and (ra[2].x, ra[2].y, ra[2].theta) == (rb[2].x, rb[2].y, rb[2].theta)))
        same_stats = a[2]["coarse_candidates"] == b[2]["coarse_candidates"]
        print(f"parity {nx}x{ny} depth {depth} n={len(scan)} scans={a[0]['num_scans']}: arrays "
              f"{same} match {same_match} coarse {same_stats} (found {ra[0]}, score {ra[1]})")
        if not same:
            for k in ("scans", "bounds", "sums"):
                x, y = a[0][k], b[0][k]
                if x.shape != y.shape:
                    print("   ", k, "shapes", x.shape, y.shape)
                else:
                    diff = np.flatnonzero(x.ravel() != y.ravel())
                    print("   ", k, "differs at", diff.size, "of", x.size, diff[:8],
                          x.ravel()[diff[:8]], y.ravel()[diff[:8]])
        bad += 0 if (same and same_match and same_stats) else 1
    del os.environ["CMX_LDS_FRONT_MIN_SCANS"]
    print("PARITY", "OK" if bad == 0 else f"FAILED ({bad})")
    return bad == 0


def timing():
    m, scan = make(42, 400, 400, 0.05, 7, 1000)
    cloud = sm.PointCloudOnDevice(scan)
    for flag in ("1", "0", "1", "0"):
        os.environ["CMX_LDS_FRONT"] = flag
        for _ in range(20):
            sm.match_full_submap_batch([m], cloud, 0.6)
        t0 = time.perf_counter()
        dev = ker = 0.0
        reps = 200
        for _ in range(reps):
            st = sm.match_full_submap_batch([m], cloud, 0.6)[3]
            dev += st["device_ms"]
            ker += st["dominant_kernel_ms"]
        dt = (time.perf_counter() - t0) / reps
        print(f"C2 single LDS_FRONT={flag}: wall {dt * 1e6:.1f} us, device {dev / reps * 1e3:.1f} us, "
              f"front end {ker / reps * 1e3:.1f} us, cand {st['candidates_scored']}")
    for waves in ("6", "9", "12"):
        os.environ["CMX_LDS_FRONT"] = "1"
        os.environ["CMX_LDS_FRONT_WAVES"] = waves
        for _ in range(10):
            sm.match_full_submap_batch([m], cloud, 0.6)
        ker = 0.0
        for _ in range(100):
            ker += sm.match_full_submap_batch([m], cloud, 0.6)[3]["dominant_kernel_ms"]
        print(f"   waves {waves}: front end {ker / 100 * 1e3:.1f} us")
    del os.environ["CMX_LDS_FRONT_WAVES"]


def threads():
    ms = [make(42, 400, 400, 0.05, 7, 1000) for _ in range(1)]
    m, scan = ms[0]
    cloud = sm.PointCloudOnDevice(scan)
    for flag in ("1", "0"):
        os.environ["CMX_LDS_FRONT"] = flag
        for nthreads in (8, 16):
            per = 300
            count = [0]

            def work():
                for _ in range(per):
                    sm.match_full_submap_batch([m], cloud, 0.6)
            for _ in range(10):
                sm.match_full_submap_batch([m], cloud, 0.6)
            ts = [threading.Thread(target=work) for _ in range(nthreads)]
            t0 = time.perf_counter()
            for t in ts:
                t.start()
            for t in ts:
                t.join()
            dt = time.perf_counter() - t0
            print(f"C2 {nthreads} threads LDS_FRONT={flag}: {dt / (per * nthreads) * 1e6:.1f} us / match")


def batch():
    matchers = []
    for seed in range(42, 58):
        cells, lim, world = synth.make_submap(seed, 400, 400, 0.05, 30, 1000, 30.0, 0.01)
        if seed == 42:
            pose = world.free_pose(1234, 0.5)
            scan = world.scan(pose, 1000, 30.0, 0.01, 7)
        matchers.append(sm.FastCorrelativeScanMatcher2D(
            sm.Grid2D(cells, lim["resolution"], lim["max_x"], lim["max_y"]), 7))
    cloud = sm.PointCloudOnDevice(scan)
    res = {}
    for flag in ("1", "0"):
        os.environ["CMX_LDS_FRONT"] = flag
        for _ in range(3):
            out = sm.match_full_submap_batch(matchers, cloud, 0.6)
        t0 = time.perf_counter()
        reps = 20
        ker = 0.0
        for _ in range(reps):
            out = sm.match_full_submap_batch(matchers, cloud, 0.6)
            ker += out[3]["dominant_kernel_ms"]
        dt = (time.perf_counter() - t0) / reps
        res[flag] = out
        print(f"C3 16-share LDS_FRONT={flag}: {dt * 1e3:.3f} ms wall, device {out[3]['device_ms']:.3f} ms, "
              f"front end {ker / reps:.3f} ms")
    a, b = res["1"], res["0"]
    print("batch equal:", np.array_equal(a[0], b[0]), np.array_equal(a[1], b[1]),
          np.array_equal(a[2], b[2]))


def front_ms(m, cloud, reps=100):
    for _ in range(10):
        sm.match_full_submap_batch([m], cloud, 0.6)
    ker = 0.0
    for _ in range(reps):
        ker += sm.match_full_submap_batch([m], cloud, 0.6)[3]["dominant_kernel_ms"]
    return ker / reps * 1e3


def tl():
    """CMX_TIMELINE=1 python tools/lds_front_probe.py tl: the in-kernel timeline of three searches."""
    m, scan = make(42, 400, 400, 0.05, 7, 1000)
    cloud = sm.PointCloudOnDevice(scan)
    os.environ["CMX_LDS_FRONT"] = "1"
    for _ in range(3):
        sm.match_full_submap_batch([m], cloud, 0.6)


def dbg():
    """Front-end time with parts of the kernel switched off (CMX_LDS_FRONT_DEBUG; wrong results)."""
    m, scan = make(42, 400, 400, 0.05, 7, 1000)
    cloud = sm.PointCloudOnDevice(scan)
    os.environ["CMX_LDS_FRONT"] = "1"
    for mode, what in (("0", "complete"), ("1", "no flushes"), ("2", "no gathers"), ("0", "complete")):
        os.environ["CMX_LDS_FRONT_DEBUG"] = mode
        print(f"   debug {mode} ({what}): front end {front_ms(m, cloud):.1f} us")
    del os.environ["CMX_LDS_FRONT_DEBUG"]


if __name__ == "__main__":
    if "tl" in sys.argv[1:]:
        tl()
        sys.exit(0)
    if "dbg" in sys.argv[1:]:
        dbg()
        sys.exit(0)
    which = sys.argv[1:] or ["parity", "time", "threads", "batch"]
    ok = True
    if "parity" in which:
        ok = parity()
    if ok and "time" in which:
        timing()
    if ok and "threads" in which:
        threads()
    if ok and "batch" in which:
        batch()
    sys.exit(0 if ok else 1)
