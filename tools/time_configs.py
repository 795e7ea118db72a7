#!/usr/bin/env python3
"""Wall-clock + device timings of the BASELINE.json configs other than the bench line
(C1 RT-2D, C4 RT-3D, C5 fast-3D per submap).  Usage: python tools/time_configs.py [c1] [c4] [c5]
"""
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from cartographer_amd import scan_matching as sm, scan_matching_3d as sm3, synth  # noqa: E402
from cartographer_amd import _lib as _cmx_lib  # noqa: E402
_cmx_lib.debug_set(timing=1)   # cmx_match_stats *_ms are recorded only on request


def timeit(fn, reps, warm=2):
    for _ in range(warm):
        fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    return (time.perf_counter() - t0) / reps, out


def c1(cpu):
    cells, lim, world = synth.make_submap(42, 200, 200, 0.05, 30, 1000, 5.0, 0.01)
    pose = world.free_pose(1234, 0.5)
    scan = world.scan(pose, 1000, 5.0, 0.01, 7)
    grid = sm.Grid2D(cells, lim["resolution"], lim["max_x"], lim["max_y"])
    m = sm.RealTimeCorrelativeScanMatcher2D(0.3, math.radians(7.0), 0.1, 0.1)
    init = sm.Rigid2d(pose[0] + 0.12, pose[1] - 0.08, pose[2] + math.radians(3.0))
    dt, (score, est) = timeit(lambda: m.match(init, scan, grid), 50)
    st = m.last_stats
    print(f"C1 rt2d: {dt * 1e6:.1f} us/match wall, device {st['device_ms'] * 1e3:.1f} us, kernel "
          f"{st['dominant_kernel_ms'] * 1e3:.1f} us, {st['candidates_scored']} cand, N={len(scan)} "
          f"-> {st['candidates_scored'] / dt:.3e} cand/s; score {score:.4f}")
    from cartographer_amd import grid_2d
    dev = grid_2d.ProbabilityGridOnDevice(0.05, (lim["max_x"], lim["max_y"]), 200, 200, cells=cells)
    dt2, (score2, _) = timeit(lambda: m.match(init, scan, dev), 50)
    st2 = m.last_stats
    print(f"   grid resident in HBM: {dt2 * 1e6:.1f} us/match wall, device "
          f"{st2['device_ms'] * 1e3:.1f} us; score equal: {score2 == score}")
    c, s_ = math.cos(pose[2]), math.sin(pose[2])
    in_map = np.zeros((scan.shape[0], 3), np.float32)
    in_map[:, 0] = pose[0] + c * scan[:, 0] - s_ * scan[:, 1]
    in_map[:, 1] = pose[1] + s_ * scan[:, 0] + c * scan[:, 1]
    t_ins, _ = timeit(lambda: dev.insert(pose[:2], in_map), 20)
    print(f"   range-data insertion on the device: {t_ins * 1e6:.1f} us/scan wall "
          f"({scan.shape[0]} rays)")
    if cpu:
        from oracle import pyoracle as orc
        t0 = time.perf_counter()
        ref = orc.rt2d_match(cells, lim["resolution"], lim["max_x"], lim["max_y"],
                             [init.x, init.y, init.theta], scan, 0.3, math.radians(7.0), 0.1, 0.1)
        t = time.perf_counter() - t0
        print(f"   oracle 1 thread: {t * 1e3:.2f} ms -> {st['candidates_scored'] / t:.3e} cand/s; "
              f"score equal: {np.float64(ref['score']) == score}")
        if orc.ref_lib() is not None:        # the reference's own source, oracle/_ref
            t0 = time.perf_counter()
            rr = orc.ref_rt2d_match(cells, lim["resolution"], lim["max_x"], lim["max_y"],
                                    [init.x, init.y, init.theta], scan, 0.3, math.radians(7.0),
                                    0.1, 0.1)
            t = time.perf_counter() - t0
            print(f"   reference's own real_time_correlative_scan_matcher_2d.cc, 1 thread: "
                  f"{t * 1e3:.2f} ms -> {st['candidates_scored'] / t:.3e} cand/s; score equal: "
                  f"{rr['score'] == score}")


def c1b(cpu):
    """C1 as a batch: B robots, each matching its own scan against its own resident grid."""
    from cartographer_amd import grid_2d
    m = sm.RealTimeCorrelativeScanMatcher2D(0.3, math.radians(7.0), 0.1, 0.1)
    for batch in (1, 8, 32, 128):
        grids, inits, scans = [], [], []
        for k in range(min(batch, 8)):          # 8 distinct worlds, reused round-robin
            cells, lim, world = synth.make_submap(42 + k, 200, 200, 0.05, 30, 1000, 5.0, 0.01)
            pose = world.free_pose(1234, 0.5)
            grids.append(grid_2d.ProbabilityGridOnDevice(0.05, (lim["max_x"], lim["max_y"]), 200,
                                                         200, cells=cells))
            scans.append(world.scan(pose, 1000, 5.0, 0.01, 7))
            inits.append(sm.Rigid2d(pose[0] + 0.12, pose[1] - 0.08, pose[2] + math.radians(3.0)))
        G = [grids[i % len(grids)] for i in range(batch)]
        I = [inits[i % len(grids)] for i in range(batch)]
        S = [scans[i % len(grids)] for i in range(batch)]
        init = np.array([[p.x, p.y, p.theta] for p in I])
        for name, b in (("host clouds", sm.Rt2DBatch(m, G, S)),
                        ("resident   ", sm.Rt2DBatch(m, G, S, resident=True))):
            dt, (scores, poses, st) = timeit(lambda: b.match(init), 30, warm=3)
            print(f"C1 batch {batch:4d} {name}: {dt * 1e6:8.1f} us / batch wall, device "
                  f"{st['device_ms'] * 1e3:7.1f} us, kernel {st['dominant_kernel_ms'] * 1e3:7.1f} us, "
                  f"{st['candidates_scored']} cand -> {st['candidates_scored'] / dt:.3e} cand/s wall, "
                  f"{st['candidates_scored'] / (st['dominant_kernel_ms'] * 1e-3):.3e} cand/s kernel")


def c4(cpu, rings=64, az=1024):
    grid, world = synth.make_submap_3d(42, 0.1, (15.0, 15.0, 7.5), 8, 32, 512)
    vox = grid.voxels()
    pos = world.free_position(77, 0.5)
    cloud = world.scan(pos, 0.3, rings, az, seed=9)
    c, s = math.cos(0.31 / 2), math.sin(0.31 / 2)
    init = sm3.Rigid3d(tuple(pos + np.array([0.07, -0.04, 0.02])), (c, 0.0, 0.0, s))
    m = sm3.RealTimeCorrelativeScanMatcher3D(0.5, math.radians(2.0), 0.1, 0.1)
    dt, (score, est) = timeit(lambda: m.match(init, cloud, 0.1, vox), 3, warm=1)
    st = m.last_stats
    print(f"C4 rt3d: {dt * 1e3:.2f} ms/match wall, device {st['device_ms']:.2f} ms, kernel "
          f"{st['dominant_kernel_ms']:.2f} ms, {st['candidates_scored']} cand, N={len(cloud)}, "
          f"voxels {len(vox)} -> {st['candidates_scored'] / dt:.3e} cand/s, "
          f"{st['candidates_scored'] * len(cloud) / dt:.3e} lookups/s; score {score:.4f}")


def c5(cpu):
    """One submap of C5: hi 0.1 m / low 0.45 m, depth 8 / frd 3, pose_graph.lua windows."""
    size = (15.0, 15.0, 7.5)
    grid, world = synth.make_submap_3d(42, 0.1, size, 8, 32, 512)
    low, _ = synth.make_submap_3d(42, 0.45, size, 8, 32, 512)
    vox, low_vox = grid.voxels(), low.voxels()
    rng = np.random.default_rng(1)
    hist = rng.uniform(0.0, 1.0, 120).astype(np.float32)
    hist[10:14] += 6.0
    pos = world.free_position(77, 0.6)
    yaw = 0.4
    full = world.scan(pos, yaw, 32, 512, seed=1)
    hi = full[::6].copy()          # ~2.7 k points after "voxel filtering"
    lo = full[::80].copy()         # ~200 points
    scan_hist = np.roll(hist, -19).copy()   # node yaw in the submap frame ~0.5 rad = 19 buckets
    opt = dict(branch_and_bound_depth=8, full_resolution_depth=3, min_rotational_score=0.77,
               min_low_resolution_score=0.35, linear_xy_search_window=5.0,
               linear_z_search_window=1.0, angular_search_window=math.radians(15.0))
    t0 = time.perf_counter()
    gm = sm3.FastCorrelativeScanMatcher3D(0.1, vox, grid.grid_size, 0.45, low_vox, hist, **opt)
    t_create = time.perf_counter() - t0
    node = sm3.Rigid3d((pos[0] + 0.8, pos[1] - 0.6, pos[2] + 0.2),
                       (math.cos((yaw + 0.1) / 2), 0.0, 0.0, math.sin((yaw + 0.1) / 2)))
    data = sm3.TrajectoryNodeData(hi, lo, scan_hist)
    dt, got = timeit(lambda: gm.match(node, sm3.Rigid3d(), data, 0.2), 5, warm=1)
    st = gm.last_stats
    print(f"C5 fast3d (1 submap): create {t_create * 1e3:.1f} ms; match {dt * 1e3:.2f} ms wall, "
          f"device {st['device_ms']:.2f} ms, kernel {st['dominant_kernel_ms']:.3f} ms, "
          f"{st['candidates_scored']} cand ({st['coarse_candidates']} coarse), "
          f"{st['num_scans']} yaws, {st['nodes_expanded']} nodes, hi N={len(hi)} lo N={len(lo)} "
          f"voxels {len(vox)}/{len(low_vox)} -> {st['candidates_scored'] / dt:.3e} cand/s; "
          f"found {got is not None} score {got['score'] if got else None}")
    if cpu:
        from oracle import pyoracle as orc
        om = orc.FastCorrelativeScanMatcher3D(0.1, vox, 0.45, low_vox, hist, 8, 3, 0.77, 0.35, 5.0,
                                              1.0, math.radians(15.0))
        t0 = time.perf_counter()
        ref = om.match(list(node.translation) + list(node.rotation), [0, 0, 0, 1, 0, 0, 0],
                       [1, 0, 0, 0], hi, lo, scan_hist, 0.2)
        t = time.perf_counter() - t0
        print(f"   oracle 1 thread: {t * 1e3:.1f} ms; found {ref['found']} score "
              f"{ref.get('score')} (gpu equal: "
              f"{got is not None and np.float32(got['score']) == np.float32(ref['score'])})")
        if orc.ref_lib() is not None:        # the reference's own source, oracle/_ref
            rm = orc.ReferenceFastCorrelativeScanMatcher3D(0.1, vox, 0.45, low_vox, hist, 8, 3,
                                                           0.77, 0.35, 5.0, 1.0,
                                                           math.radians(15.0))
            t0 = time.perf_counter()
            rr = rm.match(list(node.translation) + list(node.rotation), [0, 0, 0, 1, 0, 0, 0],
                          [1, 0, 0, 0], hi, lo, scan_hist, 0.2)
            t = time.perf_counter() - t0
            print(f"   reference's own fast_correlative_scan_matcher_3d.cc, 1 thread: "
                  f"{t * 1e3:.1f} ms; found {rr['found']} score {rr.get('score')} (gpu equal: "
                  f"{got is not None and np.float32(got['score']) == np.float32(rr['score'])})")


if __name__ == "__main__":
    which = [a for a in sys.argv[1:] if not a.startswith("-")] or ["c1", "c4", "c5"]
    cpu = "--cpu" in sys.argv
    for w in which:
        {"c1": c1, "c1b": c1b, "c4": c4, "c5": c5}[w](cpu)
