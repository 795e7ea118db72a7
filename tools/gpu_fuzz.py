#!/usr/bin/env python3
"""Seeded random differential run, DEVICE vs oracle, on small odd-shaped problems (the generators
of tests/test_reference_ref_fuzz.py, where the oracle is compared with the reference's own
sources).  Not part of the pytest suite: run it on the GPU box at the start of a round
(tools/round_start.sh) and turn whatever it finds into a test.
    python tools/gpu_fuzz.py [seconds] [seed]
"""
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from cartographer_amd import scan_matching as sm, scan_matching_3d as sm3      # noqa: E402
from cartographer_amd._lib import VOXEL_DTYPE, CmxError                       # noqa: E402
from oracle import pyoracle as orc                                            # noqa: E402


def quat(rng, max_angle):
    axis = rng.normal(size=3)
    axis /= np.linalg.norm(axis)
    a = rng.uniform(-max_angle, max_angle)
    return [math.cos(a / 2), *(axis * math.sin(a / 2))]


def voxels(rng, n, ext):
    v = np.zeros(n, VOXEL_DTYPE)
    v["x"] = rng.integers(-ext, ext + 1, n)
    v["y"] = rng.integers(-ext, ext + 1, n)
    v["z"] = rng.integers(-ext // 2, ext // 2 + 1, n)
    v["value"] = rng.integers(1, 32768, n)
    _, first = np.unique(np.stack([v["x"], v["y"], v["z"]], 1), axis=0, return_index=True)
    return v[np.sort(first)]


def case_2d(rng, report):
    nx, ny = int(rng.integers(3, 70)), int(rng.integers(3, 70))
    res = float(rng.choice([0.05, 0.1, 0.025, 0.2]))
    kind = int(rng.integers(0, 3))
    cells = np.zeros((ny, nx), np.uint16)
    if kind == 0:
        m = rng.uniform(size=cells.shape) < rng.uniform(0.02, 0.6)
        cells[m] = rng.integers(1, 32768, m.sum())
    elif kind == 1:
        m = rng.uniform(size=cells.shape) < 0.5
        cells[m] = rng.choice([3000, 16000, 30000], m.sum())
    else:
        m = rng.uniform(size=cells.shape) < 0.3
        cells[m] = rng.integers(1, 65536, m.sum())
    max_x, max_y = float(rng.uniform(-5, 5)), float(rng.uniform(-5, 5))
    n = int(rng.integers(1, 40))
    ext = max(nx, ny) * res
    pts = np.zeros((n, 3), np.float32)
    pts[:, :2] = rng.uniform(-ext * 0.7, ext * 0.7, (n, 2))
    depth = int(rng.integers(1, 8))
    lin, ang = float(rng.uniform(0.0, 0.6 * ext)), float(rng.uniform(0.0, 1.0))
    init = [max_x - rng.uniform(0, ny * res), max_y - rng.uniform(0, nx * res),
            float(rng.uniform(-3.2, 3.2))]
    min_score = float(rng.choice([0.05, 0.2, 0.5, 0.95]))
    what = dict(nx=nx, ny=ny, res=res, kind=kind, n=n, depth=depth, lin=lin, ang=ang,
                min_score=min_score)
    om = orc.FastCorrelativeScanMatcher2D(cells, res, max_x, max_y, depth, lin, ang)
    gm = sm.FastCorrelativeScanMatcher2D(sm.Grid2D(cells, res, max_x, max_y), depth, lin, ang)
    for full in (False, True):
        if full and nx * ny > 1600:
            continue
        a = om.match_full_submap(pts, min_score) if full else om.match(init, pts, min_score)
        found, score, pose = (gm.match_full_submap(pts, min_score) if full
                              else gm.match(sm.Rigid2d(*init), pts, min_score))
        ok = bool(found) == a["found"] and (not found or (
            np.float32(score) == np.float32(a["score"]) and
            np.allclose([pose.x, pose.y, pose.theta], a["pose"], rtol=0, atol=1e-12)))
        if not ok:
            report("fast2d", dict(what, full=full), (found, score), (a["found"], a.get("score")))
        # the group bounds of the front end (round 6) at ANY depth > 1, every bound checked against
        # the exact sums on the device (a violation fails the call); then with every unit's outer
        # rotations unbounded, and with exact lowest-resolution scores
        from cartographer_amd import _lib as lib
        for name, switches in (("fast2d group bounds", dict(fast2d_group=2, fast2d_group_verify=1)),
                               ("fast2d group bounds, premise failed", dict(fast2d_group=2, fast2d_group_verify=3)),
                               ("fast2d exact", dict(fast2d_group=1))):
            lib.debug_set(**switches)
            try:
                found, score, pose = (gm.match_full_submap(pts, min_score) if full
                                      else gm.match(sm.Rigid2d(*init), pts, min_score))
            finally:
                lib.debug_set(**{k: 0 for k in switches})
            ok = bool(found) == a["found"] and (not found or (
                np.float32(score) == np.float32(a["score"]) and
                np.allclose([pose.x, pose.y, pose.theta], a["pose"], rtol=0, atol=1e-12)))
            if not ok:
                report(name, dict(what, full=full), (found, score), (a["found"], a.get("score")))
    rt = (float(rng.uniform(0, 6 * res)), float(rng.uniform(0, 0.3)),
          float(rng.choice([0, 0.1, 10])), float(rng.choice([0, 0.5, 3])))
    a = orc.rt2d_match(cells, res, max_x, max_y, init, pts, *rt)
    m = sm.RealTimeCorrelativeScanMatcher2D(*rt)
    score, pose = m.match(sm.Rigid2d(*init), pts, sm.Grid2D(cells, res, max_x, max_y))
    if not (score == a["score"] and
            np.allclose([pose.x, pose.y, pose.theta], a["pose"], rtol=0, atol=1e-12)):
        report("rt2d", dict(what, rt=rt), score, a["score"])
    # the same match through the block bounds (the default only from 96 matches per call on): 4 x 4
    # blocks + tail kernel (round 6), 2 x 2 blocks + tail kernel, 2 x 2 blocks in one kernel
    from cartographer_amd import _lib
    for name, switches in (("rt2d bounds", {}), ("rt2d bounds level 2", dict(rt2d_bounds_level=2)),
                           ("rt2d bounds fused", dict(rt2d_bounds_fused=1))):
        _lib.debug_set(rt2d_bounds=1, **switches)
        try:
            score, pose = m.match(sm.Rigid2d(*init), pts, sm.Grid2D(cells, res, max_x, max_y))
        finally:
            _lib.debug_set(rt2d_bounds=0, **{k: 0 for k in switches})
        if not (score == a["score"] and
                np.allclose([pose.x, pose.y, pose.theta], a["pose"], rtol=0, atol=1e-12)):
            report(name, dict(what, rt=rt), score, a["score"])


def case_3d(rng, report):
    res = float(rng.choice([0.05, 0.1, 0.2, 0.45]))
    ext = int(rng.integers(4, 30))
    vox = voxels(rng, int(rng.integers(1, 2000)), ext)
    low_res = float(rng.choice([res, 2 * res, 0.45]))
    low = voxels(rng, int(rng.integers(1, 500)), max(2, int(ext * res / low_res)))
    n = int(rng.integers(1, 60))
    cloud = rng.uniform(-ext * res, ext * res, (n, 3)).astype(np.float32)
    lo_cloud = cloud[:: int(rng.integers(1, 5))].copy()
    init = list(rng.uniform(-0.5, 0.5, 3)) + quat(rng, 0.6)
    rt = (float(rng.uniform(0, 1.6 * res)), float(rng.uniform(0, 0.03)),
          float(rng.choice([0, 0.1, 5])), float(rng.choice([0, 0.1, 5])))
    a = orc.rt3d_match(res, vox, init, cloud, *rt)
    m = sm3.RealTimeCorrelativeScanMatcher3D(*rt)
    score, pose = m.match(sm3.Rigid3d(tuple(init[:3]), tuple(init[3:])), cloud, res, vox)
    if not (np.float32(score) == np.float32(a["score"]) and
            np.array_equal(list(pose.translation) + list(pose.rotation), a["pose"])):
        report("rt3d", dict(res=res, ext=ext, n=n, rt=rt), score, a["score"])
    depth, frd = int(rng.integers(1, 7)), int(rng.integers(1, 8))
    hs = int(rng.choice([1, 8, 30, 120]))
    hist = rng.uniform(0, 2, hs).astype(np.float32) * (rng.uniform() < 0.8)
    scan_hist = rng.uniform(0, 2, hs).astype(np.float32) * (rng.uniform() < 0.8)
    opt = dict(branch_and_bound_depth=depth, full_resolution_depth=frd,
               min_rotational_score=float(rng.choice([0.0, 0.3, 0.7])),
               min_low_resolution_score=float(rng.choice([0.0, 0.12, 0.3])),
               linear_xy_search_window=float(rng.uniform(0, 8 * res)),
               linear_z_search_window=float(rng.uniform(0, 4 * res)),
               angular_search_window=float(rng.uniform(0, 0.5)))
    om = orc.FastCorrelativeScanMatcher3D(res, vox, low_res, low, hist, depth, frd,
                                          opt["min_rotational_score"],
                                          opt["min_low_resolution_score"],
                                          opt["linear_xy_search_window"],
                                          opt["linear_z_search_window"],
                                          opt["angular_search_window"])
    grid_size = orc.grid3d_size(res, vox)
    gm = sm3.FastCorrelativeScanMatcher3D(res, vox, grid_size, low_res, low, hist, **opt)
    for d in range(depth):
        if not np.array_equal(gm.level(d), om.level(d)):
            report("fast3d level", dict(res=res, ext=ext, depth=depth, frd=frd, level=d), None, None)
            return
    node = list(rng.uniform(-1, 1, 3)) + quat(rng, 3.0)
    sub = list(rng.uniform(-1, 1, 3)) + quat(rng, 3.0)
    grav = quat(rng, 0.1)
    ms = float(rng.choice([0.05, 0.12, 0.3]))
    a = om.match(node, sub, grav, cloud, lo_cloud, scan_hist, ms)
    got = gm.match(sm3.Rigid3d(tuple(node[:3]), tuple(node[3:])),
                   sm3.Rigid3d(tuple(sub[:3]), tuple(sub[3:])),
                   sm3.TrajectoryNodeData(cloud, lo_cloud, scan_hist, tuple(grav)), ms)
    ok = (got is not None) == a["found"]
    if ok and got is not None:
        p = got["pose_estimate"]
        ok = (all(np.float32(got[k]) == np.float32(a[k])
                  for k in ("score", "rotational_score", "low_resolution_score")) and
              np.array_equal(list(p.translation) + list(p.rotation), a["pose"]))
    if not ok:
        report("fast3d", dict(res=res, ext=ext, n=n, hs=hs, min_score=ms, **opt),
               None if got is None else got["score"], (a["found"], a.get("score")))


def case_grids(rng, report):
    """Random range data into the device-resident grids (2D with a final crop, 3D) against the
    host builders (which are pinned on the reference's own inserters)."""
    from cartographer_amd import grid_2d, grid_3d, synth
    res = float(rng.choice([0.05, 0.1, 0.25, 1.0]))
    nx, ny = int(rng.integers(1, 12)), int(rng.integers(1, 12))
    corner = (float(rng.uniform(-3, 3)), float(rng.uniform(-3, 3)))
    host = synth.ProbabilityGrid(res, corner, nx, ny)
    dev = grid_2d.ProbabilityGridOnDevice(res, corner, nx, ny)
    for _ in range(6):
        origin = [corner[0] - rng.uniform(-2, 6) * res * 3, corner[1] - rng.uniform(-2, 6) * res * 3]
        n = int(rng.integers(0, 40))
        ang = rng.uniform(0, 2 * math.pi, n)
        rad = rng.uniform(0, 40 * res, n) * (rng.uniform(size=n) > 0.1)
        pts = np.zeros((n, 3), np.float32)
        pts[:, 0] = origin[0] + rad * np.cos(ang)
        pts[:, 1] = origin[1] + rad * np.sin(ang)
        split = int(rng.integers(0, n + 1))
        hit, miss = float(rng.uniform(0.51, 0.95)), float(rng.uniform(0.05, 0.49))
        free = bool(rng.integers(0, 2))
        host.insert(origin, pts[:split], pts[split:], hit, miss, free)
        dev.insert(origin, pts[:split], pts[split:], hit, miss, free)
        if host.limits != dev.limits or not np.array_equal(host.cells, dev.cells):
            report("grid2d insert", dict(res=res, nx=nx, ny=ny, n=n), dev.limits, host.limits)
            return
    cropped = host.cropped()
    dev.crop()
    if cropped.limits != dev.limits or not np.array_equal(cropped.cells, dev.cells):
        report("grid2d crop", dict(res=res, nx=nx, ny=ny), dev.limits, cropped.limits)
    res3 = float(rng.choice([0.05, 0.1, 0.45, 1.0]))
    h3, d3 = synth.HybridGrid(res3), grid_3d.HybridGridOnDevice(res3)
    for _ in range(5):
        origin = rng.uniform(-20, 20, 3).astype(np.float32) * np.float32(res3)
        n = int(rng.integers(0, 60))
        d = rng.normal(size=(n, 3))
        d /= np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-9)
        rad = rng.uniform(0, 90 * res3, (n, 1)) * (rng.uniform(size=(n, 1)) > 0.1)
        pts = (origin + d * rad).astype(np.float32)
        hit, miss = float(rng.uniform(0.51, 0.95)), float(rng.uniform(0.05, 0.49))
        free = int(rng.choice([0, 1, 2, 10, 60]))
        h3.insert(origin, pts, hit, miss, free)
        d3.insert(origin, pts, hit, miss, free)
        if h3.grid_size != d3.grid_size or not np.array_equal(h3.voxels(), d3.voxels()):
            report("grid3d insert", dict(res=res3, n=n, free=free), d3.grid_size, h3.grid_size)
            return


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    bad = []

    def report(kind, what, got, want):
        bad.append(kind)
        print(f"MISMATCH {kind}: {what} device={got} oracle={want}", flush=True)

    t0, cases = time.time(), 0
    while time.time() - t0 < seconds:
        state = rng.bit_generator.state
        try:
            (case_2d, case_3d, case_grids)[cases % 3](rng, report)
        except CmxError as exc:            # an input the C ABI rejects is reported, not fatal
            print(f"REJECTED (case {cases}): {exc}", flush=True)
        except Exception as exc:           # noqa: BLE001
            bad.append("exception")
            print(f"EXCEPTION (case {cases}, rng state {state['state']}): {exc!r}", flush=True)
        cases += 1
    print(f"gpu_fuzz: {cases} cases, {len(bad)} mismatches, seed {seed}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
