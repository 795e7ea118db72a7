#!/usr/bin/env python3
"""A few calls of every kernel family that bench.py does not time -- grid inserters (2D, 3D),
Ceres refinement, voxel filters, rotational histogram -- so that rocprofv3 sees them
(tools/profile_all.sh).  Prints wall times."""
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cartographer_amd import filters, grid_2d, grid_3d, scan_matching as sm, synth  # noqa: E402


def timed(name, fn, reps=5):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    print(f"{name}: {(time.perf_counter() - t0) / reps * 1e6:.1f} us / call")
    return out


cells, lim, world = synth.make_submap(42, 200, 200, 0.05, 30, 1000, 5.0, 0.01)
pose = world.free_pose(1234, 0.5)
scan = world.scan(pose, 1000, 5.0, 0.01, 7)
c, s_ = math.cos(pose[2]), math.sin(pose[2])
in_map = np.zeros((scan.shape[0], 3), np.float32)
in_map[:, 0] = pose[0] + c * scan[:, 0] - s_ * scan[:, 1]
in_map[:, 1] = pose[1] + s_ * scan[:, 0] + c * scan[:, 1]
dev = grid_2d.ProbabilityGridOnDevice(0.05, (lim["max_x"], lim["max_y"]), 200, 200, cells=cells)
timed("grid2d insert (891 rays)", lambda: dev.insert(pose[:2], in_map))
ceres = sm.CeresScanMatcher2D(1.0, 10.0, 40.0, False, 20)
init = sm.Rigid2d(pose[0] + 0.03, pose[1] - 0.02, pose[2] + 0.01)
timed("ceres2d match on the resident grid", lambda: ceres.match((init.x, init.y), init, scan, dev))

g3 = grid_3d.HybridGridOnDevice(0.1)
grid, world3 = synth.make_submap_3d(42, 0.1, (15.0, 15.0, 7.5), 8, 32, 512)
pos = world3.free_position(77, 0.5)
cloud = world3.scan(pos, 0.3, 32, 512, seed=9)
in_map3 = (cloud + pos).astype(np.float32)
timed("grid3d insert (16k rays)", lambda: g3.insert(pos, in_map3), reps=3)
raw = world3.scan(pos, 0.3, 64, 1024, seed=3)
timed("voxel filter 65k points, 0.15 m", lambda: filters.voxel_filter(raw, 0.15))
timed("adaptive voxel filter 65k points", lambda: filters.adaptive_voxel_filter(raw, 2.0, 150, 15.0))
timed("rotational histogram 65k points", lambda: filters.compute_histogram(raw, 120))
