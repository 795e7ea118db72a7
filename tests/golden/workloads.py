"""The inputs of the golden reference results (tests/golden/reference_results.json): seeded
synthetic worlds from cartographer_amd.synth (C++ that travels with the repo), so only the
reference's OUTPUTS need to be stored.  Shared by the generator (make_reference_results.py, runs
the reference's own sources through oracle/_ref) and by the tests that replay them on the oracle
(CPU) and on the device (GPU).
"""
import math
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def quat(angle, axis):
    axis = np.asarray(axis, np.float64)
    axis = axis / np.linalg.norm(axis)
    s = math.sin(angle / 2)
    return [math.cos(angle / 2), axis[0] * s, axis[1] * s, axis[2] * s]


def fast2d_bench(synth):
    """bench.py's workload (BASELINE config[1]): submap seed 42, the scan bench.py draws."""
    cells, lim, world = synth.make_submap(42, 400, 400, 0.05, 30, 1000, 30.0, 0.01)
    pose = world.free_pose(1234, 0.5)
    scan = world.scan(pose, 1000, 30.0, 0.01, 7)
    init = [pose[0] + 0.9, pose[1] - 0.7, pose[2] + 0.2]
    return dict(cells=cells, lim=lim, scan=scan, depth=7, init=init, truth=list(pose))


def rt2d_c1(synth):
    """BASELINE config[0] as tools/time_configs.py c1 runs it."""
    cells, lim, world = synth.make_submap(42, 200, 200, 0.05, 30, 1000, 5.0, 0.01)
    pose = world.free_pose(1234, 0.5)
    scan = world.scan(pose, 1000, 5.0, 0.01, 7)
    init = [pose[0] + 0.12, pose[1] - 0.08, pose[2] + math.radians(3.0)]
    return dict(cells=cells, lim=lim, scan=scan, init=init, lin=0.3, ang=math.radians(7.0),
                tw=0.1, rw=0.1)


def rt2d_tsdf():
    f = np.load(os.path.join(HERE, "rt2d_tsdf_fixture.npz"))
    return dict(tsd=f["tsd"], weight=f["weight"], res=float(f["resolution"]),
                max_x=float(f["max_x"]), max_y=float(f["max_y"]),
                truncation=float(f["truncation_distance"]), max_weight=float(f["max_weight"]),
                cloud=f["cloud"],
                # CreateRealTimeCorrelativeScanMatcherTestOptions2D (..._2d_test.cc:40-50)
                lin=0.6, ang=0.16, tw=0.0, rw=0.0, init=[0.02, -0.03, 0.05])


def rt3d(synth):
    grid, world = synth.make_submap_3d(3, 0.1, (8.0, 8.0, 4.0), 4, 8, 96)
    pos = world.free_position(4, 0.5)
    cloud = world.scan(pos, 0.3, 6, 64, seed=9)
    init = list(pos + np.array([0.07, -0.04, 0.02])) + quat(0.31, [0.1, -0.2, 0.97])
    return dict(res=0.1, vox=grid.voxels(), cloud=cloud, init=init, lin=0.2,
                ang=math.radians(1.0), tw=0.1, rw=0.1)


def fast3d(synth):
    seed = 21
    grid, world = synth.make_submap_3d(seed, 0.1, (9.0, 8.0, 4.0), 5, 10, 128)
    low, _ = synth.make_submap_3d(seed, 0.45, (9.0, 8.0, 4.0), 5, 10, 128)
    rng = np.random.default_rng(seed)
    hist = rng.uniform(0.0, 1.0, 120).astype(np.float32)
    hist[10:14] += 6.0
    scan_hist = np.roll(hist, -19).copy()
    pos = world.free_position(seed + 3, 0.6)
    yaw = 0.4
    hi = world.scan(pos, yaw, 8, 96, seed=1)
    lo = hi[::7].copy()
    submap_pose = [0.3, -0.2, 0.1] + quat(0.2, [0, 0, 1])
    c, s = math.cos(0.2), math.sin(0.2)
    local = np.array([pos[0] + 0.35, pos[1] - 0.25, pos[2] + 0.1])
    node_t = [submap_pose[0] + c * local[0] - s * local[1],
              submap_pose[1] + s * local[0] + c * local[1], submap_pose[2] + local[2]]
    node_pose = node_t + quat(0.2 + yaw + 0.1, [0, 0, 1])
    return dict(res=0.1, vox=grid.voxels(), grid_size=grid.grid_size, low_res=0.45,
                low_vox=low.voxels(), hist=hist, scan_hist=scan_hist, hi=hi, lo=lo,
                node_pose=node_pose, submap_pose=submap_pose, gravity=quat(0.01, [1, 0, 0]),
                options=dict(depth=6, frd=3, min_rot=0.9, min_low=0.3, lin_xy=1.5, lin_z=0.5,
                             ang=math.radians(20.0)),
                min_score=0.15)


def rt3d_c4(synth, rings=64, az=1024):
    """BASELINE config[3] exactly as bench.py (Rt3DWorkload) and tools/time_configs.py c4 run it:
    64 rings x 1024 azimuths against a 150^3 HybridGrid, window 0.5 m / 2 deg (L = 5, A = 5:
    1331 translations x 1331 rotations = 1 771 561 candidates)."""
    grid, world = synth.make_submap_3d(42, 0.1, (15.0, 15.0, 7.5), 8, 32, 512)
    pos = world.free_position(77, 0.5)
    cloud = world.scan(pos, 0.3, rings, az, seed=9)
    c, s = math.cos(0.31 / 2), math.sin(0.31 / 2)
    init = list(pos + np.array([0.07, -0.04, 0.02])) + [c, 0.0, 0.0, s]
    return dict(res=0.1, vox=grid.voxels(), cloud=cloud, init=init, lin=0.5,
                ang=math.radians(2.0), tw=0.1, rw=0.1)


def rt3d_c4_shaped(synth):
    """C4's SHAPE at a size the oracle finishes in seconds: L = 5 (11^3 translations, 6^3 = 216
    groups of 2x2x2 per rotation -- the flat 192-lane group mapping of rt_3d.hip), A = 3
    (343 rotations), a tilted initial orientation, ~4 k points."""
    grid, world = synth.make_submap_3d(5, 0.1, (9.0, 8.0, 4.0), 5, 10, 128)
    pos = world.free_position(6, 0.6)
    cloud = world.scan(pos, 0.2, 32, 128, seed=3)
    init = list(pos + np.array([0.21, -0.13, 0.08])) + quat(0.27, [0.15, -0.25, 0.95])
    # max range 8.91 m -> angular step 0.01121 rad; 1.9 deg / step = 2.96 -> A = 3 (the tests
    # assert the candidate count 1331 * 343).
    return dict(res=0.1, vox=grid.voxels(), cloud=cloud, init=init, lin=0.5,
                ang=math.radians(1.9), tw=0.1, rw=0.1, num_candidates=1331 * 343)
