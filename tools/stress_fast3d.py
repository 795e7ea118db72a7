"""Stress of the fast-3D search on a small scene: repeats single and batched searches and
reports any result that differs from the first one (python tools/stress_fast3d.py [rounds])."""
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from cartographer_amd import scan_matching_3d as sm3, synth  # noqa: E402
from test_oracle_reference_pins_3d import quat_from_angle_axis  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
hist = np.zeros(16, np.float32)
opt = dict(branch_and_bound_depth=5, full_resolution_depth=2, min_rotational_score=0.0,
           min_low_resolution_score=0.2, linear_xy_search_window=1.0,
           linear_z_search_window=0.4, angular_search_window=math.radians(10.0))
matchers, worlds = [], []
for k in range(3):
    grid, world = synth.make_submap_3d(70 + k, 0.2, (8.0, 8.0, 3.0), 4, 8, 96)
    vox = grid.voxels()
    matchers.append(sm3.FastCorrelativeScanMatcher3D(0.2, vox, grid.grid_size, 0.2, vox, hist, **opt))
    worlds.append(world)
pos = worlds[0].free_position(200, 0.6)
hi = worlds[0].scan(pos, 0.0, 6, 64, seed=0)
data = sm3.TrajectoryNodeData(hi, hi[::5].copy(), hist, tuple(quat_from_angle_axis(0.01, [1, 0, 0])))
node = sm3.Rigid3d(tuple(pos + np.array([0.3, -0.2, 0.1])), tuple(quat_from_angle_axis(0.05, [0, 0, 1])))
ident = sm3.Rigid3d()
pairs = [(0, False, 0.12), (1, False, 0.12), (2, False, 0.99), (0, True, 0.12), (1, True, 0.3),
         (2, False, 0.12), (0, False, 0.4)]


def key(r):
    return None if r is None else (np.float32(r["score"]), np.float32(r["low_resolution_score"]),
                                   tuple(r["pose_estimate"].translation))


def single():
    out = []
    for k, full, t in pairs:
        out.append(key(matchers[k].match_full_submap(node.rotation, ident.rotation, data, t) if full
                       else matchers[k].match(node, ident, data, t)))
    return out


def batch():
    got, _ = sm3.fast3d_match_batch([matchers[k] for k, _, _ in pairs], [node] * len(pairs),
                                    [ident] * len(pairs), [f for _, f, _ in pairs],
                                    [t for _, _, t in pairs], data)
    return [key(g) for g in got]


first = single()
print("reference:", first, flush=True)
bad = 0
for r in range(rounds):
    for name, fn in (("single", single), ("batch", batch)):
        got = fn()
        if got != first:
            bad += 1
            print(f"round {r} {name}: DIFFERENT", [(i, a, b) for i, (a, b) in enumerate(zip(first, got)) if a != b],
                  flush=True)
print(f"{rounds} rounds, {bad} differing results")
