#!/usr/bin/env python3
"""In-kernel timeline of one C1 batch call (debug switch timeline): python tools/probes/c1_timeline.py
<matches> [name=value ...]"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cartographer_amd import _lib, grid_2d, scan_matching as sm, synth
import numpy as np
num = int(sys.argv[1])
sets = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in sys.argv[2:]}
rt = sm.RealTimeCorrelativeScanMatcher2D(0.3, math.radians(7.0), 0.1, 0.1)
G, I, S = [], [], []
for k in range(8):
    cells, lim, world = synth.make_submap(42 + k, 200, 200, 0.05, 30, 1000, 5.0, 0.01)
    pose = world.free_pose(1234, 0.5)
    G.append(grid_2d.ProbabilityGridOnDevice(0.05, (lim["max_x"], lim["max_y"]), 200, 200, cells=cells))
    S.append(world.scan(pose, 1000, 5.0, 0.01, 7))
    I.append([pose[0] + 0.12, pose[1] - 0.08, pose[2] + math.radians(3.0)])
batch = sm.Rt2DBatch(rt, [G[i % 8] for i in range(num)], [S[i % 8] for i in range(num)], resident=True)
init = np.array([I[i % 8] for i in range(num)], np.float64)
for _ in range(3):
    batch.match(init)
_lib.debug_set(timeline=1, **sets)
print(f"--- C1 batch {num} {sets}", file=sys.stderr)
scores, poses, stats = batch.match(init)
print("    score[0]", float(scores[0]), file=sys.stderr)
