import sys, math
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from cartographer_amd import synth, scan_matching as sm
cells, lim, world = synth.make_submap(42, 400, 400, 0.05, 30, 1000, 30.0, 0.01)
pose = world.free_pose(1234, 0.5)
scan = world.scan(pose, 1000, 30.0, 0.01, 7)
grid = sm.Grid2D(cells, lim["resolution"], lim["max_x"], lim["max_y"])
gm = sm.FastCorrelativeScanMatcher2D(grid, 4, 1.0, 0.3)
for n in (1, 2, 63):
    print("n", n, flush=True)
    r = gm.match(sm.Rigid2d(0.5, 0.5, 0.1), scan[:n], 0.05)
    print(r, gm.last_stats, flush=True)
