import sys, math
sys.path.insert(0, '/root/repo')
import numpy as np
from cartographer_amd import synth, scan_matching as sm
from oracle import pyoracle as orc
cells, lim, world = synth.make_submap(42, 400, 400, 0.05, 30, 1000, 30.0, 0.01)
truth = world.free_pose(1234, 0.5)
scan = world.scan(truth, 1000, 30.0, 0.01, 7)
init = [truth[0] + 0.4, truth[1] - 0.3, truth[2] + 0.15]
grid = sm.Grid2D(cells, lim["resolution"], lim["max_x"], lim["max_y"])
for depth in (7, 5, 3):
  for ms in (0.1, 0.55):
    gm = sm.FastCorrelativeScanMatcher2D(grid, depth, 7.0, math.radians(30.0))
    om = orc.FastCorrelativeScanMatcher2D(cells, 0.05, lim["max_x"], lim["max_y"], depth, 7.0, math.radians(30.0))
    r = gm.match(sm.Rigid2d(*init), scan, ms)
    ref = om.match(init, scan, ms)
    print(depth, ms, "gpu", r[0], r[1], gm.last_stats)
    print("   ref", ref["found"], ref["score"], ref["candidates_scored"], ref["coarse_candidates"], ref["nodes_expanded"])
    a = gm.debug_prepare(sm.Rigid2d(*init), scan, False)
    b = om.prepare(init, scan, False)
    print("   sums equal", np.array_equal(a["sums"], b["sums"]), "bounds equal", np.array_equal(a["bounds"], b["bounds"]), a["bounds"][:2])
