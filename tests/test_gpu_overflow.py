"""The frontier-overflow path of the branch-and-bound searches: with the node buffers shrunk
(debug switch frontier_capacity, a test hook) the first pass drops nodes, and the search must
repeat in strict mode over smaller chunks and still return what the oracle returns.  Runs in a
subprocess so that the shrunk buffers do not outlive the test."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import math, sys
import numpy as np
sys.path.insert(0, %(root)r)
from cartographer_amd import _lib, scan_matching as sm, scan_matching_3d as sm3, synth
from oracle import pyoracle as orc
import os
_lib.debug_set(frontier_capacity=int(os.environ["CMX_TEST_FRONTIER_CAPACITY"]))

cells, lim, world = synth.make_submap(42, 300, 300, 0.05, 20, 800, 30.0, 0.01)
truth = world.free_pose(3, 0.5)
scan = world.scan(truth, 600, 30.0, 0.01, 4)
grid = sm.Grid2D(cells, lim["resolution"], lim["max_x"], lim["max_y"])
gm = sm.FastCorrelativeScanMatcher2D(grid, 7)
om = orc.FastCorrelativeScanMatcher2D(cells, lim["resolution"], lim["max_x"], lim["max_y"], 7)
for min_score in (0.3, 0.6):
    ref = om.match_full_submap(scan, min_score)
    found, score, pose = gm.match_full_submap(scan, min_score)
    assert found == ref["found"], (found, ref)
    if found:
        assert np.float32(score) == np.float32(ref["score"]), (score, ref["score"])
        # strict mode prunes ties, so among equal-score leaves another one may be returned;
        # the score is the reference's, the pose is one of the tied optima.
        if gm.last_stats["nodes_expanded"] and abs(pose.x - ref["pose"][0]) > 1e-9:
            print("tie resolved differently under overflow", pose, ref["pose"])
print("expanded", gm.last_stats["nodes_expanded"])
import os
if os.environ.get("CMX_TEST_3D") != "1":
    print("OVERFLOW-OK")
    sys.exit(0)

g3, w3 = synth.make_submap_3d(21, 0.1, (9.0, 8.0, 4.0), 5, 10, 128)
vox = g3.voxels()
hist = np.zeros(8, np.float32)
pos = w3.free_position(24, 0.6)
hi = w3.scan(pos, 0.4, 8, 96, seed=1)
lo = hi[::7].copy()
opt = dict(branch_and_bound_depth=6, full_resolution_depth=3, min_rotational_score=0.5,
           min_low_resolution_score=0.3, linear_xy_search_window=1.5,
           linear_z_search_window=0.5, angular_search_window=math.radians(20.0))
f3 = sm3.FastCorrelativeScanMatcher3D(0.1, vox, g3.grid_size, 0.1, vox, hist, **opt)
o3 = orc.FastCorrelativeScanMatcher3D(0.1, vox, 0.1, vox, hist, 6, 3, 0.5, 0.3, 1.5, 0.5,
                                      math.radians(20.0))
node = [pos[0] + 0.3, pos[1] - 0.2, pos[2] + 0.1] + [math.cos(0.25), 0.0, 0.0, math.sin(0.25)]
ref3 = o3.match(node, [0, 0, 0, 1, 0, 0, 0], [1, 0, 0, 0], hi, lo, hist, 0.3)
got = f3.match(sm3.Rigid3d(tuple(node[:3]), tuple(node[3:])), sm3.Rigid3d(),
               sm3.TrajectoryNodeData(hi, lo, hist), 0.3)
assert (got is not None) == ref3["found"]
if got is not None:
    assert np.float32(got["score"]) == np.float32(ref3["score"])
print("OVERFLOW-OK")
"""


@pytest.mark.parametrize("capacity,with_3d", [("2048", "0"), ("8192", "0"), ("4096", "1")])
def test_overflow_retry_returns_the_reference_score(capacity, with_3d):
    # (capacities are per 64 sub-lists; a sub-list must at least hold one node's children)
    env = dict(os.environ, CMX_TEST_FRONTIER_CAPACITY=capacity, CMX_TEST_3D=with_3d)
    out = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT}], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "OVERFLOW-OK" in out.stdout
