"""C4 (RT-3D): the group pass under a few workgroup shapes.   python tools/group_sweep.py"""
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cartographer_amd import _lib, scan_matching_3d as sm3, synth  # noqa: E402
from cartographer_amd import _lib as _cmx_lib  # noqa: E402
_cmx_lib.debug_set(timing=1)   # cmx_match_stats *_ms are recorded only on request

grid, world = synth.make_submap_3d(42, 0.1, (15.0, 15.0, 7.5), 8, 32, 512)
vox = grid.voxels()
pos = world.free_position(77, 0.5)
cloud = world.scan(pos, 0.3, 64, 1024, seed=9)
c, s = math.cos(0.31 / 2), math.sin(0.31 / 2)
init = sm3.Rigid3d(tuple(pos + np.array([0.07, -0.04, 0.02])), (c, 0.0, 0.0, s))
m = sm3.RealTimeCorrelativeScanMatcher3D(0.5, math.radians(2.0), 0.1, 0.1)
ref = None
for rots, tile_kb in [tuple(a.split(":")) for a in sys.argv[1:]] or (("8", "44"), ("4", "44"), ("2", "44")):
    _lib.debug_set(rt3d_group_rotations=int(rots), rt3d_group_tile_kb=int(tile_kb))
    best, group = 1e9, 1e9
    for rep in range(4):
        score, est = m.match(init, cloud, 0.1, vox)
        best = min(best, m.last_stats["device_ms"])
        group = min(group, m.last_stats["dominant_kernel_ms"])
    key = (np.float32(score), tuple(est.translation), tuple(est.rotation))
    ref = ref or key
    print(f"group rotations {rots} tile {tile_kb} KB: device {best:.3f} ms, group pass {group:.3f} ms"
          f"{'' if key == ref else '  RESULT DIFFERS'}", flush=True)
