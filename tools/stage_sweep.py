"""C4 (RT-3D): the staged second round under a few settings (segment ends in sixteenths of a
window, lanes per work block, rotations per list).   python tools/stage_sweep.py"""
import itertools
import math
import os
import sys
import time

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
settings = [("48", "512", "8")]
settings += [(seg, "512", "8") for seg in ("36", "37", "38", "47", "4a", "58", "5a", "6a", "26")]
settings += [("48", thr, rot) for thr, rot in (("256", "8"), ("1024", "8"), ("512", "4"), ("256", "4"))]
settings += [("48", "512", "8")]
for seg, threads, rots in settings:
    _lib.debug_set(rt3d_segments=int(seg[0], 16) | (int(seg[1], 16) << 8),
                   rt3d_cand_threads=int(threads), rt3d_cand_rotations=int(rots))
    best = 1e9
    for rep in range(4):
        score, est = m.match(init, cloud, 0.1, vox)
        best = min(best, m.last_stats["device_ms"])
    key = (np.float32(score), tuple(est.translation), tuple(est.rotation))
    ref = ref or key
    print(f"segments {seg} lanes {threads} rotations {rots}: device {best:.3f} ms"
          f"{'' if key == ref else '  RESULT DIFFERS'}", flush=True)
