"""C4 (RT-3D) timing probe: two matches per listed mode (1: bulk passes, 0: exhaustive), pass timings from stderr.
   python tools/time_rt3d.py [modes...]   (default: 1 0)"""
import math
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cartographer_amd import _lib, scan_matching_3d as sm3, synth  # noqa: E402

grid, world = synth.make_submap_3d(42, 0.1, (15.0, 15.0, 7.5), 8, 32, 512)
vox = grid.voxels()
pos = world.free_position(77, 0.5)
cloud = world.scan(pos, 0.3, 64, 1024, seed=9)
c, s = math.cos(0.31 / 2), math.sin(0.31 / 2)
init = sm3.Rigid3d(tuple(pos + np.array([0.07, -0.04, 0.02])), (c, 0.0, 0.0, s))
m = sm3.RealTimeCorrelativeScanMatcher3D(0.5, math.radians(2.0), 0.1, 0.1)
if os.environ.get("CMX_NO_REPORT") != "1":      # (the report's statistics atomics cost ~0.5 ms per pass)
    _lib.debug_set(rt3d_report=1)
for mode in (sys.argv[1:] or ["1", "0"]):          # 1: bulk passes, 0: exhaustive kernel only
    _lib.debug_set(rt3d_legacy=1 - int(mode))
    for rep in range(2):
        t0 = time.perf_counter()
        score, est = m.match(init, cloud, 0.1, vox)
        dt = time.perf_counter() - t0
        print(f"bulk {mode} rep {rep}: wall {dt * 1e3:.1f} ms score {score:.6f} stats {m.last_stats}",
              flush=True)
