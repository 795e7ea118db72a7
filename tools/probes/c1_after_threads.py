"""C1 batches in a process that has already served 8 concurrent fast-2D callers (eight workspaces
and streams in the pool, as in bench.py's default run): do the parts of a batch still overlap?
   python tools/probes/c1_after_threads.py [matches ...]"""
import argparse
import math
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from cartographer_amd import grid_2d, scan_matching as sm, synth  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [1024]
args = argparse.Namespace(submaps=0, grid=400, depth=7, beams=1000, min_score=0.6, scans=1)
w = bench.Fast2DWorkload(args, 0, 0, 1, sharded=False)
pool = ThreadPoolExecutor(8)
list(pool.map(lambda _: [w.search() for _ in range(50)], range(8)))

m = sm.RealTimeCorrelativeScanMatcher2D(0.3, math.radians(7.0), 0.1, 0.1)
worlds = []
for k in range(8):
    cells, lim, world = synth.make_submap(42 + k, 200, 200, 0.05, 30, 1000, 5.0, 0.01)
    pose = world.free_pose(1234, 0.5)
    worlds.append((grid_2d.ProbabilityGridOnDevice(0.05, (lim["max_x"], lim["max_y"]), 200, 200, cells=cells),
                   world.scan(pose, 1000, 5.0, 0.01, 7),
                   (pose[0] + 0.12, pose[1] - 0.08, pose[2] + math.radians(3.0))))
for batch in sizes:
    b = sm.Rt2DBatch(m, [worlds[i % 8][0] for i in range(batch)], [worlds[i % 8][1] for i in range(batch)],
                     resident=True)
    init = np.array([worlds[i % 8][2] for i in range(batch)])
    for _ in range(5):
        b.match(init)
    t0 = time.perf_counter()
    for _ in range(30):
        b.match(init)
    print(f"C1 batch {batch} after 8 concurrent fast-2D callers: {(time.perf_counter() - t0) / 30 * 1e6:.1f} us per call",
          flush=True)
