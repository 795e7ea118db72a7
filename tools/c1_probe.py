"""C1 (real-time 2D) probe: wall / device / tile-kernel time and the three stages' candidate
counts for single matches and batches on resident grids.
   python tools/c1_probe.py [batch ...] [--grid 200] [--dirty] [--set name=value ...]"""
import argparse
import math
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cartographer_amd import _lib, grid_2d, scan_matching as sm, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("batches", nargs="*", type=int, default=[1, 128])
ap.add_argument("--grid", type=int, default=200)
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--dirty", action="store_true", help="bump every grid's version before each call")
ap.add_argument("--set", action="append", default=[], help="debug switch name=value")
args = ap.parse_args()
if args.set and hasattr(_lib.lib(), "cmx_debug_set"):
    _lib.debug_set(**{kv.split("=")[0]: int(kv.split("=")[1]) for kv in args.set})

m = sm.RealTimeCorrelativeScanMatcher2D(0.3, math.radians(7.0), 0.1, 0.1)
worlds = []
for k in range(8):                      # 8 distinct worlds, reused round-robin
    cells, lim, world = synth.make_submap(42 + k, args.grid, args.grid, 0.05, 30, 1000, 5.0, 0.01)
    pose = world.free_pose(1234, 0.5)
    worlds.append((grid_2d.ProbabilityGridOnDevice(0.05, (lim["max_x"], lim["max_y"]), args.grid,
                                                   args.grid, cells=cells),
                   world.scan(pose, 1000, 5.0, 0.01, 7),
                   sm.Rigid2d(pose[0] + 0.12, pose[1] - 0.08, pose[2] + math.radians(3.0)), pose))
for batch in args.batches:
    G = [worlds[i % 8][0] for i in range(batch)]
    S = [worlds[i % 8][1] for i in range(batch)]
    I = np.array([[worlds[i % 8][2].x, worlds[i % 8][2].y, worlds[i % 8][2].theta] for i in range(batch)])
    b = sm.Rt2DBatch(m, G, S, resident=True)
    for _ in range(3):
        b.match(I)
    wall = dev = ker = 0.0
    for _ in range(args.reps):
        if args.dirty:
            for g, scan, _, pose in worlds[:min(batch, 8)]:
                c, s_ = math.cos(pose[2]), math.sin(pose[2])
                pts = np.zeros((8, 3), np.float32)          # a tiny scan: bumps the version
                pts[:, 0] = pose[0] + c * scan[:8, 0] - s_ * scan[:8, 1]
                pts[:, 1] = pose[1] + s_ * scan[:8, 0] + c * scan[:8, 1]
                g.insert(pose[:2], pts)
        t0 = time.perf_counter()
        scores, poses, st = b.match(I)
        wall += time.perf_counter() - t0
        dev += st["device_ms"]
        ker += st["dominant_kernel_ms"]
    wall /= args.reps
    cand = st["candidates_scored"]
    print(f"C1 grid {args.grid} batch {batch:5d}{' dirty' if args.dirty else ''}: wall {wall * 1e6:8.1f} us, "
          f"device {dev / args.reps * 1e3:7.1f} us, bulk kernel {ker / args.reps * 1e3:7.1f} us -> "
          f"{cand / wall:.3e} cand/s wall; per match: refined {st.get('refined_candidates', 0) / batch:.1f}, "
          f"f32 finalists {st.get('finalists', 0) / batch:.1f}; score[0] {scores[0]:.6f}", flush=True)
